// Occlusion kernels of the MI355X-native Shrake-Rupley engine (gfx950 only).
//
// Atoms are taken in cell-sorted order (a workgroup's waves work on spatial neighbours, and the
// XCD-aware remap gives every XCD a contiguous range of structures so its L2 holds only those).
// Four kernels make the same decisions:
//   k_occlusion_mx (occlusion_mx.inc)      batches of 32 768 atoms or more: 64 atoms per wave, a group
//       of neighbouring atoms shares one union of swept atoms in registers, point tests on the matrix
//       instructions (f16 filter, exact f32); atoms it cannot take go to the batch's deferred list;
//   k_occlusion_fast (occlusion_fast.inc)  smaller batches, one wavefront per atom: the straight-line kernel for the common case
//       (n_points <= 128, <= 4 remainder points); atoms it cannot take (more than 256 atoms in
//       the culled runs, more than 144 candidates) go to the batch's deferred list;
//   k_occlusion_v3 (occlusion_v3.inc)      the general kernel: any n_points (groups of two
//       64-point chunks), any list length (several flushes), any remainder count; runs over the
//       deferred list after the fast kernel, or over every atom when the fast one does not apply;
//   k_occlusion_v0 (occlusion_v0.inc)      all candidates x all points, no culling: an
//       independent implementation for A/B checks (RSASA_OCCLUSION_KERNEL=0).
//
// Stages of the per-atom kernels (details differ between fast and v3, see the files; k_occlusion_mx: its own file):
//   PROLOGUE (per wave, a group of its atoms at once, 64 (atom, run) pairs per pass): the 25
//      x-runs of cells of the 5x5x5 block around each atom's cell (search_extent = 2, reference
//      spatial_grid.rs:47: max_search / cell_size is exactly 2 in f32), culled and trimmed with
//      a conservative lower bound on the distance.
//   Then per atom:
//   1. the run lengths are prefix-summed (DPP); a 256-bit mask of run starts and a per-run
//      `start - prefix` table map every flat position of the concatenated runs to its atom;
//   2. SWEEP: 64 flat positions per iteration: the reference's candidate rule
//      d^2 <= (r_i + max_r + 2p)^2 (spatial_grid.rs:307-308,335); accepted atoms go to the
//      wave's LDS list (fast: records after the id rule; v3: indices, id rule in the prep pass);
//   3. PREP (lane = candidate): v and limit_j (lib.rs:128-136); the list becomes
//      (vx, vy, vz, limit) records, NEAR candidates (strong occluders) first;
//   4. remainder points (scalar rule, lib.rs:163-218): fast: uniform values tested inside the
//      prep pass; v3: their own pass, lanes tiled (point x candidate);
//   5. PHASE A (lane = sphere point): near candidates are broadcast from LDS and tested against
//      all fused-rule points (lib.rs:143-147): ~90 % of the points are occluded after ~11
//      candidates;
//   6. PHASE B: the surviving points are compacted and tested against the far candidates with
//      lanes tiled as (survivor x candidate), e.g. 8 survivors x 8 candidates per instruction.
//
// The result is an OR over the candidate set, so the order of tests is free; every individual
// test uses the reference's exact f32 expressions.
#include "device_utils.h"

#include <atomic>

namespace rsasa {
namespace {

struct OccArgs {
    BatchView b;
    Lattice lat;
    uint32_t n_blocks;        // launched workgroups (for the XCD swizzle)
    uint32_t atoms_per_wave;  // consecutive cell-sorted atoms handled by one wave
    uint32_t debug_stop;      // timing ablation (v3 only)
};

#include "occlusion_v0.inc"

__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}


#include "occlusion_v3.inc"
#include "occlusion_fast.inc"
#include "occlusion_mx.inc"

template <int NCH>
void launch_v0(const OccArgs &a, hipStream_t stream)
{
    if (a.b.id) hipLaunchKernelGGL((k_occlusion_v0<NCH, true>), dim3(a.n_blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_occlusion_v0<NCH, false>), dim3(a.n_blocks), dim3(256), 0, stream, a);
}

template <bool HAS_ID, bool HAS_REM>
void launch_fast2(bool half1, uint32_t n_blocks, hipStream_t stream, const OccArgs3 &a3)
{
    if (half1) hipLaunchKernelGGL((k_occlusion_fast<HAS_ID, HAS_REM, true>), dim3(n_blocks), dim3(256), 0, stream, a3);
    else hipLaunchKernelGGL((k_occlusion_fast<HAS_ID, HAS_REM, false>), dim3(n_blocks), dim3(256), 0, stream, a3);
}

// Compute units of the current device (persistent launches are sized by what the GPU holds at once: 256 on MI355X; a
// partitioned or different device gets its own count; results never depend on it).  The per-XCD pieces of such a launch
// (occlusion_mx.inc) assume eight XCDs dealt round-robin: on another layout the pieces only lose their L2 locality.
uint32_t device_cus()
{
    static std::atomic<uint32_t> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u;
    uint32_t n = cached[dev].load(std::memory_order_relaxed);
    if (!n) {
        int cu = 0;
        n = (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) ? (uint32_t)cu : 256u;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// NW waves per workgroup, each with up to atoms_per_wave atoms
// OcclusionChain: in front of / behind the launch that does a batch's occlusion work
void chain_begin(const OcclusionChain *chain, hipStream_t stream)
{
    if (!chain) return;
    if (chain->wait) (void)hipStreamWaitEvent(stream, chain->wait, 0);  // (a failure shows as the launch's)
    if (chain->start) (void)hipEventRecord(chain->start, stream);
}
void chain_end(const OcclusionChain *chain, hipStream_t stream)
{
    if (chain && chain->done) (void)hipEventRecord(chain->done, stream);
}

template <int NT, bool HAS_ID, bool MULTI, int NW>
void launch_mx1(bool rem, uint32_t n_blocks, uint32_t lds_bytes, hipStream_t stream, const OccArgs3 &a3)
{
    if (rem) hipLaunchKernelGGL((k_occlusion_mx<NT, HAS_ID, true, MULTI, NW>), dim3(n_blocks), dim3(64 * NW), lds_bytes, stream, a3);
    else hipLaunchKernelGGL((k_occlusion_mx<NT, HAS_ID, false, MULTI, NW>), dim3(n_blocks), dim3(64 * NW), lds_bytes, stream, a3);
}

template <int NT, bool MULTI, int NW>
void launch_mx(bool has_id, bool rem, uint32_t n_atoms, uint32_t lds_bytes, hipStream_t stream, const OccArgs3 &a3,
               const OcclusionChain *chain)
{
    // persistent waves (occlusion_mx.inc): as many workgroups as the GPU holds at once - 28 waves per CU at 72
    // registers, 24 at 80 (NW >= 8) - or fewer when the batch has fewer blocks of atoms_per_wave atoms than that
    const uint32_t resident = device_cus() * (NW == 4 ? 7u : NW == 8 ? 3u : 2u);
    const uint32_t n_blocks = mx_persistent(MULTI) ? min(cdiv(n_atoms, NW * a3.atoms_per_wave), resident) : cdiv(n_atoms, NW * a3.atoms_per_wave);
    if (mx_persistent(MULTI)) (void)hipMemsetAsync(a3.claim, 0, 8u * kClaimStride * 4u, stream);  // (a failure shows as the launch's)
    if (has_id && a3.ids_check) {
        // BatchView::ids_check: which structures keep their ids is known on the device only - both instantiations are launched,
        // each works on the structures that are its own (StructGrid::odd_radii bit 1), and the one with ids returns at once
        // when there is none (its workgroups touch nothing, the claim counters included).
        const bool dropped = !chain || chain->expect_ids_dropped;
        if (chain && chain->solo && dropped) {
            // (the caller runs the batch again if the guess was wrong: OcclusionChain::solo)
            OccArgs3 only = a3;
            only.ids_check = 2u;
            chain_begin(chain, stream);
            launch_mx1<NT, false, MULTI, NW>(rem, n_blocks, lds_bytes, stream, only);
        } else if (dropped) {
            // (the instantiation with ids is expected to return at once - no structure kept its ids last time -: in front of
            // the chain's wait, where it runs beside the neighbour's kernel for nothing)
            launch_mx1<NT, true, MULTI, NW>(rem, n_blocks, lds_bytes, stream, a3);
            chain_begin(chain, stream);
            // (persistent waves claim their blocks from counters: either launch of the pair may have worked through them)
            if (mx_persistent(MULTI)) (void)hipMemsetAsync(a3.claim, 0, 8u * kClaimStride * 4u, stream);
            launch_mx1<NT, false, MULTI, NW>(rem, n_blocks, lds_bytes, stream, a3);
        } else {
            // some structures keep their ids, the others do not: both launches work, each on the structures that are its own
            chain_begin(chain, stream);
            launch_mx1<NT, false, MULTI, NW>(rem, n_blocks, lds_bytes, stream, a3);
            if (mx_persistent(MULTI)) (void)hipMemsetAsync(a3.claim, 0, 8u * kClaimStride * 4u, stream);
            launch_mx1<NT, true, MULTI, NW>(rem, n_blocks, lds_bytes, stream, a3);
        }
    } else {
        chain_begin(chain, stream);
        if (has_id) launch_mx1<NT, true, MULTI, NW>(rem, n_blocks, lds_bytes, stream, a3);
        else launch_mx1<NT, false, MULTI, NW>(rem, n_blocks, lds_bytes, stream, a3);
    }
    chain_end(chain, stream);
}

// (Waves per workgroup of k_occlusion_mx with at most 128 points: 4, sharing an LDS copy of the point tables.  Single-wave
// workgroups that read the tables from global memory - a wave's LDS is free the moment it ends, wave-slot utilisation
// 90 -> 94 % - measured the same time, 3.497 against 3.473 ms: DESIGN.md 7, tools/experiments/round6_removed_switches.patch.)
constexpr uint32_t kMxLdsBudget = 157u * 1024u;  // what workgroups of k_occlusion_mx can share of a CU's 160 KB (see launch_occlusion)

// static LDS of the many-point instantiations per wave (the same for every NW: the lists are per wave).  Contexts on
// several host threads may launch at once: the cached value is an atomic (every thread that finds it empty stores the
// same number).
uint32_t mx_multi_lds_per_wave()
{
    static std::atomic<uint32_t> cached{0};
    uint32_t bytes = cached.load(std::memory_order_relaxed);
    if (!bytes) {
        hipFuncAttributes at{};
        if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&k_occlusion_mx<8, true, true, true, 4>)) == hipSuccess && at.sharedSizeBytes)
            bytes = (uint32_t)at.sharedSizeBytes / 4u;
        else
            bytes = 3584u;
        cached.store(bytes, std::memory_order_relaxed);
    }
    return bytes;
}

void launch_fast(bool has_id, bool rem, bool half1, uint32_t n_blocks, hipStream_t stream, const OccArgs3 &a3)
{
    if (has_id && rem) launch_fast2<true, true>(half1, n_blocks, stream, a3);
    else if (has_id) launch_fast2<true, false>(half1, n_blocks, stream, a3);
    else if (rem) launch_fast2<false, true>(half1, n_blocks, stream, a3);
    else launch_fast2<false, false>(half1, n_blocks, stream, a3);
}

}  // namespace

void launch_occlusion(const BatchView &b, const Lattice &lat, const OcclusionTuning &tune,
                      OcclusionPart part, hipStream_t stream, const OcclusionChain *chain)
{
    // (the chain's events are recorded on every path: only the matrix-core kernel's launch places them itself)
    struct ChainGuard {
        const OcclusionChain *chain;
        hipStream_t stream;
        bool begun = false, ended = false;
        void begin() { if (!begun) { chain_begin(chain, stream); begun = true; } }
        ~ChainGuard() { begin(); if (!ended) chain_end(chain, stream); }
    } guard{chain, stream};
    if (!b.n_atoms) return;
    OccArgs a{b, lat, 0, 1, tune.debug_stop};
    const uint32_t n_chunks = (lat.n_points + kWave - 1) / kWave;
    // the straight-line kernels: few remainder points; up to 128 points (k_occlusion_fast), or up to
    // kMxMaxPoints with the matrix-core kernel
    // (6 = default: the matrix-core kernel once the batch has enough atoms to fill the GPU with its
    // 64-atom waves; smaller batches - single structures - finish sooner on the per-atom kernels)
    const bool mx = occlusion_uses_mx(tune, lat, b.n_atoms);
    const bool fast = tune.kernel_version >= 4 && tune.debug_stop == 0 && (n_chunks <= 2 || mx) &&
                      lat.n_points - lat.n_fused <= kFastMaxRem;
    if (!fast && part == kOccHead) return;  // only the fast kernel takes a partial range
    if (!(fast && mx)) guard.begin();  // (launch_mx places the chain's events around its working launch)
    if (tune.kernel_version == 0) {
        // reference kernel: all candidates against all points, one atom per wave
        a.n_blocks = cdiv(b.n_atoms, 4);
        if (n_chunks <= 2) launch_v0<2>(a, stream);
        else if (n_chunks <= 4) launch_v0<4>(a, stream);
        else launch_v0<16>(a, stream);
        return;
    }
    if (tune.atoms_per_wave > 0) {
        a.atoms_per_wave = tune.atoms_per_wave;
    } else {
        // keep >= ~8 waves per SIMD in flight on 256 CUs before giving a wave more than one atom
        const uint32_t waves_full = 256u * 4u * 8u * 4u;
        a.atoms_per_wave = max(1u, b.n_atoms / waves_full);
    }
    // (the fast kernel takes its atoms in groups of kFastAtomsPerWave, so any count works there)
    a.atoms_per_wave = min(a.atoms_per_wave, fast ? 4u * (uint32_t)kFastAtomsPerWave : (uint32_t)kMaxAtomsPerWave);
    if (fast && a.atoms_per_wave > (uint32_t)kFastAtomsPerWave)
        a.atoms_per_wave -= a.atoms_per_wave % (uint32_t)kFastAtomsPerWave;
    a.n_blocks = cdiv(cdiv(b.n_atoms, a.atoms_per_wave), 4);
    OccArgs3 a3 = make_args3(a);
    if (fast) {
        // straight-line kernel for every atom it can take; the rest go through the general kernel
        a3.work_list_out = b.deferred_list;
        a3.work_count_out = &b.status->deferred;
        a3.defer_flag = mx ? nullptr : b.defer_flag;  // (only k_occlusion_fast reports through the flag)
        const bool rem = lat.n_points != lat.n_fused;
        const bool half1 = lat.n_fused <= 96u;  // the second chunk's fused points fit half a wave
        a3.part = part;
        a3.claim = b.claim + (part == kOccRest ? 8u * kClaimStride : 0u);
        if (mx) {
            // group-union sweep + matrix-core point tests: 64 atoms per wave
            // 64 atoms per wave once that still leaves about four rounds of waves (256 CUs x 4 SIMDs x 7
            // waves resident) - fewer (down to 8) for smaller batches, where the last, partly filled round
            // and, for single structures, the time of one wave set the time of the call.  With many
            // points an atom is a microsecond of work and a wave's atoms share less of it: 16 rounds.
            const bool multi = lat.n_points > 128u;  // (the remainder points are columns of the matrix tests too)
            const uint32_t rounds = multi ? 16u : 4u, apw_max = multi ? kMxAtomsMulti : kMxAtoms;
            uint32_t apw = apw_max;
            if (tune.atoms_per_wave > 0) apw = min(tune.atoms_per_wave, apw_max);
            else while (apw > 8u && (uint64_t)apw * (256u * 4u * 7u * rounds) > b.n_atoms) apw >>= 1;
            a3.atoms_per_wave = apw;
            // dynamic LDS: the points as f32 (16 B each) and f16 (8 B each) matrix operands
            const bool has_id = b.id != nullptr;
            // (+ 16 zero entries of the f32 table: the column that pads phase B's last round)
            constexpr int kNW = 4;
            constexpr uint32_t kTabs = 1u;
            if (lat.n_points <= 96u) launch_mx<6, false, kNW>(has_id, rem, b.n_atoms, kTabs * (24u * 96u + 256u), stream, a3, chain);
            else if (lat.n_points <= 112u) launch_mx<7, false, kNW>(has_id, rem, b.n_atoms, kTabs * (24u * 112u + 256u), stream, a3, chain);
            else if (lat.n_points <= 128u) launch_mx<8, false, kNW>(has_id, rem, b.n_atoms, kTabs * (24u * 128u + 256u), stream, a3, chain);
            else {
                // More points: whole tiles of 16 points in the two tables, the patch table behind them (16 bytes per tile,
                // whole blocks of 64, and a zero entry).  The tables are per workgroup, the lists per wave (3.4 KB): pick
                // the waves per workgroup - 4, 8 or 12; 9 or 14 spread unevenly over the four SIMDs and measured 18 %
                // slower - that keeps most waves on a CU.  Its 160 KB of LDS are not all there for the taking: three
                // workgroups of 52 240 bytes ran side by side, three of 53 776 did not, so the budget is set to 157 KB.  With
                // equal wave counts the smaller workgroup wins (960 points: 3 x 8 waves 0.87 ms, 2 x 12 waves 0.93 ms).
                const uint32_t dyn = 24u * 16u * cdiv(lat.n_points, 16u) + 256u + 16u * (cdiv(cdiv(lat.n_points, 16u), 64u) * 64u) + 16u;
                const uint32_t per_wave = mx_multi_lds_per_wave();
                uint32_t best_nw = 4u, best_waves = 0u;
                for (uint32_t nw = 4u; nw <= 12u; nw += 4u) {
                    const uint32_t total = per_wave * nw + dyn;
                    const uint32_t waves = min(nw * (kMxLdsBudget / total), nw >= 8u ? 24u : 28u);  // (72 / 80 registers: 7 / 6 waves per SIMD)
                    if (waves > best_waves) { best_nw = nw; best_waves = waves; }
                }
                if (best_nw == 4u) launch_mx<8, true, 4>(has_id, rem, b.n_atoms, dyn, stream, a3, chain);
                else if (best_nw == 8u) launch_mx<8, true, 8>(has_id, rem, b.n_atoms, dyn, stream, a3, chain);
                else launch_mx<8, true, 12>(has_id, rem, b.n_atoms, dyn, stream, a3, chain);
            }
            guard.begun = guard.ended = true;  // (done by launch_mx)
        } else {
            launch_fast(b.id != nullptr, rem, half1, a.n_blocks, stream, a3);
        }
        if (part == kOccHead) return;  // the general kernel follows the last fast launch
        if (a3.defer_flag) return;     // the caller launches it if the flag says so
        a3.work_list = b.deferred_list;
        a3.work_count = &b.status->deferred;
        a3.atoms_per_wave = 1;
        // grid-stride over the list, which is usually empty: as many workgroups as the last batch's list would have kept
        // busy twice over (16 at least; all 1 024 while nothing is known)
        const uint32_t want = tune.deferred_hint == 0xFFFFFFFFu ? 1024u : min(1024u, max(16u, cdiv(tune.deferred_hint, 2u)));
        const uint32_t n_blocks = min(cdiv(b.n_atoms, 4), want);
        if (b.id) hipLaunchKernelGGL((k_occlusion_v3<2, true, false>), dim3(n_blocks), dim3(256), 0, stream, a3);
        else hipLaunchKernelGGL((k_occlusion_v3<2, false, false>), dim3(n_blocks), dim3(256), 0, stream, a3);
        return;
    }
#ifdef RSASA_ABLATE  // timing ablation build (make ablate, tools/ablate.sh): results are wrong; not in the shipped library
    if (tune.debug_stop != 0) {
        if (b.id) hipLaunchKernelGGL((k_occlusion_v3<2, true, true>), dim3(a.n_blocks), dim3(256), 0, stream, a3);
        else hipLaunchKernelGGL((k_occlusion_v3<2, false, true>), dim3(a.n_blocks), dim3(256), 0, stream, a3);
    } else
#endif
    if (b.id) {
        hipLaunchKernelGGL((k_occlusion_v3<2, true, false>), dim3(a.n_blocks), dim3(256), 0, stream, a3);
    } else {
        hipLaunchKernelGGL((k_occlusion_v3<2, false, false>), dim3(a.n_blocks), dim3(256), 0, stream, a3);
    }
}

bool occlusion_uses_mx(const OcclusionTuning &tune, const Lattice &lat, uint32_t n_atoms)
{
    return tune.kernel_version >= 5 && tune.debug_stop == 0 && lat.n_points <= kMxMaxPoints &&
           lat.n_points - lat.n_fused <= kFastMaxRem && (tune.kernel_version == 5 || n_atoms >= kMxMinAtoms);
}

void launch_occlusion_deferred(const BatchView &b, const Lattice &lat, hipStream_t stream)
{
    if (!b.n_atoms) return;
    OccArgs a{b, lat, 0, 1, 0};
    OccArgs3 a3 = make_args3(a);
    a3.work_list = b.deferred_list;
    a3.work_count = &b.status->deferred;
    a3.atoms_per_wave = 1;
    const uint32_t n_blocks = min(cdiv(b.n_atoms, 4), 1024u);
    if (b.id) hipLaunchKernelGGL((k_occlusion_v3<2, true, false>), dim3(n_blocks), dim3(256), 0, stream, a3);
    else hipLaunchKernelGGL((k_occlusion_v3<2, false, false>), dim3(n_blocks), dim3(256), 0, stream, a3);
}

}  // namespace rsasa

#ifdef MX_STAGE_PROF
// diagnostic build only (tools/mx_stage_prof.py): the stage stamps of k_occlusion_mx summed over every wave since the last call
extern "C" __attribute__((visibility("default"))) int rsasa_debug_mx_prof(unsigned long long *out)
{
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rsasa::g_mx_prof), sizeof(zero)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(rsasa::g_mx_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif
