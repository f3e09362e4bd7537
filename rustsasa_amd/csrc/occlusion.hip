// Occlusion kernels of the MI355X-native Shrake-Rupley engine (gfx950 only).
//
// One wavefront per atom, atoms taken in cell-sorted order (a workgroup's four
// waves work on spatial neighbours, and the XCD-aware remap gives every XCD a
// contiguous range of structures so its L2 holds only those).
//
//   1. lanes 0..24 own the 25 x-runs of cells of the 5x5x5 block around the
//      atom's cell (search_extent = 2, reference spatial_grid.rs:47: the ratio
//      max_search / cell_size is exactly 2 in f32).  Runs (and their end cells)
//      that cannot hold a candidate are culled with a conservative lower bound.
//   2. SWEEP: the runs are flattened and swept 64 atoms at a time with the
//      reference's candidate rule d^2 <= (r_i + max_r + 2p)^2
//      (spatial_grid.rs:307-308,335) and id rule (:314).  Accepted atoms are
//      appended to the wave's LDS list, NEAR ones (strong occluders) from the
//      front and FAR ones from the back.
//   3. PREP: lanes become candidates and compute limit_j (lib.rs:128-136).
//   4. PHASE A: lanes become sphere points; every NEAR candidate is broadcast
//      from LDS and tested against all points (lib.rs:143-147).  Typically
//      ~85 % of the points are already occluded after ~10 near candidates.
//   5. PHASE B: lanes become candidates again; each SURVIVING point is
//      broadcast and tested against all candidates at once, with the fused
//      `<` rule for points below n_fused and the reference's scalar remainder
//      rule (unfused, `<=`, lib.rs:185-186) for the rest.
//
// The result is an OR over the candidate set, so the order of tests is free;
// every individual test uses the reference's exact f32 expressions.
#include "device_utils.h"

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

constexpr int kCap = 160;             // LDS candidate slots per wave
constexpr int kFlushAt = kCap - 64;   // flush once more than this many are queued

__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// ------------------------------------------------------------------ v2 ----
//
// Same mathematics as v1, restructured to cut scalar-ALU and per-iteration
// overhead (the scalar unit is shared by a CU's four SIMDs):
//   * the culled x-runs are cut into 16-atom segments; a sweep iteration takes
//     four segments (16 lanes each), so the lane -> atom map is one LDS read;
//   * the remainder points (scalar rule, lib.rs:163-218) are decided in their
//     own pass, lanes tiled as (point x candidate);
//   * after phase A the surviving points are compacted through LDS and tested
//     against the far candidates with lanes tiled as (point x candidate), e.g.
//     16 survivors x 4 candidates per instruction.

constexpr int kSegCap = 64;  // 16-atom segments per table round

template <int NCH>
struct OccV2Cfg {
    static constexpr int kMaxSurvChunks = NCH <= 4 ? 1 : 4;  // compacted survivor chunks
};

template <int NCH, bool HAS_ID>
__global__ __launch_bounds__(256) void k_occlusion_v2(OccArgs a)
{
    constexpr int MAXC = OccV2Cfg<NCH>::kMaxSurvChunks;
    const BatchView &b = a.b;
    if (batch_aborted(b.status)) return;
    __shared__ float4 s_cand[4][kCap];
    __shared__ uint2 s_seg[4][kSegCap + 4];
    __shared__ float4 s_pts[4][64 * MAXC];

    const uint32_t lane = lane_id();
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t bid = blockIdx.x;
    {
        const uint32_t per = a.n_blocks / 8u;
        if (bid < per * 8u) bid = (bid % 8u) * per + bid / 8u;
    }
    const uint32_t apw = a.atoms_per_wave;
    const uint32_t p_begin = (bid * 4u + w) * apw;
    if (p_begin >= b.n_atoms) return;
    const uint32_t p_end = min(p_begin + apw, b.n_atoms);

    const float probe = b.probe;
    const uint32_t n_points = a.lat.n_points;
    const uint32_t n_fused = a.lat.n_fused;
    const uint32_t n_rem = n_points - n_fused;  // < 16
    const uint32_t n_chunks = (n_points + kWave - 1) / kWave;
    const bool counting = b.neighbor_counts != nullptr;
    const float neg_inf = __int_as_float(0xFF800000);

    // fused-rule points: lanes = points, NCH chunks per group
    float sx[NCH], sy[NCH], sz[NCH];
    unsigned long long invalid[NCH];
    auto load_group = [&](uint32_t ch0) {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const uint32_t pi = (ch0 + c) * kWave + lane;  // lattice arrays are zero padded
            const bool in_range = (ch0 + c) < n_chunks;
            sx[c] = in_range ? a.lat.x[pi] : 0.0f;
            sy[c] = in_range ? a.lat.y[pi] : 0.0f;
            sz[c] = in_range ? a.lat.z[pi] : 0.0f;
            invalid[c] = __ballot(!(in_range && pi < n_fused));
        }
    };
    const bool single_group = n_chunks <= (uint32_t)NCH;
    if (single_group) load_group(0);

    // remainder points: lanes tiled as (point = lane % P, candidate group = lane / P)
    uint32_t rem_shift = 0;
    while ((1u << rem_shift) < n_rem) rem_shift++;
    const uint32_t rem_pt = lane & ((1u << rem_shift) - 1u);
    const uint32_t rem_grp = lane >> rem_shift;
    const uint32_t rem_groups = 64u >> rem_shift;
    float rpx = 0.f, rpy = 0.f, rpz = 0.f;
    if (rem_pt < n_rem) {
        rpx = a.lat.x[n_fused + rem_pt];
        rpy = a.lat.y[n_fused + rem_pt];
        rpz = a.lat.z[n_fused + rem_pt];
    }
    const unsigned long long rem_low = n_rem ? ((1ull << n_rem) - 1ull) : 0ull;

    for (uint32_t p = p_begin; p < p_end; p++) {
        const float4 me = b.sorted_xyzr[p];
        const StructGrid g = b.grids[b.sid[p]];
        const float R = me.w + probe;                    // lib.rs:101
        const float R2 = R * R;                          // lib.rs:102
        const float twoR = 2.0f * R;                     // lib.rs:136
        const float sr = me.w + g.max_r + 2.0f * probe;  // spatial_grid.rs:307
        const float sr2 = sr * sr;                       // spatial_grid.rs:308
        unsigned long long my_id = 0;
        if (HAS_ID) my_id = b.sorted_id[p];

        // -- 1. x-runs of the 5x5x5 block, culled by a lower bound on the distance
        const float fx = (me.x - g.min_x) * g.inv_cell;  // spatial_grid.rs:139-141
        const float fy = (me.y - g.min_y) * g.inv_cell;
        const float fz = (me.z - g.min_z) * g.inv_cell;
        const uint32_t cx = min(f2u_sat(fx), g.dim_x - 1u);
        const uint32_t cy = min(f2u_sat(fy), g.dim_y - 1u);
        const uint32_t cz = min(f2u_sat(fz), g.dim_z - 1u);
        const float tol = 8.0f * 1.1920929e-7f * fmaxf(fmaxf(fabsf(fx), fabsf(fy)), fabsf(fz)) + 1e-5f;
        const float srn = sr * g.inv_cell;
        const float thr = srn * srn * 1.00001f;
        uint32_t run_start = 0, run_len = 0;
        if (lane < 25) {
            const int dy = (int)(lane % 5u) - 2, dz = (int)(lane / 5u) - 2;
            const int yy = (int)cy + dy, zz = (int)cz + dz;
            float gy = dy > 0 ? (float)yy - fy : (dy < 0 ? fy - (float)(yy + 1) : 0.0f);
            float gz = dz > 0 ? (float)zz - fz : (dz < 0 ? fz - (float)(zz + 1) : 0.0f);
            gy = fmaxf(gy - tol, 0.0f);
            gz = fmaxf(gz - tol, 0.0f);
            const float b2 = thr - gy * gy - gz * gz;
            if (yy >= 0 && yy < (int)g.dim_y && zz >= 0 && zz < (int)g.dim_z && b2 >= 0.0f) {
                const float gl = fx - (float)cx, gr = (float)(cx + 1u) - fx;
                const float gl1 = fmaxf(gl - tol, 0.0f);         // to cell cx-1
                const float gl2 = fmaxf(gl + 1.0f - tol, 0.0f);  // to cell cx-2
                const float gr1 = fmaxf(gr - tol, 0.0f);         // to cell cx+1
                const float gr2 = fmaxf(gr + 1.0f - tol, 0.0f);  // to cell cx+2
                const int lo = gl2 * gl2 <= b2 ? 2 : (gl1 * gl1 <= b2 ? 1 : 0);
                const int hi = gr2 * gr2 <= b2 ? 2 : (gr1 * gr1 <= b2 ? 1 : 0);
                const uint32_t x0 = (uint32_t)max((int)cx - lo, 0);
                const uint32_t x1 = min(cx + (uint32_t)hi, g.dim_x - 1u);
                const uint32_t c0 = g.cell_base + x0 + (uint32_t)yy * g.dim_x +
                                    (uint32_t)zz * g.dim_x * g.dim_y;
                run_start = b.cells[c0];
                run_len = b.cells[c0 + (x1 - x0) + 1u] - run_start;
            }
        }
        // 16-atom segments of the runs
        const uint32_t nseg = (run_len + 15u) >> 4;
        const uint32_t seg_incl = wave_incl_scan(nseg);
        const uint32_t n_seg_total = (uint32_t)__builtin_amdgcn_readlane((int)seg_incl, 31);
        const uint32_t seg_excl = seg_incl - nseg;

        float accessible = 0.0f;
        uint32_t k_total = 0;

        for (uint32_t ch0 = 0; ch0 < n_chunks; ch0 += NCH) {
            if (!single_group) load_group(ch0);
            unsigned long long occ[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) occ[c] = invalid[c];
            unsigned long long rem_mask = 0ull;
            const bool do_rem = (ch0 == 0) && (n_rem != 0);

            uint32_t nA = 0, nB = 0;
            bool done = false, flushed = false, counted = false;
            float acc_fused = 0.0f;

            for (uint32_t round_base = 0; round_base < n_seg_total && !done; round_base += kSegCap) {
                const uint32_t n_round = min(n_seg_total - round_base, (uint32_t)kSegCap);
                for (uint32_t j = 0; j < nseg; j++) {
                    const uint32_t idx = seg_excl + j - round_base;  // wraps for earlier rounds
                    if (idx < (uint32_t)kSegCap)
                        s_seg[w][idx] = make_uint2(run_start + 16u * j, min(16u, run_len - 16u * j));
                }
                if (lane < 4) s_seg[w][n_round + lane] = make_uint2(0u, 0u);
                wave_lds_fence();

                for (uint32_t it = 0; it * 4u < n_round && !done; it++) {
                    // -- 2. sweep four segments (16 lanes each)
                    const uint2 sg = s_seg[w][it * 4u + (lane >> 4)];
                    const uint32_t q = sg.x + (lane & 15u);
                    bool accept = false, near = false;
                    float4 cand = make_float4(0.f, 0.f, 0.f, 0.f);
                    if ((lane & 15u) < sg.y) {
                        const float4 o = b.sorted_xyzr[q];
                        const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;  // lib.rs:129-131
                        const float d2 = dx * dx + dy * dy + dz * dz;  // spatial_grid.rs:321 == lib.rs:132
                        accept = (q != p) && (d2 <= sr2);              // spatial_grid.rs:335
                        if (HAS_ID) {
                            if (accept) accept = b.sorted_id[q] != my_id;  // spatial_grid.rs:314
                        }
                        near = accept && d2 < R2;                      // heuristic split only
                        cand = make_float4(dx, dy, dz, o.w);
                    }
                    const unsigned long long mN = __ballot(near);
                    const unsigned long long mF = __ballot(accept && !near);
                    if (accept) {
                        const uint32_t slot = near ? nA + mbcnt64(mN) : (uint32_t)kCap - 1u - (nB + mbcnt64(mF));
                        s_cand[w][slot] = cand;
                    }
                    nA += (uint32_t)__popcll(mN);
                    nB += (uint32_t)__popcll(mF);
                    const bool last = (round_base + kSegCap >= n_seg_total) && ((it + 1u) * 4u >= n_round);
                    if (nA + nB <= (uint32_t)kFlushAt && !last) continue;

                    // ---- flush: decide points against the queued candidates ----
                    const uint32_t K = nA + nB;
                    if (ch0 == 0) k_total += K;
                    wave_lds_fence();
                    // -- 3. prep: limit_j of every queued candidate (lane = candidate)
                    for (uint32_t k0 = 0; k0 < K; k0 += 64u) {
                        const uint32_t k = k0 + lane;
                        if (k < K) {
                            const uint32_t slot = k < nA ? k : (uint32_t)kCap - K + k;
                            const float4 e = s_cand[w][slot];
                            const float d2 = e.x * e.x + e.y * e.y + e.z * e.z;  // lib.rs:132-133
                            const float tj = e.w + probe;                        // spatial_grid.rs:336
                            const float t = tj * tj;                             // spatial_grid.rs:339
                            s_cand[w][slot].w = (t - d2 - R2) / twoR;            // lib.rs:136
                        }
                    }
                    wave_lds_fence();
                    // -- remainder points, scalar rule (lib.rs:185-186,206-207): plain products, `<=`
                    if (do_rem) {
                        unsigned long long m = 0ull;
                        for (uint32_t k0 = 0; k0 < K; k0 += rem_groups) {
                            const uint32_t k = k0 + rem_grp;
                            float4 cd = make_float4(0.f, 0.f, 0.f, neg_inf);
                            if (k < K) cd = s_cand[w][k < nA ? k : (uint32_t)kCap - K + k];
                            m |= __ballot((rpx * cd.x + rpy * cd.y + rpz * cd.z) <= cd.w);
                        }
                        for (uint32_t sh = 1u << rem_shift; sh < 64u; sh <<= 1) m |= m >> sh;
                        rem_mask |= m & rem_low;
                    }
                    // -- 4. phase A: near candidates against all fused-rule points (lane = point)
#pragma unroll 2
                    for (uint32_t k = 0; k < nA; k++) {
                        const float4 cd = s_cand[w][k];
#pragma unroll
                        for (int c = 0; c < NCH; c++) {
                            // lib.rs:143-146: mul_add(sx, vx, mul_add(sy, vy, sz * vz)) < limit
                            const float dot = __builtin_fmaf(sx[c], cd.x, __builtin_fmaf(sy[c], cd.y, sz[c] * cd.z));
                            occ[c] |= __ballot(dot < cd.w);
                        }
                    }
                    // -- 5. phase B: far candidates against the survivors
                    uint32_t S = 0;
#pragma unroll
                    for (int c = 0; c < NCH; c++) S += (uint32_t)__popcll(~occ[c]);
                    const bool fast = last && !flushed && S <= 64u * MAXC;
                    if (fast) {
                        counted = true;
                        if (S == 0u || nB == 0u) {
                            acc_fused = (float)S;
                        } else {
                            // compact the survivors' unit vectors through LDS
                            uint32_t base = 0;
#pragma unroll
                            for (int c = 0; c < NCH; c++) {
                                const unsigned long long sv = ~occ[c];
                                if ((sv >> lane) & 1ull)
                                    s_pts[w][base + mbcnt64(sv)] = make_float4(sx[c], sy[c], sz[c], 0.f);
                                base += (uint32_t)__popcll(sv);
                            }
                            wave_lds_fence();
                            const uint32_t far0 = (uint32_t)kCap - nB;
                            uint32_t n_occluded = 0;
                            if (S <= 64u) {
                                // lanes tiled as (survivor = lane % P, candidate group = lane / P)
                                const uint32_t ps = S <= 8u ? 3u : (S <= 16u ? 4u : (S <= 32u ? 5u : 6u));
                                const uint32_t G = 64u >> ps;
                                const uint32_t pt = lane & ((1u << ps) - 1u);
                                const uint32_t grp = lane >> ps;
                                float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (pt < S) pv = s_pts[w][pt];
                                unsigned long long m = 0ull;
                                uint32_t k0 = 0;
                                for (; k0 + G <= nB; k0 += G) {
                                    const float4 cd = s_cand[w][far0 + k0 + grp];
                                    m |= __ballot(__builtin_fmaf(pv.x, cd.x, __builtin_fmaf(pv.y, cd.y, pv.z * cd.z)) < cd.w);
                                }
                                if (k0 < nB) {
                                    float4 cd = make_float4(0.f, 0.f, 0.f, neg_inf);
                                    if (k0 + grp < nB) cd = s_cand[w][far0 + k0 + grp];
                                    m |= __ballot(__builtin_fmaf(pv.x, cd.x, __builtin_fmaf(pv.y, cd.y, pv.z * cd.z)) < cd.w);
                                }
                                for (uint32_t sh = 1u << ps; sh < 64u; sh <<= 1) m |= m >> sh;
                                const unsigned long long low = S >= 64u ? ~0ull : ((1ull << S) - 1ull);
                                n_occluded = (uint32_t)__popcll(m & low);
                            } else {
                                // several compacted chunks of survivors, far candidates broadcast
                                float px[MAXC], py[MAXC], pz[MAXC];
                                unsigned long long m[MAXC];
#pragma unroll
                                for (int c = 0; c < MAXC; c++) {
                                    const uint32_t i = (uint32_t)c * 64u + lane;
                                    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
                                    if (i < S) pv = s_pts[w][i];
                                    px[c] = pv.x; py[c] = pv.y; pz[c] = pv.z;
                                    m[c] = 0ull;
                                }
                                for (uint32_t k = 0; k < nB; k++) {
                                    const float4 cd = s_cand[w][far0 + k];
#pragma unroll
                                    for (int c = 0; c < MAXC; c++)
                                        m[c] |= __ballot(__builtin_fmaf(px[c], cd.x, __builtin_fmaf(py[c], cd.y, pz[c] * cd.z)) < cd.w);
                                }
#pragma unroll
                                for (int c = 0; c < MAXC; c++) {
                                    const uint32_t left = S > (uint32_t)c * 64u ? S - (uint32_t)c * 64u : 0u;
                                    const unsigned long long low = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
                                    n_occluded += (uint32_t)__popcll(m[c] & low);
                                }
                            }
                            acc_fused = (float)(S - n_occluded);
                        }
                    } else {
                        // generic path (several flushes, or too many survivors): far candidates
                        // broadcast against all points, masks kept exact
                        const uint32_t far0 = (uint32_t)kCap - nB;
                        for (uint32_t k = 0; k < nB; k++) {
                            const float4 cd = s_cand[w][far0 + k];
#pragma unroll
                            for (int c = 0; c < NCH; c++) {
                                const float dot = __builtin_fmaf(sx[c], cd.x, __builtin_fmaf(sy[c], cd.y, sz[c] * cd.z));
                                occ[c] |= __ballot(dot < cd.w);
                            }
                        }
                        flushed = true;
                        if (!counting) {
                            unsigned long long all = ~0ull;
#pragma unroll
                            for (int c = 0; c < NCH; c++) all &= occ[c];
                            done = (all == ~0ull) && (!do_rem || rem_mask == rem_low);  // lib.rs:149-152
                        }
                    }
                    wave_lds_fence();
                    nA = nB = 0;
                }
            }
            if (!counted) {
#pragma unroll
                for (int c = 0; c < NCH; c++) acc_fused += (float)__popcll(~occ[c]);  // lib.rs:156-159
            }
            accessible += acc_fused;
            if (do_rem) accessible += (float)(n_rem - (uint32_t)__popcll(rem_mask));  // lib.rs:215-217
        }

        if (lane == 0) {
            const uint32_t orig = b.sorted_orig[p];
            const float surface_area = (4.0f * 3.14159274101257324219f) * R2;  // 4.0 * PI * r2, lib.rs:220
            const float inv_n = 1.0f / (float)n_points;                        // lib.rs:221
            b.atom_sasa[orig] = surface_area * accessible * inv_n;             // lib.rs:222
            if (counting) b.neighbor_counts[orig] = k_total;
        }
        wave_lds_fence();
    }
}

#include "occlusion_v3.inc"

template <int NCH>
void launch_occ(const OccArgs &a, int version, hipStream_t stream)
{
    const bool id = a.b.id != nullptr;
    if (version == 0) {
        if (id) hipLaunchKernelGGL((k_occlusion_v0<NCH, true>), dim3(a.n_blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_occlusion_v0<NCH, false>), dim3(a.n_blocks), dim3(256), 0, stream, a);
    } else if (version == 3) {
        const OccArgs3 a3 = make_args3(a);
        if (id) hipLaunchKernelGGL((k_occlusion_v3<NCH, true>), dim3(a.n_blocks), dim3(256), 0, stream, a3);
        else hipLaunchKernelGGL((k_occlusion_v3<NCH, false>), dim3(a.n_blocks), dim3(256), 0, stream, a3);
    } else {
        if (id) hipLaunchKernelGGL((k_occlusion_v2<NCH, true>), dim3(a.n_blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_occlusion_v2<NCH, false>), dim3(a.n_blocks), dim3(256), 0, stream, a);
    }
}

}  // namespace

void launch_occlusion(const BatchView &b, const Lattice &lat, const OcclusionTuning &tune,
                      hipStream_t stream)
{
    if (!b.n_atoms) return;
    OccArgs a{b, lat, 0, 1, tune.debug_stop};
    if (tune.kernel_version == 0) {
        a.atoms_per_wave = 1;
    } else if (tune.atoms_per_wave > 0) {
        a.atoms_per_wave = tune.atoms_per_wave;
    } else {
        // keep >= ~8 waves per SIMD in flight on 256 CUs before giving a wave more than one atom
        const uint32_t waves_full = 256u * 4u * 8u * 4u;
        a.atoms_per_wave = max(1u, min(16u, b.n_atoms / waves_full));
    }
    if (tune.kernel_version == 3) a.atoms_per_wave = min(a.atoms_per_wave, (uint32_t)kMaxAtomsPerWave);
    a.n_blocks = cdiv(cdiv(b.n_atoms, a.atoms_per_wave), 4);
    const uint32_t n_chunks = (lat.n_points + kWave - 1) / kWave;
    // v3 walks the chunk groups inside one sweep, two chunks at a time, for any n_points
    if (n_chunks <= 2 || tune.kernel_version >= 3) launch_occ<2>(a, tune.kernel_version, stream);
    else if (n_chunks <= 4) launch_occ<4>(a, tune.kernel_version, stream);
    else launch_occ<16>(a, tune.kernel_version, stream);
}

}  // namespace rsasa
