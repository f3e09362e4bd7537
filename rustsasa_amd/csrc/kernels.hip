// HIP kernels of the MI355X-native Shrake-Rupley engine (gfx950 / CDNA4 only).
//
// Pipeline for one batch of independent structures (all on one stream):
//   k_init_acc -> k_bounds -> k_finalize_grids            per-structure cell grids
//   k_zero_cells -> k_cell_hist -> k_scan_* -> k_scatter  counting sort into cells
//   k_occlusion                                           candidate gather + point tests
//   k_residue_sums                                        ResidueLevel aggregation
//
// Numerics: every expression that decides a point's fate is evaluated in
// IEEE binary32 in the reference's operation order (this file is compiled
// with -ffp-contract=off; the only fused operations are the explicit
// __builtin_fmaf calls that restate pulp's mul_add_f32s, reference
// src/lib.rs:143-144).  Citations are relative to the reference tree.
#include "device_types.h"

namespace rsasa {

namespace {

constexpr int kWave = 64;

// ---------------------------------------------------------------- helpers --

__device__ __forceinline__ int f2ord(float f)
{
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int o)
{
    return __int_as_float(o >= 0 ? o : o ^ 0x7FFFFFFF);
}

// Rust `f as u32`: saturating, NaN -> 0 (spatial_grid.rs:40-42,139-141).
__device__ __forceinline__ uint32_t f2u_sat(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

// Orders this wave's LDS writes before its later LDS reads (same wave only).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T>
__device__ __forceinline__ T wave_bcast(T v, int src_lane)
{
    return __shfl(v, src_lane, kWave);
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v)
{
    const uint32_t l = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T t = __shfl_up(v, d, kWave);
        if (l >= (uint32_t)d) v += t;
    }
    return v;
}

// Inclusive scan over a workgroup of kWave * NW threads; returns the inclusive
// value and the workgroup total.  `smem` holds NW words.
template <int NW, typename T>
__device__ __forceinline__ T block_incl_scan(T v, T *smem, T &total)
{
    const uint32_t l = lane_id(), w = threadIdx.x / kWave;
    T inc = wave_incl_scan(v);
    __syncthreads();
    if (l == kWave - 1) smem[w] = inc;
    __syncthreads();
    T off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        T s = smem[i];
        if ((uint32_t)i < w) off += s;
        tot += s;
    }
    total = tot;
    return inc + off;
}

// ------------------------------------------------------ per-structure grid --

__global__ void k_init_acc(StructAcc *acc, uint32_t n_structures, BatchStatus *status)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) {
        status->overflow = 0;
        status->grid_too_large = 0;
        status->bad_input = 0;
        status->total_cells = 0;
    }
    if (s >= n_structures) return;
    const int pinf = f2ord(__int_as_float(0x7F800000)), ninf = f2ord(__int_as_float(0xFF800000));
    StructAcc a;
    a.min_x = a.min_y = a.min_z = pinf;   // spatial_grid.rs:113
    a.max_x = a.max_y = a.max_z = ninf;   // spatial_grid.rs:114
    a.max_r = f2ord(0.0f);                // fold(0.0f32, f32::max), lib.rs:262
    a.pad = 0;
    acc[s] = a;
}

// calculate_bounds (spatial_grid.rs:108-122) + max radius (lib.rs:259-262):
// one workgroup per <=4096-atom slice of a structure, integer atomics on
// order-preserving images so the result is exact and order independent.
__global__ __launch_bounds__(256) void k_bounds(BatchView b)
{
    const Segment seg = b.segments[blockIdx.x];
    float mnx = __int_as_float(0x7F800000), mny = mnx, mnz = mnx;
    float mxx = __int_as_float(0xFF800000), mxy = mxx, mxz = mxx;
    float mr = 0.0f;
    for (uint32_t i = seg.begin + threadIdx.x; i < seg.end; i += blockDim.x) {
        float x = b.x[i], y = b.y[i], z = b.z[i], r = b.radius[i];
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x);
        mny = fminf(mny, y); mxy = fmaxf(mxy, y);
        mnz = fminf(mnz, z); mxz = fmaxf(mxz, z);
        mr = fmaxf(mr, r);
        b.sid[i] = seg.sid;
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, d, kWave)); mxx = fmaxf(mxx, __shfl_xor(mxx, d, kWave));
        mny = fminf(mny, __shfl_xor(mny, d, kWave)); mxy = fmaxf(mxy, __shfl_xor(mxy, d, kWave));
        mnz = fminf(mnz, __shfl_xor(mnz, d, kWave)); mxz = fmaxf(mxz, __shfl_xor(mxz, d, kWave));
        mr = fmaxf(mr, __shfl_xor(mr, d, kWave));
    }
    if (lane_id() == 0 && seg.begin + (threadIdx.x & ~(kWave - 1)) < seg.end) {
        StructAcc *a = &b.acc[seg.sid];
        atomicMin(&a->min_x, f2ord(mnx)); atomicMax(&a->max_x, f2ord(mxx));
        atomicMin(&a->min_y, f2ord(mny)); atomicMax(&a->max_y, f2ord(mxy));
        atomicMin(&a->min_z, f2ord(mnz)); atomicMax(&a->max_z, f2ord(mxz));
        atomicMax(&a->max_r, f2ord(mr));
    }
}

// SpatialGrid::new parameters (spatial_grid.rs:35-44 with cell_size from
// lib.rs:76) for every structure, plus the exclusive scan of the cell counts
// that places each structure's cells in the batch-wide cell array.
__global__ __launch_bounds__(1024) void k_finalize_grids(BatchView b)
{
    __shared__ unsigned long long smem[16];
    unsigned long long carry = 0;
    bool too_large = false, bad = false;
    for (uint32_t base = 0; base < b.n_structures; base += blockDim.x) {
        uint32_t s = base + threadIdx.x;
        uint32_t ncells = 0;
        StructGrid g = {};
        if (s < b.n_structures) {
            StructAcc a = b.acc[s];
            float max_r = ord2f(a.max_r);
            float cell = b.probe + max_r;                                  // lib.rs:76
            float inv = 1.0f / cell;                                       // spatial_grid.rs:36
            float mn[3] = {ord2f(a.min_x) - cell, ord2f(a.min_y) - cell, ord2f(a.min_z) - cell};
            float mx[3] = {ord2f(a.max_x) + cell, ord2f(a.max_y) + cell, ord2f(a.max_z) + cell};
            uint32_t d[3];
#pragma unroll
            for (int k = 0; k < 3; k++)                                    // spatial_grid.rs:39-43
                d[k] = f2u_sat(ceilf((mx[k] - mn[k]) * inv)) + 1u;
            unsigned long long nc = (unsigned long long)d[0] * d[1] * d[2];
            if (!(cell > 0.0f) || !(inv < __int_as_float(0x7F800000))) { bad = true; nc = 1; d[0] = d[1] = d[2] = 1; }
            if (nc > 0x7FFFFFFFull) { too_large = true; nc = 1; d[0] = d[1] = d[2] = 1; }
            ncells = (uint32_t)nc;
            g.min_x = mn[0]; g.min_y = mn[1]; g.min_z = mn[2];
            g.inv_cell = inv;
            g.dim_x = d[0]; g.dim_y = d[1]; g.dim_z = d[2];
            g.max_r = max_r;
            g.cell_size = cell;
            g.n_cells = ncells;
        }
        // 64-bit scan: cell indices are 32-bit, so a batch is limited to 2^32 - 2 cells and
        // anything beyond that is reported as an overflow of the workspace capacity.
        unsigned long long total;
        unsigned long long inc = block_incl_scan<16>((unsigned long long)ncells, smem, total);
        if (s < b.n_structures) {
            unsigned long long cb = carry + (inc - ncells);
            g.cell_base = (uint32_t)(cb > 0xFFFFFFFFull ? 0xFFFFFFFFull : cb);
            b.grids[s] = g;
        }
        carry += total;
        __syncthreads();
    }
    too_large = __syncthreads_or(too_large);
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        b.status->total_cells = carry;
        b.status->grid_too_large = too_large ? 1u : 0u;
        b.status->bad_input = bad ? 1u : 0u;
        b.status->overflow = (carry > b.cell_capacity) ? 1u : 0u;
    }
}

__device__ __forceinline__ bool batch_aborted(const BatchStatus *st)
{
    return (st->overflow | st->grid_too_large) != 0;
}

__global__ __launch_bounds__(256) void k_zero_cells(BatchView b)
{
    if (batch_aborted(b.status)) return;
    const uint64_t n = b.status->total_cells + 1;  // + end sentinel
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        b.cells[i] = 0u;
}

// get_cell_index_static (spatial_grid.rs:133-143).  The clamps only matter for
// non-finite input and keep the index inside the structure's cells.
__device__ __forceinline__ void cell_coords(const StructGrid &g, float x, float y, float z,
                                            uint32_t &cx, uint32_t &cy, uint32_t &cz)
{
    cx = min(f2u_sat((x - g.min_x) * g.inv_cell), g.dim_x - 1u);
    cy = min(f2u_sat((y - g.min_y) * g.inv_cell), g.dim_y - 1u);
    cz = min(f2u_sat((z - g.min_z) * g.inv_cell), g.dim_z - 1u);
}

// Count atoms per cell (spatial_grid.rs:53-62); the atomic's return value is
// the atom's slot inside its cell, so the scatter needs no second atomic.
__global__ __launch_bounds__(256) void k_cell_hist(BatchView b)
{
    if (batch_aborted(b.status)) return;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_atoms) return;
    const StructGrid g = b.grids[b.sid[i]];
    uint32_t cx, cy, cz;
    cell_coords(g, b.x[i], b.y[i], b.z[i], cx, cy, cz);
    uint32_t cell = g.cell_base + cx + cy * g.dim_x + cz * g.dim_x * g.dim_y;
    b.cell_of[i] = cell;
    b.rank_of[i] = atomicAdd(&b.cells[cell], 1u);
}

// Exclusive prefix sum of the cell counts (spatial_grid.rs:65-68), three
// phases over a fixed grid of kScanBlocks workgroups.
__device__ __forceinline__ void scan_range(const BatchView &b, uint64_t &begin, uint64_t &end)
{
    const uint64_t n = b.status->total_cells + 1;
    uint64_t chunk = (n + kScanBlocks - 1) / kScanBlocks;
    chunk = (chunk + 1023) & ~uint64_t(1023);
    begin = (uint64_t)blockIdx.x * chunk;
    if (begin > n) begin = n;
    end = begin + chunk;
    if (end > n) end = n;
}

__global__ __launch_bounds__(256) void k_scan_reduce(BatchView b)
{
    if (batch_aborted(b.status)) return;
    __shared__ uint32_t smem[4];
    uint64_t begin, end;
    scan_range(b, begin, end);
    uint32_t sum = 0;
    for (uint64_t i = begin + threadIdx.x; i < end; i += blockDim.x) sum += b.cells[i];
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) sum += __shfl_xor(sum, d, kWave);
    if (lane_id() == 0) smem[threadIdx.x / kWave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) b.scan_block_sums[blockIdx.x] = smem[0] + smem[1] + smem[2] + smem[3];
}

__global__ __launch_bounds__(kScanBlocks) void k_scan_block_sums(BatchView b)
{
    if (batch_aborted(b.status)) return;
    __shared__ uint32_t smem[16];
    uint32_t v = b.scan_block_sums[threadIdx.x], total;
    uint32_t inc = block_incl_scan<16>(v, smem, total);
    b.scan_block_sums[threadIdx.x] = inc - v;
}

__global__ __launch_bounds__(256) void k_scan_apply(BatchView b)
{
    if (batch_aborted(b.status)) return;
    __shared__ uint32_t smem[4];
    uint64_t begin, end;
    scan_range(b, begin, end);
    uint32_t running = b.scan_block_sums[blockIdx.x];
    for (uint64_t tile = begin; tile < end; tile += 1024) {
        uint64_t i0 = tile + (uint64_t)threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < end) ? b.cells[i0 + k] : 0u;
        uint32_t tsum = v[0] + v[1] + v[2] + v[3], total;
        uint32_t inc = block_incl_scan<4>(tsum, smem, total);
        uint32_t ex = running + inc - tsum;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < end) b.cells[i0 + k] = ex;
            ex += v[k];
        }
        running += total;
    }
}

// Scatter into the cell-sorted arrays (spatial_grid.rs:70-93).  Order inside a
// cell is arrival order; results do not depend on it (occlusion is an OR over
// the whole candidate set).
__global__ __launch_bounds__(256) void k_scatter(BatchView b)
{
    if (batch_aborted(b.status)) return;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n_atoms) return;
    uint32_t pos = b.cells[b.cell_of[i]] + b.rank_of[i];
    b.sorted_xyzr[pos] = make_float4(b.x[i], b.y[i], b.z[i], b.radius[i]);
    b.sorted_orig[pos] = i;
    if (b.id) b.sorted_id[pos] = b.id[i];
}

// ---------------------------------------------------------------- occlusion --
//
// One wavefront per atom, atoms taken in cell-sorted order.
//   1. lanes 0..24 fetch the 25 x-runs of cells that make up the 5x5x5 block
//      around the atom's cell (search_extent = 2, spatial_grid.rs:47: the ratio
//      max_search / cell_size is exactly 2 in f32);
//   2. the runs are flattened and swept 64 atoms at a time: distance test with
//      the reference's candidate rule d^2 <= (r_i + max_r + 2p)^2
//      (spatial_grid.rs:307-308,335) and id rule (:314); each accepted lane
//      computes its neighbour's (v, limit) (lib.rs:128-136) and appends it to
//      the wave's LDS list;
//   3. lanes become sphere points: every candidate is broadcast from LDS and
//      tested against NCH chunks of 64 points (lib.rs:143-147), with the
//      reference's remainder rule for the last n_points mod W points
//      (lib.rs:185-186); surviving points are counted with ballot/popcount.

constexpr int kCandCap = 192;    // LDS candidate slots per wave
constexpr int kCandFlush = 128;  // flush once more than this many are queued

struct OccArgs {
    BatchView b;
    Lattice lat;
    uint32_t n_blocks;  // launched workgroups (for the XCD swizzle)
};

template <int NCH, bool HAS_ID>
__global__ __launch_bounds__(256) void k_occlusion(OccArgs a)
{
    const BatchView &b = a.b;
    if (batch_aborted(b.status)) return;
    __shared__ float4 s_cand[4][kCandCap];
    __shared__ uint32_t s_run_excl[4][32];
    __shared__ uint32_t s_run_start[4][32];

    const uint32_t lane = lane_id();
    const uint32_t w = threadIdx.x / kWave;
    // XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs, so give
    // each XCD a contiguous range of cell-sorted atoms (its L2 then holds only
    // its own structures).
    uint32_t bid = blockIdx.x;
    {
        const uint32_t per = a.n_blocks / 8u;
        if (bid < per * 8u) bid = (bid % 8u) * per + bid / 8u;
    }
    const uint32_t p = bid * 4u + w;
    if (p >= b.n_atoms) return;

    const float probe = b.probe;
    const float4 me = b.sorted_xyzr[p];
    const StructGrid g = b.grids[b.sid[p]];
    const float R = me.w + probe;                       // lib.rs:101
    const float R2 = R * R;                             // lib.rs:102
    const float twoR = 2.0f * R;                        // lib.rs:136
    const float sr = me.w + g.max_r + 2.0f * probe;     // spatial_grid.rs:307
    const float sr2 = sr * sr;                          // spatial_grid.rs:308
    unsigned long long my_id = 0;
    if (HAS_ID) my_id = b.sorted_id[p];

    // -- 1. the 25 x-runs of the 5x5x5 cell block
    uint32_t cx, cy, cz;
    cell_coords(g, me.x, me.y, me.z, cx, cy, cz);
    uint32_t run_start = 0, run_len = 0;
    if (lane < 25) {
        const int yy = (int)cy + (int)(lane % 5u) - 2;
        const int zz = (int)cz + (int)(lane / 5u) - 2;
        if (yy >= 0 && yy < (int)g.dim_y && zz >= 0 && zz < (int)g.dim_z) {
            const uint32_t x0 = cx >= 2u ? cx - 2u : 0u;
            const uint32_t x1 = min(cx + 2u, g.dim_x - 1u);
            const uint32_t c0 = g.cell_base + x0 + (uint32_t)yy * g.dim_x +
                                (uint32_t)zz * g.dim_x * g.dim_y;
            run_start = b.cells[c0];
            run_len = b.cells[c0 + (x1 - x0) + 1u] - run_start;
        }
    }
    const uint32_t run_incl = wave_incl_scan(run_len);
    const uint32_t total = wave_bcast(run_incl, 31);
    if (lane < 32) {
        s_run_excl[w][lane] = lane < 25 ? run_incl - run_len : 0xFFFFFFFFu;
        s_run_start[w][lane] = run_start;
    }
    wave_lds_fence();

    const uint32_t n_chunks = (a.lat.n_points + kWave - 1) / kWave;
    float accessible = 0.0f;
    uint32_t k_total = 0;

    for (uint32_t ch0 = 0; ch0 < n_chunks; ch0 += NCH) {
        // sphere points of this group of chunks; lanes past n_points start occluded
        float sx[NCH], sy[NCH], sz[NCH];
        unsigned long long occ[NCH], rem[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const uint32_t pi = (ch0 + c) * kWave + lane;  // lattice arrays are zero padded
            const bool in_range = (ch0 + c) < n_chunks;
            sx[c] = in_range ? a.lat.x[pi] : 0.0f;
            sy[c] = in_range ? a.lat.y[pi] : 0.0f;
            sz[c] = in_range ? a.lat.z[pi] : 0.0f;
            occ[c] = __ballot(!(in_range && pi < a.lat.n_points));
            rem[c] = __ballot(in_range && pi >= a.lat.n_fused && pi < a.lat.n_points);
        }

        uint32_t count = 0;
        bool all_occluded = false;
        for (uint32_t base = 0; base < total && !all_occluded; base += kWave) {
            // -- 2. sweep 64 atoms of the block
            const uint32_t f = base + lane;
            bool accept = false;
            float4 cand = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < total) {
                uint32_t lo = 0;
#pragma unroll
                for (int step = 16; step > 0; step >>= 1)
                    if (s_run_excl[w][lo + step] <= f) lo += step;
                const uint32_t q = s_run_start[w][lo] + (f - s_run_excl[w][lo]);
                const float4 o = b.sorted_xyzr[q];
                const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;  // lib.rs:129-131
                const float d2 = dx * dx + dy * dy + dz * dz;  // spatial_grid.rs:321 == lib.rs:132
                accept = (q != p) && (d2 <= sr2);              // spatial_grid.rs:335
                if (HAS_ID) {
                    if (accept) accept = b.sorted_id[q] != my_id;  // spatial_grid.rs:314
                }
                const float tj = o.w + probe;                  // spatial_grid.rs:336
                const float t = tj * tj;
                cand = make_float4(dx, dy, dz, (t - d2 - R2) / twoR);  // lib.rs:136
            }
            const unsigned long long m = __ballot(accept);
            if (accept) {
                const uint32_t slot = count + __builtin_amdgcn_mbcnt_hi(
                                                  (uint32_t)(m >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                s_cand[w][slot] = cand;
            }
            count += (uint32_t)__popcll(m);
            const bool last = base + kWave >= total;
            if (count <= kCandFlush && !last) continue;

            // -- 3. point tests over the queued candidates
            wave_lds_fence();
            if (ch0 == 0) k_total += count;
            for (uint32_t k = 0; k < count; k++) {
                const float4 cd = s_cand[w][k];
                bool every = true;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    // lib.rs:143-146: mul_add(sx, vx, mul_add(sy, vy, sz * vz)) < limit
                    const float dot = __builtin_fmaf(sx[c], cd.x, __builtin_fmaf(sy[c], cd.y, sz[c] * cd.z));
                    bool hit = dot < cd.w;
                    if (rem[c] != 0ull) {
                        // lib.rs:185-186,206-207: plain products, `<=`
                        const float dotu = sx[c] * cd.x + sy[c] * cd.y + sz[c] * cd.z;
                        const bool is_rem = (rem[c] >> lane) & 1ull;
                        hit = is_rem ? (dotu <= cd.w) : hit;
                    }
                    occ[c] |= __ballot(hit);
                    every = every && (occ[c] == ~0ull);
                }
                if (every) { all_occluded = true; break; }     // lib.rs:149-152
            }
            wave_lds_fence();
            count = 0;
        }
        if (all_occluded && ch0 == 0) k_total = 0xFFFFFFFFu;  // swept only partly: recount below
#pragma unroll
        for (int c = 0; c < NCH; c++) accessible += (float)__popcll(~occ[c]);  // lib.rs:156-159
    }

    if (b.neighbor_counts && k_total == 0xFFFFFFFFu) {
        // the early exit skipped part of the sweep; count the candidates without staging them
        k_total = 0;
        for (uint32_t base = 0; base < total; base += kWave) {
            const uint32_t f = base + lane;
            bool accept = false;
            if (f < total) {
                uint32_t lo = 0;
#pragma unroll
                for (int step = 16; step > 0; step >>= 1)
                    if (s_run_excl[w][lo + step] <= f) lo += step;
                const uint32_t q = s_run_start[w][lo] + (f - s_run_excl[w][lo]);
                const float4 o = b.sorted_xyzr[q];
                const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
                const float d2 = dx * dx + dy * dy + dz * dz;
                accept = (q != p) && (d2 <= sr2);
                if (HAS_ID) {
                    if (accept) accept = b.sorted_id[q] != my_id;
                }
            }
            k_total += (uint32_t)__popcll(__ballot(accept));
        }
    }

    if (lane == 0) {
        const uint32_t orig = b.sorted_orig[p];
        const float surface_area = (4.0f * 3.14159274101257324219f) * R2;  // 4.0 * PI * r2, lib.rs:220
        const float inv_n = 1.0f / (float)a.lat.n_points;      // lib.rs:221
        b.atom_sasa[orig] = surface_area * accessible * inv_n; // lib.rs:222
        if (b.neighbor_counts) b.neighbor_counts[orig] = k_total;
    }
}

// ResidueLevel value: strictly sequential f32 sum of the residue's atoms in
// input order (options.rs:209-216, utils.rs:14-22).  One thread per residue.
__global__ __launch_bounds__(256) void k_residue_sums(BatchView b)
{
    if (batch_aborted(b.status)) return;
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= b.n_residues) return;
    float total = 0.0f;
    for (uint32_t i = b.residue_offsets[k], e = b.residue_offsets[k + 1]; i < e; i++)
        total += b.atom_sasa[i];
    b.residue_sasa[k] = total;
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace

void launch_grid_build(const BatchView &b, hipStream_t stream)
{
    hipLaunchKernelGGL(k_init_acc, dim3(cdiv(b.n_structures > 0 ? b.n_structures : 1, 256)), dim3(256), 0, stream,
                       b.acc, b.n_structures, b.status);
    if (b.n_segments)
        hipLaunchKernelGGL(k_bounds, dim3(b.n_segments), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(k_finalize_grids, dim3(1), dim3(1024), 0, stream, b);
    hipLaunchKernelGGL(k_zero_cells, dim3(2048), dim3(256), 0, stream, b);
    if (b.n_atoms)
        hipLaunchKernelGGL(k_cell_hist, dim3(cdiv(b.n_atoms, 256)), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(k_scan_reduce, dim3(kScanBlocks), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(kScanBlocks), 0, stream, b);
    hipLaunchKernelGGL(k_scan_apply, dim3(kScanBlocks), dim3(256), 0, stream, b);
    if (b.n_atoms)
        hipLaunchKernelGGL(k_scatter, dim3(cdiv(b.n_atoms, 256)), dim3(256), 0, stream, b);
}

template <int NCH>
static void launch_occ(const OccArgs &a, hipStream_t stream)
{
    if (a.b.id)
        hipLaunchKernelGGL((k_occlusion<NCH, true>), dim3(a.n_blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((k_occlusion<NCH, false>), dim3(a.n_blocks), dim3(256), 0, stream, a);
}

void launch_occlusion(const BatchView &b, const Lattice &lat, hipStream_t stream)
{
    if (!b.n_atoms) return;
    OccArgs a{b, lat, cdiv(b.n_atoms, 4)};
    const uint32_t n_chunks = (lat.n_points + kWave - 1) / kWave;
    if (n_chunks <= 2) launch_occ<2>(a, stream);
    else if (n_chunks <= 4) launch_occ<4>(a, stream);
    else launch_occ<16>(a, stream);
}

void launch_residue_sums(const BatchView &b, hipStream_t stream)
{
    if (!b.n_residues || !b.residue_sasa) return;
    hipLaunchKernelGGL(k_residue_sums, dim3(cdiv(b.n_residues, 256)), dim3(256), 0, stream, b);
}

}  // namespace rsasa
