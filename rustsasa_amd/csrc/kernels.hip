// HIP kernels of the MI355X-native Shrake-Rupley engine (gfx950 / CDNA4 only).
//
// Pipeline for one batch of independent structures (all on one stream):
//   k_init_acc -> k_bounds -> k_grid_params/scan/bases    per-structure cell grids
//   k_sort_window                                         counting sort into cells, in LDS
//   k_zero_cells -> k_cell_hist -> k_scan_* -> k_scatter  the same for structures of 65536 atoms or more
//   k_occlusion                                           candidate gather + point tests
//   k_residue_sums                                        ResidueLevel aggregation
//
// Numerics: every expression that decides a point's fate is evaluated in
// IEEE binary32 in the reference's operation order (this file is compiled
// with -ffp-contract=off; the only fused operations are the explicit
// __builtin_fmaf calls that restate pulp's mul_add_f32s, reference
// src/lib.rs:143-144).  Citations are relative to the reference tree.
#include "device_utils.h"

#include <algorithm>

namespace rsasa {

namespace {

// ------------------------------------------------------ per-structure grid --

__global__ void k_init_acc(StructAcc *acc, uint32_t n_structures, BatchStatus *status, uint32_t ids_needed, uint32_t *ids_seg, uint32_t ids_seg_words)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t i = s; ids_seg && i < 2u * ids_seg_words; i += gridDim.x * blockDim.x) ids_seg[i] = 0u;  // (BatchView::ids_seg)
    if (s == 0) {
        status->overflow = 0;
        status->grid_too_large = 0;
        status->bad_input = 0;
        status->deferred = 0;
        status->total_cells = 0;
        status->tail_cell_begin = 0;
        status->tail_atom_base = 0;
        status->n_windows = 0;
        status->grid_cells = 0;
        status->ids_needed = ids_needed;  // (0 with BatchView::ids_check: k_ids_distinct raises it)
        status->ids_unordered = 0;        // (k_bounds raises it)
    }
    if (s >= n_structures) return;
    const int pinf = f2ord(__int_as_float(0x7F800000)), ninf = f2ord(__int_as_float(0xFF800000));
    StructAcc a;
    a.min_x = a.min_y = a.min_z = pinf;   // spatial_grid.rs:113
    a.max_x = a.max_y = a.max_z = ninf;   // spatial_grid.rs:114
    a.max_r = f2ord(0.0f);                // fold(0.0f32, f32::max), lib.rs:262
    a.n_atoms = 0;
    a.first_atom = 0xFFFFFFFFu;
    a.odd_radii = 0;
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
    bool odd_r = false;  // a radius outside [0, 64] (or NaN), a coordinate beyond 1e8 (or NaN / infinite): see StructGrid::odd_radii
    if (b.ids_check) {
        // BatchView::ids_check: an id that is not larger than its predecessor's in the same structure (host-folded ids
        // stand for themselves: folds that rise are ids that differ).  The verdict is the STRUCTURE's (StructAcc::odd_radii
        // bit 2: its ids are in no order; bit 1: it keeps its ids), the batch's flags only say "some structure".
        bool falls = false;
        for (uint32_t i = seg.begin + threadIdx.x; i < seg.end; i += blockDim.x)
            if (i > seg.begin || seg.continues) falls |= load_id(b.id, b.id32, i) <= load_id(b.id, b.id32, i - 1u);
        if (falls) {
            b.status->ids_unordered = 1u;
            // (no tables in this batch: nobody will look closer this time - the structure keeps its ids; ids_needed counts them)
            const int before = atomicOr(&b.acc[seg.sid].odd_radii, b.ids_tables ? 4 : 6);
            if (!b.ids_tables && (before & 2) == 0) atomicAdd(&b.status->ids_needed, 1u);
        }
    }
    for (uint32_t i = seg.begin + threadIdx.x; i < seg.end; i += blockDim.x) {
        float x = b.x[i], y = b.y[i], z = b.z[i], r = load_radius(b.radius, b.radius8, b.radius_table, i);
        odd_r |= !(r >= 0.0f && r <= 64.0f) | !(fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) <= 1e8f) | (x != x) | (y != y) | (z != z);
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x);
        mny = fminf(mny, y); mxy = fmaxf(mxy, y);
        mnz = fminf(mnz, z); mxz = fmaxf(mxz, z);
        mr = fmaxf(mr, r);
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, d, kWave)); mxx = fmaxf(mxx, __shfl_xor(mxx, d, kWave));
        mny = fminf(mny, __shfl_xor(mny, d, kWave)); mxy = fmaxf(mxy, __shfl_xor(mxy, d, kWave));
        mnz = fminf(mnz, __shfl_xor(mnz, d, kWave)); mxz = fmaxf(mxz, __shfl_xor(mxz, d, kWave));
        mr = fmaxf(mr, __shfl_xor(mr, d, kWave));
    }
    // one set of atomics per workgroup: a structure of a million atoms is 245 workgroups hammering
    // the same seven words (waves without atoms contribute the neutral elements)
    __shared__ int s_red[4][8];
    const uint32_t wv = threadIdx.x / kWave;
    if (lane_id() == 0) {
        s_red[wv][0] = f2ord(mnx); s_red[wv][1] = f2ord(mny); s_red[wv][2] = f2ord(mnz);
        s_red[wv][3] = f2ord(mxx); s_red[wv][4] = f2ord(mxy); s_red[wv][5] = f2ord(mxz);
        s_red[wv][6] = f2ord(mr);
    }
    __syncthreads();
    if (threadIdx.x < 7u && seg.begin < seg.end) {
        const uint32_t k = threadIdx.x;
        StructAcc *a = &b.acc[seg.sid];
        int *dst = k == 0 ? &a->min_x : k == 1 ? &a->min_y : k == 2 ? &a->min_z : k == 3 ? &a->max_x : k == 4 ? &a->max_y : k == 5 ? &a->max_z : &a->max_r;
        if (k < 3u) atomicMin(dst, min(min(s_red[0][k], s_red[1][k]), min(s_red[2][k], s_red[3][k])));
        else atomicMax(dst, max(max(s_red[0][k], s_red[1][k]), max(s_red[2][k], s_red[3][k])));
    }
    if (threadIdx.x == 0) {
        atomicAdd(&b.acc[seg.sid].n_atoms, seg.end - seg.begin);
        atomicMin(&b.acc[seg.sid].first_atom, seg.begin);
    }
    if (ballot64(odd_r) != 0ull && lane_id() == 0) atomicOr(&b.acc[seg.sid].odd_radii, 1);
}

// BatchView::ids_check, second step - for the structures whose ids k_bounds found in no order (hashes: what
// SASAOptions::process passes, options.rs:183): are the ids of the structure all different?  One workgroup per structure
// puts its ids into an open-addressing table in LDS (the entry is the atom's number; an occupied slot is decided on the
// ids themselves - the full 64-bit ids, so the answer is exact whatever the hash does; or the host's 32-bit folds, where
// equal folds count as "maybe equal ids": such a structure keeps its ids, which is always correct).  The first equal pair
// marks THE STRUCTURE as one that keeps its ids (StructGrid::odd_radii bit 1) and raises BatchStatus::ids_needed ("some
// structure does").  Two launches share the structures by size (device_types.h kIdSlotsSmall / kIdSlotsLarge): one
// workgroup per structure with the small table - larger structures are not its business, except that it marks the ones
// too large for either table -, and one per entry of the host's list of larger structures with the large one (a
// workgroup that only finds out that it has nothing to do would still wait for 144 KB of LDS).
template <uint32_t SLOTS, uint32_t THREADS, bool LARGE>
__global__ __launch_bounds__(THREADS) void k_ids_distinct(BatchView b)
{
    const uint32_t s = LARGE ? b.large_sids[blockIdx.x] : blockIdx.x;
    // (the structure's own flags, final since k_bounds / k_grid_params: every thread reads the same words - no workgroup
    // of this launch changes another structure's)
    if ((b.grids[s].odd_radii & 6u) != 4u) return;  // ids that rise, or a structure that keeps its ids already
    const uint32_t n = b.acc[s].n_atoms, a0 = b.acc[s].first_atom;
    if (n < 2u) return;
    if (n > (LARGE ? kIdAtomsLarge : kIdAtomsSmall)) {
        if (!LARGE && n > kIdAtomsLarge && threadIdx.x == 0u) {  // too large for either table: its ids stay in play
            atomicOr(&b.grids[s].odd_radii, 2u);
            atomicAdd(&b.status->ids_needed, 1u);
        }
        return;
    }
    // entries: the atom's number + 1 - 32 bits wide in the small table, 16 in the large one (two per word: a slot is taken
    // by a compare-and-swap of the word that holds it)
    constexpr uint32_t kWords = LARGE ? SLOTS / 2u : SLOTS;
    __shared__ uint32_t s_tab[kWords];
    for (uint32_t k = threadIdx.x; k < kWords; k += THREADS) s_tab[k] = 0u;
    __syncthreads();
    auto take = [&](uint32_t slot, uint32_t value) -> uint32_t {  // 0: the slot was free and is `value`'s now; else who holds it
        if (!LARGE) return atomicCAS(&s_tab[slot], 0u, value);
        uint32_t *w = &s_tab[slot >> 1];
        const uint32_t sh = (slot & 1u) * 16u;
        for (;;) {
            const uint32_t old = *(volatile uint32_t *)w;
            const uint32_t there = (old >> sh) & 0xFFFFu;
            if (there) return there;
            if (atomicCAS(w, old, old | (value << sh)) == old) return 0u;
        }
    };
    bool equal = false;
    // (a thread's next kAhead ids are requested together: one memory round trip per kAhead atoms instead of one per atom; the
    // large table's launch over the real-coordinates batch - 1 200 structures above 4 096 atoms, one workgroup per CU at a time -
    // 173 -> 158 us.  A medium table of 48 KB for the structures of up to 12 288 atoms, three workgroups per CU, was no better:
    // 133 us behind the large table's 100, one after the other on the stream - an insert costs what its probes' id loads cost)
    constexpr uint32_t kAhead = LARGE ? 8u : 4u;
    for (uint32_t i0 = threadIdx.x; i0 < n && !equal; i0 += THREADS * kAhead) {
        uint64_t ahead[kAhead];
#pragma unroll
        for (uint32_t u = 0; u < kAhead; u++) {
            const uint32_t i = i0 + u * THREADS;
            ahead[u] = i < n ? load_id(b.id, b.id32, a0 + i) : 0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < kAhead; u++) {
            const uint32_t i = i0 + u * THREADS;
            if (i >= n || equal) break;
            const uint64_t mine = ahead[u];
            uint32_t h = (uint32_t)(((uint64_t)(fold_id(mine) * 0x9E3779B1u) * SLOTS) >> 32);
            for (;;) {
                const uint32_t there = take(h, i + 1u);
                if (there == 0u) break;
                // (an entry is the number of one of this structure's atoms - anything else would count as an equal pair,
                // never as an address)
                if (there > n || load_id(b.id, b.id32, a0 + there - 1u) == mine) { equal = true; break; }
                h = h + 1u == SLOTS ? 0u : h + 1u;
            }
        }
    }
    if (equal && (atomicOr(&b.grids[s].odd_radii, 2u) & 2u) == 0u) atomicAdd(&b.status->ids_needed, 1u);  // (once per structure)
}

// BatchView::ids_seg: which runs of 64 cell-sorted atoms hold atoms of structures that keep their ids, and which hold
// atoms of structures that do not (one thread per structure; the verdicts are final, the structures placed).
__global__ __launch_bounds__(256) void k_ids_segments(BatchView b)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= b.n_structures) return;
    const StructGrid g = b.grids[s];
    if (!g.n_atoms) return;
    uint32_t *bits = b.ids_seg + ((g.odd_radii & 2u) ? 0u : b.ids_seg_words);
    const uint32_t s0 = g.sorted_base >> 6, s1 = (g.sorted_base + g.n_atoms - 1u) >> 6;
    for (uint32_t w = s0 >> 5; w <= (s1 >> 5) && w < b.ids_seg_words; w++) {
        const uint32_t lo = max(s0, w << 5) & 31u, hi = min(s1, (w << 5) + 31u) & 31u;
        atomicOr(&bits[w], (0xFFFFFFFFu >> (31u - hi)) & (0xFFFFFFFFu << lo));
    }
}

// SpatialGrid::new parameters (spatial_grid.rs:35-44 with cell_size from lib.rs:76).
__device__ __forceinline__ StructGrid make_grid(const StructAcc &a, float probe, bool &bad, bool &too_large)
{
    StructGrid g = {};
    float max_r = ord2f(a.max_r);
    float cell = probe + max_r;                                    // lib.rs:76
    float inv = 1.0f / cell;                                       // spatial_grid.rs:36
    float mn[3] = {ord2f(a.min_x) - cell, ord2f(a.min_y) - cell, ord2f(a.min_z) - cell};
    float mx[3] = {ord2f(a.max_x) + cell, ord2f(a.max_y) + cell, ord2f(a.max_z) + cell};
    uint32_t d[3];
#pragma unroll
    for (int k = 0; k < 3; k++)                                    // spatial_grid.rs:39-43
        d[k] = f2u_sat(ceilf((mx[k] - mn[k]) * inv)) + 1u;
    unsigned long long nc = (unsigned long long)d[0] * d[1] * d[2];
    if (a.n_atoms == 0) {
        // an empty structure has no atoms to place or to look up: one cell, whatever the probe radius
        // (probe 0 would make its cell size 0 - not an error of the batch)
        nc = 1; d[0] = d[1] = d[2] = 1;
        mn[0] = mn[1] = mn[2] = 0.0f;
        inv = 1.0f; cell = 1.0f;
    } else if (!(cell > 0.0f) || !(inv < __int_as_float(0x7F800000)) || !(cell < __int_as_float(0x7F800000))) {
        // (probe + largest radius zero, negative, NaN - or infinite: a cell size the culling arithmetic cannot work with)
        bad = true; nc = 1; d[0] = d[1] = d[2] = 1;
    }
    // (an infinite coordinate saturates an extent and the + 1 wraps it to 0 - where the reference's u32 arithmetic
    // panics: the structure's grid is "too large" too)
    if (nc > 0x7FFFFFFFull || (a.n_atoms != 0 && (d[0] == 0u || d[1] == 0u || d[2] == 0u))) { too_large = true; nc = 1; d[0] = d[1] = d[2] = 1; }
    g.min_x = mn[0]; g.min_y = mn[1]; g.min_z = mn[2];
    g.inv_cell = inv;
    g.dim_x = d[0]; g.dim_y = d[1]; g.dim_z = d[2];
    g.max_r = max_r;
    g.cell_size = cell;
    g.n_cells = (uint32_t)nc;
    g.atom_begin = a.n_atoms ? a.first_atom : 0u;
    g.n_atoms = a.n_atoms;
    // 16-bit LDS counters and positions: structures with fewer than 65536 atoms are binned in LDS,
    // one k_sort_window workgroup per window of kWindowCells cells; the others by the batch-wide kernels
    g.in_lds = a.n_atoms < kLdsMaxAtoms ? 1u : 0u;
    g.odd_radii = ((uint32_t)a.odd_radii & 7u) | (grid_group_shift(a.n_atoms, (uint32_t)nc) << 8);  // (bits 1, 2: the ids' verdict, k_bounds)
    return g;
}

// Grids of all structures plus the exclusive scans that place each structure's cells in the
// batch-wide cell array and its atoms in the cell-sorted arrays.  Structures binned in LDS
// (k_sort_window; 16-bit cell starts) come first in both, in structure order; the others form the
// "tail" that the batch-wide histogram / scan / scatter kernels handle (32-bit cell starts).  The
// LDS-binned structures' windows are numbered by the same scan: GridSums::atoms_s carries the atoms
// in its low and the windows in its high 32 bits (a batch has fewer than 2^32 atoms).  Three small kernels:
//   k_grid_params  (one thread per structure)  grid + per-workgroup sums of (cells, atoms) x (LDS, tail)
//   k_grid_scan    (one workgroup)             exclusive scan of those sums, totals -> BatchStatus
//   k_grid_bases   (one thread per structure)  cell_base / sorted_base of every structure
// inclusive scan of four running sums over a workgroup of NW waves; returns the workgroup totals
template <int NW>
__device__ __forceinline__ GridSums block_scan_sums(GridSums &v, unsigned long long (*part)[4])
{
    const uint32_t l = lane_id(), wv = threadIdx.x / kWave;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned long long t0 = __shfl_up(v.cells_s, d, kWave), t1 = __shfl_up(v.cells_l, d, kWave);
        const unsigned long long t2 = __shfl_up(v.atoms_s, d, kWave), t3 = __shfl_up(v.atoms_l, d, kWave);
        if (l >= (uint32_t)d) { v.cells_s += t0; v.cells_l += t1; v.atoms_s += t2; v.atoms_l += t3; }
    }
    __syncthreads();
    if (l == kWave - 1) { part[wv][0] = v.cells_s; part[wv][1] = v.cells_l; part[wv][2] = v.atoms_s; part[wv][3] = v.atoms_l; }
    __syncthreads();
    unsigned long long off[4] = {0, 0, 0, 0}, tot[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NW; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long x = part[i][k];
            if ((uint32_t)i < wv) off[k] += x;
            tot[k] += x;
        }
    }
    v.cells_s += off[0]; v.cells_l += off[1]; v.atoms_s += off[2]; v.atoms_l += off[3];
    return GridSums{tot[0], tot[1], tot[2], tot[3]};
}

__global__ __launch_bounds__(256) void k_grid_params(BatchView b)
{
    __shared__ unsigned long long part[4][4];
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    GridSums v = {0, 0, 0, 0};
    unsigned long long n_cells = 0;
    if (s < b.n_structures) {
        bool bad = false, too_large = false;
        const StructGrid g = make_grid(b.acc[s], b.probe, bad, too_large);
        b.grids[s] = g;  // cell_base / sorted_base follow in k_grid_bases
        if (bad) b.status->bad_input = 1u;
        if (too_large) b.status->grid_too_large = 1u;
        if (g.in_lds) { v.cells_s = lds_cell_slots(g.n_cells); v.atoms_s = g.n_atoms | ((unsigned long long)grid_windows(g.n_cells) << 32); }
        else { v.cells_l = g.n_cells; v.atoms_l = g.n_atoms; }
        n_cells = g.n_cells;
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) n_cells += __shfl_xor(n_cells, d, kWave);
    if (lane_id() == 0 && n_cells) atomicAdd((unsigned long long *)&b.status->grid_cells, n_cells);
    const GridSums tot = block_scan_sums<4>(v, part);
    if (threadIdx.x == 0) b.grid_sums[blockIdx.x] = tot;
}

template <int NW>
__global__ __launch_bounds__(NW * kWave) void k_grid_scan(BatchView b, uint32_t n_parts)
{
    __shared__ unsigned long long part[NW][4];
    GridSums carry = {0, 0, 0, 0};
    for (uint32_t base = 0; base < n_parts; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        GridSums mine = {0, 0, 0, 0};
        if (i < n_parts) mine = b.grid_sums[i];
        GridSums v = mine;
        const GridSums tot = block_scan_sums<NW>(v, part);
        if (i < n_parts)
            b.grid_sums[i] = GridSums{carry.cells_s + v.cells_s - mine.cells_s, carry.cells_l + v.cells_l - mine.cells_l,
                                      carry.atoms_s + v.atoms_s - mine.atoms_s, carry.atoms_l + v.atoms_l - mine.atoms_l};
        carry.cells_s += tot.cells_s; carry.cells_l += tot.cells_l;
        carry.atoms_s += tot.atoms_s; carry.atoms_l += tot.atoms_l;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // the LDS-binned structures' 16-bit cell starts come first, two per 32-bit entry; the tail
        // starts on a 1024-entry boundary (vector accesses of the scan kernels).  Cell indices are
        // 32-bit: a batch is limited to 2^32 - 16 entries of either kind, more is reported as an overflow.
        const unsigned long long tail_begin = (carry.cells_s / 2ull + 1023ull) & ~1023ull;
        const unsigned long long total = tail_begin + carry.cells_l;
        const unsigned long long n_windows = carry.atoms_s >> 32;
        b.status->total_cells = total;
        b.status->tail_cell_begin = tail_begin;
        b.status->tail_atom_base = (uint32_t)carry.atoms_s;
        b.status->n_windows = (uint32_t)min(n_windows, 0xFFFFFFFFull);
        b.status->overflow = (total > b.cell_capacity || total > 0xFFFFFFF0ull || carry.cells_s > 0xFFFFFFF0ull ||
                              n_windows > b.window_capacity) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_grid_bases(BatchView b)
{
    __shared__ unsigned long long part[4][4];
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ncells = 0, na = 0;
    bool in_lds = false;
    if (s < b.n_structures) {
        ncells = b.grids[s].n_cells;
        na = b.grids[s].n_atoms;
        in_lds = b.grids[s].in_lds != 0u;
    }
    const uint32_t slots = lds_cell_slots(ncells), n_win = grid_windows(ncells);
    const unsigned long long own = na | ((unsigned long long)n_win << 32);
    GridSums v = {in_lds ? slots : 0u, in_lds ? 0u : ncells, in_lds ? own : 0ull, in_lds ? 0u : na};
    (void)block_scan_sums<4>(v, part);
    if (s < b.n_structures) {
        const GridSums base = b.grid_sums[blockIdx.x];
        const unsigned long long cb = in_lds ? base.cells_s + v.cells_s - slots
                                             : b.status->tail_cell_begin + base.cells_l + v.cells_l - ncells;
        const unsigned long long as = base.atoms_s + v.atoms_s - own;  // atoms | windows << 32 before this structure
        b.grids[s].cell_base = (uint32_t)(cb > 0xFFFFFFFFull ? 0xFFFFFFFFull : cb);
        b.grids[s].sorted_base = in_lds ? (uint32_t)as : (uint32_t)(b.status->tail_atom_base + base.atoms_l + v.atoms_l - na);
        if (in_lds && !batch_aborted(b.status)) {
            const uint32_t w0 = (uint32_t)(as >> 32);
            for (uint32_t w = 0; w < n_win; w++) b.windows[w0 + w] = make_uint4(s, w, b.grids[s].atom_begin, na);
        }
    }
}

// Counting sort of ONE window (kWindowCells consecutive cells) of one structure in LDS: 16-bit
// counters (two cells per word), histogram with LDS atomics, in-place exclusive scan, then the
// cell starts and the window's sorted atoms go to global memory.  Replaces k_zero_cells /
// k_cell_hist / k_scan_* / k_scatter for structures with fewer than 65536 atoms: the cell array is
// written once - as 16-bit positions relative to the structure's first sorted atom, half the
// bytes of absolute ones - and never read back.  The windows of a structure are independent
// workgroups (BatchView::windows): each looks at all atoms of the structure, counts the ones in
// earlier windows (they precede its own in the sorted order) and bins its own.
//
// The sorted records leave through the LDS.  A scattered store costs the CU's memory pipeline one
// cycle per lane (64 lines per instruction), five arrays make five of them per atom, and two
// workgroups per CU do not hide that.  So every atom first takes its position from its cell's
// cursor (cell and position stay in registers, kSlots atoms per thread), and once the cursors
// are dead the counter memory stages the records by position: they go out as whole lines.  Atoms
// beyond the registers' share (structures with more than kSlots * 1024 atoms) keep their position
// in memory (rank_of); positions beyond the staging area (a window with more atoms than it
// holds) are stored directly.
//
// SINGLE: a batch of ONE structure whose grid and status the host has already computed (the
// per-structure call, context.cpp run_small_host_batch).  They arrive as kernel arguments - no upload
// precedes the launch, the inputs are read from pinned host memory - and the first workgroup stores
// them where the later kernels look for them.
constexpr uint32_t sort_window_threads(bool single) { return single ? 1024u : (uint32_t)RSASA_SORT_THREADS; }
#ifdef RSASA_SORT_PROF
// throw-away build (tools/sort_prof.py): per-phase time of k_sort_window, 10 ns ticks summed over workgroups
__device__ unsigned long long g_sort_prof[16];
#define SORT_STAMP(k) do { if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
    atomicAdd(&g_sort_prof[k], t_ - t_prev_); t_prev_ = t_; } } while (0)
#else
#define SORT_STAMP(k)
#endif
template <bool SINGLE>
__global__ __launch_bounds__(sort_window_threads(SINGLE), SINGLE ? 4 : (RSASA_SORT_THREADS == 1024 ? 8 : 7)) void k_sort_window(BatchView b, SingleJob single)
{
    constexpr uint32_t kThreads = sort_window_threads(SINGLE);
#ifdef RSASA_SORT_PROF
    unsigned long long t_prev_ = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int kSlots = 8;                              // atoms per thread with position and coordinates in registers
    constexpr uint32_t kStage = kWindowCells * 2u / 32u;   // 32-byte records the counter memory stages
    constexpr uint32_t kPer = ((kWindowCells / 2u + kThreads - 1u) / kThreads) | 1u;  // counter words a thread scans (odd: no bank conflicts)
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[kWindowCells / 2 + 4 + kPer + 3];
    __shared__ uint32_t smem32[16];
    __shared__ uint32_t s_below;
    // A workgroup's time is a chain of memory round trips (two workgroups per CU do not hide them), so the chain is
    // kept short: the work-list entry carries the structure's atom range and is fetched beside the status words, and
    // the coordinates of all slots are requested before the grid parameters have arrived.
    uint32_t s, c0, a0, n_at;
    StructGrid g;
    if (SINGLE) {
        s = 0;
        c0 = blockIdx.x * kWindowCells;  // (one workgroup per window was launched)
        g = single.grid;
        a0 = g.atom_begin; n_at = g.n_atoms;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            b.grids[0] = g;
            *b.status = single.status;
        }
    } else {
        const uint4 job = b.windows[blockIdx.x];  // (the launch covers the list's capacity: in bounds, maybe stale)
        if (batch_aborted(b.status) || blockIdx.x >= b.status->n_windows) return;
        s = job.x;
        c0 = job.y * kWindowCells;
        a0 = job.z; n_at = job.w;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t a1 = a0 + n_at;
    const float *__restrict__ px = b.x, *__restrict__ py = b.y, *__restrict__ pz = b.z;
    auto pr = [&](uint32_t i) { return load_radius(b.radius, b.radius8, b.radius_table, i); };
    const uint32_t *__restrict__ pid32 = b.id32;
    const uint64_t *__restrict__ pid = pid32 ? reinterpret_cast<const uint64_t *>(pid32) : b.id;  // (non-null: the batch has ids)
    uint32_t *__restrict__ rank_of = b.rank_of;  // sorted position of the atoms without a slot
    uint4 *stage = reinterpret_cast<uint4 *>(s_cnt);
    // ---- the structure's first kSlots * 1024 atoms (slot k of a thread: atom a0 + tid + 1024 k) ----
    float x[kSlots], y[kSlots], z[kSlots];
    // SINGLE: the whole record stays in registers (the inputs sit in host memory there: one trip over
    // the link instead of two; a workgroup has the CU to itself and 128 VGPRs)
    float kr[SINGLE ? kSlots : 1];
    uint64_t kid[SINGLE ? kSlots : 1];
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        x[k] = y[k] = z[k] = 0.f;
        if (SINGLE) { kr[k] = 0.f; kid[k] = 0ull; }
        if (kThreads * k < n_at) {
            const uint32_t i = min(a0 + tid + kThreads * k, a1 - 1u);
            x[k] = px[i]; y[k] = py[i]; z[k] = pz[i];
            if (SINGLE) {
                kr[k] = pr(i);
                if (pid) kid[k] = load_id(b.id, pid32, i);
            }
        }
    }
    if (!SINGLE) g = b.grids[s];
    if (!SINGLE && b.ids_check && (g.odd_radii & 2u) == 0u) pid = nullptr;  // (this structure's ids are all different: as good as none)
    const uint32_t dim_xy = g.dim_x * g.dim_y;
    const uint32_t n_cells = min(kWindowCells, g.n_cells - c0), n_words = (n_cells + 1u) >> 1;
    const bool last_window = c0 + n_cells == g.n_cells;
    // (counters, the end marker's word, and - for the scan's whole runs - up to a multiple of kPer words)
    for (uint32_t i = tid; i < ((n_words / kPer + 1u) * kPer + 3u) / 4u; i += kThreads) stage[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) s_below = 0;
    // cell relative to the window; no atom in this slot or another window's: >= n_cells.  Once the positions are
    // known, rpos alone says whether the slot is this window's (kNotMine).
    constexpr uint32_t kNotMine = 0xFFFFFFFFu;
    uint32_t rcell[kSlots], rpos[kSlots];
    uint32_t below = 0;  // atoms in earlier windows (wave-uniform count)
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        uint32_t cx, cy, cz;
        cell_coords(g, x[k], y[k], z[k], cx, cy, cz);
        const bool live = a0 + tid + kThreads * k < a1;
        const uint32_t c = cx + cy * g.dim_x + cz * dim_xy;
        rcell[k] = live ? c - c0 : 0xFFFFFFFFu;
        rpos[k] = kNotMine;
        if (c0) below += (uint32_t)__popcll(ballot64(live && c < c0));
    }
    const uint32_t a_rest = a0 + kThreads * kSlots;  // first atom without a slot
    __syncthreads();
    SORT_STAMP(0);
    // ---- count (spatial_grid.rs:53-62) ----
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        const uint32_t lc = rcell[k];
        if (lc < n_cells) atomicAdd(&s_cnt[lc >> 1], 1u << ((lc & 1u) * 16u));
    }
    for (uint32_t i0 = a_rest + tid; i0 < a1; i0 += 4u * kThreads) {  // four atoms per trip, loads first
        float xx[4], yy[4], zz[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = min(i0 + kThreads * k, a1 - 1u);
            xx[k] = px[i]; yy[k] = py[i]; zz[k] = pz[i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t cx, cy, cz;
            cell_coords(g, xx[k], yy[k], zz[k], cx, cy, cz);
            const bool live = i0 + kThreads * k < a1;
            const uint32_t c = cx + cy * g.dim_x + cz * dim_xy, lc = c - c0;
            if (live && lc < n_cells) atomicAdd(&s_cnt[lc >> 1], 1u << ((lc & 1u) * 16u));
            if (c0) below += (uint32_t)__popcll(ballot64(live && c < c0));
        }
    }
    if (c0 && lane_id() == 0 && below) atomicAdd(&s_below, below);
    __syncthreads();
    SORT_STAMP(1);
    // ---- exclusive scan (spatial_grid.rs:65-68): every thread owns kPer consecutive words (odd stride: no
    // bank conflicts), fetched in one go, sums them, the partial sums are scanned across the workgroup, and the
    // words are rewritten as (start of the even cell | start of the odd one << 16), counted from the
    // structure's first sorted atom: < 65536, it has < 65536 atoms
    const uint32_t placed = s_below;  // the window's first position
    uint32_t total;
    if constexpr (SINGLE) {  // (keeps its registers for the records)
        const uint32_t per = ((n_words + kThreads - 1u) / kThreads) | 1u;
        const uint32_t w0 = min(tid * per, n_words), w1 = min(w0 + per, n_words);
        uint32_t sum = 0;
        for (uint32_t j = w0; j < w1; j++) {
            const uint32_t v = s_cnt[j];
            sum += (v & 0xFFFFu) + (v >> 16);
        }
        uint32_t running = placed + block_incl_scan<kThreads / kWave>(sum, smem32, total) - sum;
        for (uint32_t j = w0; j < w1; j++) {
            const uint32_t v = s_cnt[j];
            const uint32_t lo = v & 0xFFFFu, hi = v >> 16;
            s_cnt[j] = running | ((running + lo) << 16);
            running += lo + hi;
        }
    } else {
        // (whole runs only: the counters were zeroed up to a multiple of kPer words past the end marker's word, and
        // a zero counter behind the last cell takes the window's end as its start - which is what the marker is)
        const uint32_t w0 = tid * kPer;
        const bool mine = w0 <= n_words;
        uint32_t *own = s_cnt + (mine ? w0 : 0u);
        uint32_t cv[kPer];
        uint32_t sum = 0;
        if (mine) {
#pragma unroll
            for (uint32_t j = 0; j < kPer; j++) cv[j] = own[j];
#pragma unroll
            for (uint32_t j = 0; j < kPer; j++) sum += (cv[j] & 0xFFFFu) + (cv[j] >> 16);
        }
        uint32_t running = placed + block_incl_scan<kThreads / kWave>(sum, smem32, total) - sum;
        if (mine) {
#pragma unroll
            for (uint32_t j = 0; j < kPer; j++) {
                const uint32_t lo = cv[j] & 0xFFFFu, hi = cv[j] >> 16;
                own[j] = running | ((running + lo) << 16);
                running += lo + hi;
            }
        }
    }
    // the end marker the last cell's run length is read from: with an odd number of cells it is
    // the upper half of the last word already (an empty cell's start), else the word after
    if (SINGLE && last_window && tid == 0 && (n_cells & 1u) == 0u) s_cnt[n_words] = g.n_atoms;
    __syncthreads();
    SORT_STAMP(2);
    // ---- cell starts: the words as they are, eight cells per store ----
    {
        uint4 *out = reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(b.cells) + g.cell_base + c0);
        const uint32_t n16 = n_cells + (last_window ? 1u : 0u);
        for (uint32_t i = tid; i < (n16 + 7u) / 8u; i += kThreads) out[i] = stage[i];
    }
    __syncthreads();  // the starts turn into the cells' cursors
    SORT_STAMP(3);
    // ---- positions (spatial_grid.rs:70-93): an atom takes the next free position of its cell.
    // The order inside a cell is the order of arrival - the results do not depend on it.
#pragma unroll
    for (int k = 0; k < kSlots; k++) {
        if (!SINGLE) asm volatile("" : "+v"(rcell[k]));  // (word and shift are computed again, not kept from the count pass)
        const uint32_t lc = rcell[k];
        if (lc < n_cells) {
            const uint32_t sh = (lc & 1u) * 16u;
            rpos[k] = (atomicAdd(&s_cnt[lc >> 1], 1u << sh) >> sh) & 0xFFFFu;
        }
    }
    for (uint32_t i0 = a_rest + tid; i0 < a1; i0 += 4u * kThreads) {  // atoms without a slot: position through memory
        float xx[4], yy[4], zz[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = min(i0 + kThreads * k, a1 - 1u);
            xx[k] = px[i]; yy[k] = py[i]; zz[k] = pz[i];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + kThreads * k;
            uint32_t cx, cy, cz;
            cell_coords(g, xx[k], yy[k], zz[k], cx, cy, cz);
            const uint32_t lc = cx + cy * g.dim_x + cz * dim_xy - c0;
            if (i < a1 && lc < n_cells) {
                const uint32_t sh = (lc & 1u) * 16u;
                rank_of[i] = (atomicAdd(&s_cnt[lc >> 1], 1u << sh) >> sh) & 0xFFFFu;
            }
        }
    }
    __syncthreads();  // the cursors are dead: their memory stages the records, by position
    SORT_STAMP(4);
    const uint32_t out0 = g.sorted_base + placed;  // this window's atoms are [out0, out0 + total)
    const uint32_t n_staged = min(total, kStage);
    // 32 bytes per atom: (x, y, z, radius) and (input index, id fold, id)
    constexpr int kChunk = 4;  // slots whose loads are in flight together (64 registers per wave)
    uint32_t tid5 = tid;       // (an index the compiler cannot tie to the first pass: it would keep that pass's
    asm volatile("" : "+v"(tid5));  // 64-bit addresses alive across the whole kernel, in scratch)
#pragma unroll
    for (int k0 = 0; k0 < kSlots; k0 += kChunk) {
        if (kThreads * k0 < n_at) {
            float4 v[kChunk];
            uint64_t id[kChunk];
#pragma unroll
            for (int k = 0; k < kChunk; k++) {
                v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                id[k] = 0ull;
                if (SINGLE) {
                    v[k] = make_float4(x[k0 + k], y[k0 + k], z[k0 + k], kr[k0 + k]);
                    id[k] = kid[k0 + k];
                } else if (rpos[k0 + k] != kNotMine) {
                    const uint32_t i = a0 + tid5 + kThreads * (k0 + k);
                    v[k] = make_float4(px[i], py[i], pz[i], pr(i));
                    if (pid) id[k] = load_id(b.id, pid32, i);
                }
            }
#pragma unroll
            for (int k = 0; k < kChunk; k++) {
                const uint32_t rel = rpos[k0 + k] - placed, i = a0 + tid5 + kThreads * (k0 + k);
                if (rpos[k0 + k] != kNotMine) {
                    if (rel < kStage) {
                        stage[2u * rel] = __builtin_bit_cast(uint4, v[k]);
                        stage[2u * rel + 1u] = make_uint4(i, fold_id(id[k]), (uint32_t)id[k], (uint32_t)(id[k] >> 32));
                    } else {
                        const uint32_t p = g.sorted_base + rpos[k0 + k];
                        b.sorted_xyzr[p] = v[k];
                        b.sorted_orig[p] = i;
                        if (pid) { if (b.sorted_id) b.sorted_id[p] = id[k]; b.sorted_id32[p] = fold_id(id[k]); }
                    }
                }
            }
        }
    }
    for (uint32_t i0 = a_rest + tid; i0 < a1; i0 += 4u * kThreads) {
        float4 v[4];
        uint64_t idr[4];
        uint32_t pos[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = min(i0 + kThreads * k, a1 - 1u);
            pos[k] = rank_of[i];  // (another window's atom: not ours to read, ignored below)
            v[k] = make_float4(px[i], py[i], pz[i], pr(i));
            idr[k] = pid ? load_id(b.id, pid32, i) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + kThreads * k;
            uint32_t cx, cy, cz;
            cell_coords(g, v[k].x, v[k].y, v[k].z, cx, cy, cz);
            const uint32_t lc = cx + cy * g.dim_x + cz * dim_xy - c0;
            if (i < a1 && lc < n_cells) {
                const uint32_t rel = pos[k] - placed;
                if (rel < kStage) {
                    stage[2u * rel] = __builtin_bit_cast(uint4, v[k]);
                    stage[2u * rel + 1u] = make_uint4(i, fold_id(idr[k]), (uint32_t)idr[k], (uint32_t)(idr[k] >> 32));
                } else {
                    const uint32_t p = g.sorted_base + pos[k];
                    b.sorted_xyzr[p] = v[k];
                    b.sorted_orig[p] = i;
                    if (pid) { if (b.sorted_id) b.sorted_id[p] = idr[k]; b.sorted_id32[p] = fold_id(idr[k]); }
                }
            }
        }
    }
    __syncthreads();
    SORT_STAMP(5);
    for (uint32_t j = tid; j < n_staged; j += kThreads) {
        const uint4 q = stage[2u * j + 1u];
        b.sorted_xyzr[out0 + j] = __builtin_bit_cast(float4, stage[2u * j]);
        b.sorted_orig[out0 + j] = q.x;
        if (pid) {
            b.sorted_id32[out0 + j] = q.y;
            if (b.sorted_id) b.sorted_id[out0 + j] = (uint64_t)q.z | ((uint64_t)q.w << 32);
        }
    }
    for (uint32_t j = tid; j < total; j += kThreads) b.sid_sorted[out0 + j] = s;
#ifdef RSASA_SORT_PROF
    __syncthreads();
    SORT_STAMP(6);
    if (threadIdx.x == 0) { atomicAdd(&g_sort_prof[8], 1ull); atomicAdd(&g_sort_prof[9], (unsigned long long)n_cells);
                            atomicAdd(&g_sort_prof[10], (unsigned long long)total); atomicAdd(&g_sort_prof[11], (unsigned long long)g.n_atoms); }
#endif
}
#ifdef RSASA_SORT_PROF
extern "C" __attribute__((visibility("default"))) int rsasa_debug_sort_prof(unsigned long long *out)
{
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_prof), sizeof(zero)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sort_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif

// ---- batch-wide path for the structures of the tail (cells do not fit the LDS) ----

__global__ __launch_bounds__(256) void k_zero_cells(BatchView b)
{
    if (batch_aborted(b.status)) return;
    // tail cells + end sentinel, rounded up to whole 16-byte vectors (the buffer has the slack)
    const uint64_t n4 = (b.status->total_cells + 1 + 3) / 4;
    uint4 *c4 = reinterpret_cast<uint4 *>(b.cells);
    for (uint64_t i = b.status->tail_cell_begin / 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (uint64_t)gridDim.x * blockDim.x)
        c4[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Count atoms per cell (spatial_grid.rs:53-62); the atomic's return value is
// the atom's slot inside its cell, so the scatter needs no second atomic.
constexpr uint32_t kSegmentParts = 4;  // workgroups per bounds segment in the tail's histogram / scatter

// kSegmentParts workgroups per bounds segment (<= kSegmentAtoms atoms of ONE structure): the segments of
// LDS-binned structures - nearly all of them in a batch of proteins - return at once instead of
// reading every atom's structure and grid.
__global__ __launch_bounds__(256) void k_cell_hist(BatchView b)
{
    if (batch_aborted(b.status)) return;
    const Segment seg = b.segments[blockIdx.x / kSegmentParts];
    const StructGrid g = b.grids[seg.sid];
    if (g.in_lds) return;
    const uint32_t part = kSegmentAtoms / kSegmentParts, p0 = seg.begin + (blockIdx.x % kSegmentParts) * part;
    for (uint32_t i = p0 + threadIdx.x; i < min(seg.end, p0 + part); i += blockDim.x) {
        uint32_t cx, cy, cz;
        cell_coords(g, b.x[i], b.y[i], b.z[i], cx, cy, cz);
        const uint32_t cell = g.cell_base + cx + cy * g.dim_x + cz * g.dim_x * g.dim_y;
        b.cell_of[i] = cell;
        b.rank_of[i] = atomicAdd(&b.cells[cell], 1u);
    }
}

// Exclusive prefix sum of the cell counts (spatial_grid.rs:65-68), three
// phases over a fixed grid of kScanBlocks workgroups.
__device__ __forceinline__ void scan_range(const BatchView &b, uint64_t &begin, uint64_t &end)
{
    const uint64_t first = b.status->tail_cell_begin;  // multiple of 1024
    const uint64_t last = b.status->total_cells + 1;   // tail cells + end sentinel
    uint64_t chunk = (last - first + kScanBlocks - 1) / kScanBlocks;
    chunk = (chunk + 1023) & ~uint64_t(1023);
    begin = first + (uint64_t)blockIdx.x * chunk;
    if (begin > last) begin = last;
    end = begin + chunk;
    if (end > last) end = last;
}

__global__ __launch_bounds__(256) void k_scan_reduce(BatchView b)
{
    if (batch_aborted(b.status)) return;
    __shared__ uint32_t smem[4];
    uint64_t begin, end;
    scan_range(b, begin, end);
    // 16-byte loads: `begin` is a multiple of 1024 entries and the tail past `end` was zeroed
    uint32_t sum = 0;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(b.cells);
    for (uint64_t i = begin / 4 + threadIdx.x; i < (end + 3) / 4; i += blockDim.x) {
        const uint4 v = c4[i];
        sum += v.x + v.y + v.z + v.w;
    }
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
    uint32_t running = b.status->tail_atom_base + b.scan_block_sums[blockIdx.x];
    uint4 *c4 = reinterpret_cast<uint4 *>(b.cells);
    for (uint64_t tile = begin; tile < end; tile += 1024) {
        const uint64_t i0 = tile + (uint64_t)threadIdx.x * 4;
        // whole 16-byte vectors: entries past `end` inside the last vector are zero (k_zero_cells)
        const bool live = i0 < end;
        const uint4 v = live ? c4[i0 / 4] : make_uint4(0u, 0u, 0u, 0u);
        uint32_t tsum = v.x + v.y + v.z + v.w, total;
        uint32_t inc = block_incl_scan<4>(tsum, smem, total);
        const uint32_t ex = running + inc - tsum;
        if (live) c4[i0 / 4] = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        running += total;
    }
}

// Scatter into the cell-sorted arrays (spatial_grid.rs:70-93).  Order inside a
// cell is arrival order; results do not depend on it (occlusion is an OR over
// the whole candidate set).
__global__ __launch_bounds__(256) void k_scatter(BatchView b)
{
    if (batch_aborted(b.status)) return;
    const Segment seg = b.segments[blockIdx.x / kSegmentParts];
    const uint32_t s = seg.sid;
    if (b.grids[s].in_lds) return;
    const uint32_t part = kSegmentAtoms / kSegmentParts, p0 = seg.begin + (blockIdx.x % kSegmentParts) * part;
    for (uint32_t i = p0 + threadIdx.x; i < min(seg.end, p0 + part); i += blockDim.x) {
        const uint32_t pos = b.cells[b.cell_of[i]] + b.rank_of[i];
        b.sorted_xyzr[pos] = make_float4(b.x[i], b.y[i], b.z[i], load_radius(b.radius, b.radius8, b.radius_table, i));
        b.sorted_orig[pos] = i;
        b.sid_sorted[pos] = s;
        if (b.sorted_id32 && !(b.ids_check && (b.grids[s].odd_radii & 2u) == 0u)) { const uint64_t v = load_id(b.id, b.id32, i); if (b.sorted_id) b.sorted_id[pos] = v; b.sorted_id32[pos] = fold_id(v); }
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

}  // namespace

// Grid parameters and placement of every structure (everything the two binning routes need).
void launch_grid_prepare(const BatchView &b, hipStream_t stream)
{
    hipLaunchKernelGGL(k_init_acc, dim3(cdiv(b.n_structures > 0 ? b.n_structures : 1, 256)), dim3(256), 0, stream,
                       b.acc, b.n_structures, b.status, b.ids_check ? 0u : 1u, b.ids_check ? b.ids_seg : nullptr, b.ids_seg_words);
    if (b.n_segments)
        hipLaunchKernelGGL(k_bounds, dim3(b.n_segments), dim3(256), 0, stream, b);
    const uint32_t n_parts = cdiv(b.n_structures > 0 ? b.n_structures : 1, 256);
    hipLaunchKernelGGL(k_grid_params, dim3(n_parts), dim3(256), 0, stream, b);
    if (n_parts <= 256) hipLaunchKernelGGL(k_grid_scan<1>, dim3(1), dim3(64), 0, stream, b, n_parts);
    else hipLaunchKernelGGL(k_grid_scan<16>, dim3(1), dim3(1024), 0, stream, b, n_parts);
    hipLaunchKernelGGL(k_grid_bases, dim3(n_parts), dim3(256), 0, stream, b);
    if (b.ids_check && b.ids_tables && b.n_structures) {
        // Both return at once unless k_bounds found ids that do not rise - but their workgroups ask for 32 / 144 KB of LDS
        // even to return, which they only get in the tail of the neighbouring batch's occlusion kernel: they are only
        // launched when the context's last batch had such ids (BatchView::ids_tables), and last of the small kernels, where
        // the binning that follows waits for LDS anyway (in front of the grid kernels they held those up as well).
        hipLaunchKernelGGL((k_ids_distinct<kIdSlotsSmall, 256u, false>), dim3(b.n_structures), dim3(256), 0, stream, b);
        if (b.n_large)
            hipLaunchKernelGGL((k_ids_distinct<kIdSlotsLarge, 1024u, true>), dim3(b.n_large), dim3(1024), 0, stream, b);
    }
    if (b.ids_check && b.ids_seg && b.n_structures) hipLaunchKernelGGL(k_ids_segments, dim3(n_parts), dim3(256), 0, stream, b);
}

// Binning of the structures with fewer than 65536 atoms: one workgroup per window of cells.  The
// work list is written on the device (k_grid_bases); `window_capacity` workgroups are launched and
// the surplus exits.
void launch_sort_lds(const BatchView &b, hipStream_t stream)
{
    if (b.window_capacity) hipLaunchKernelGGL(k_sort_window<false>, dim3(b.window_capacity), dim3(sort_window_threads(false)), 0, stream, b, SingleJob{});
}

// One structure, grid and status from the host (see k_sort_window<true>): one workgroup per window.
void launch_sort_single(const BatchView &b, const SingleJob &job, hipStream_t stream)
{
    const uint32_t n_win = grid_windows(job.grid.n_cells);
    if (n_win) hipLaunchKernelGGL(k_sort_window<true>, dim3(n_win), dim3(sort_window_threads(true)), 0, stream, b, job);
}

// Batch-wide binning of the other structures (the tail).  Independent of launch_sort_lds: the
// context runs it on a second stream, next to the LDS binning and the first occlusion launch.
void launch_sort_tail(const BatchView &b, hipStream_t stream)
{
    hipLaunchKernelGGL(k_zero_cells, dim3(2048), dim3(256), 0, stream, b);
    if (b.n_segments)
        hipLaunchKernelGGL(k_cell_hist, dim3(b.n_segments * kSegmentParts), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(k_scan_reduce, dim3(kScanBlocks), dim3(256), 0, stream, b);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(kScanBlocks), 0, stream, b);
    hipLaunchKernelGGL(k_scan_apply, dim3(kScanBlocks), dim3(256), 0, stream, b);
    if (b.n_segments)
        hipLaunchKernelGGL(k_scatter, dim3(b.n_segments * kSegmentParts), dim3(256), 0, stream, b);
}

// Trajectory frames: xyz is frame-major [n_frames][n_atoms][3] (what MD readers hand over);
// radii, ids and residue offsets are given once and tiled over the frames.  `res_stride` entries
// of res_out per frame: n_res when the residues cover the atoms exactly (a frame's last residue
// ends where the next frame's first begins), n_res + 1 otherwise - the extra entry closes the
// frame's last residue, and the "residue" between it and the next frame's first offset collects the
// uncovered atoms (the caller skips it when copying the sums out).
__global__ __launch_bounds__(256) void k_expand_frames(const float *xyz, const float *radius,
                                                       const uint64_t *id, const uint32_t *res_off,
                                                       uint32_t n_atoms, uint32_t n_frames,
                                                       uint32_t res_stride, float *x, float *y, float *z,
                                                       float *r, uint64_t *id_out, uint32_t *res_out)
{
    const uint32_t n_res = res_stride;
    const uint64_t total = (uint64_t)n_atoms * n_frames;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const uint32_t a = (uint32_t)(i % n_atoms);
        x[i] = xyz[3 * i + 0];
        y[i] = xyz[3 * i + 1];
        z[i] = xyz[3 * i + 2];
        r[i] = radius[a];
        if (id) id_out[i] = id[a];
    }
    const uint64_t total_res = (uint64_t)n_res * n_frames;
    if (res_off && i <= total_res) {
        const uint32_t f = (uint32_t)(i / n_res), k = (uint32_t)(i % n_res);
        res_out[i] = i == total_res ? n_atoms * n_frames : f * n_atoms + res_off[k];
    }
}

void launch_expand_frames(const float *xyz, const float *radius, const uint64_t *id,
                          const uint32_t *res_off, uint32_t n_atoms, uint32_t n_frames, uint32_t res_stride,
                          float *x, float *y, float *z, float *r, uint64_t *id_out,
                          uint32_t *res_out, hipStream_t stream)
{
    const uint64_t n = std::max<uint64_t>((uint64_t)n_atoms * n_frames, (uint64_t)res_stride * n_frames + 1);
    hipLaunchKernelGGL(k_expand_frames, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, xyz,
                       radius, id, res_off, n_atoms, n_frames, res_stride, x, y, z, r, id_out, res_out);
}

// The call combiner's batches (combine.cpp): the callers' atoms sit in pinned host memory as 24-byte records
// (x, y, z, r, id: rsasa_atom_t), the batch's header (status, grids, binning work list) beside them.  One launch reads
// both across the link - every thread one record, coalesced, thousands of reads in flight - and writes the columns and
// the header into device memory: no copy engine involved (a hipMemcpyAsync costs the stream 15 us before the first
// kernel starts; the binning kernel reading the pinned columns itself, one structure per workgroup, took 70 us).
__global__ __launch_bounds__(256) void k_unpack_atoms(const uint2 *__restrict__ rec, uint32_t n, float *__restrict__ x, float *__restrict__ y,
                                                      float *__restrict__ z, float *__restrict__ r, uint64_t *__restrict__ id,
                                                      const uint32_t *__restrict__ hdr_src, uint32_t *__restrict__ hdr_dst, uint32_t hdr_words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hdr_words) hdr_dst[i] = hdr_src[i];
    if (i >= n) return;
    const uint2 a = rec[3u * i], b = rec[3u * i + 1u], c = rec[3u * i + 2u];
    x[i] = __uint_as_float(a.x);
    y[i] = __uint_as_float(a.y);
    z[i] = __uint_as_float(b.x);
    r[i] = __uint_as_float(b.y);
    if (id) id[i] = (uint64_t)c.x | ((uint64_t)c.y << 32);
}

void launch_unpack_atoms(const void *records, uint32_t n_atoms, float *x, float *y, float *z, float *r, uint64_t *id,
                         const void *hdr_src, void *hdr_dst, uint32_t hdr_bytes, hipStream_t stream)
{
    const uint32_t words = hdr_bytes / 4u, threads = std::max(n_atoms, words);
    if (!threads) return;
    hipLaunchKernelGGL(k_unpack_atoms, dim3(cdiv(threads, 256u)), dim3(256), 0, stream, (const uint2 *)records, n_atoms, x, y, z, r, id,
                       (const uint32_t *)hdr_src, (uint32_t *)hdr_dst, words);
}

void launch_residue_sums(const BatchView &b, hipStream_t stream)
{
    if (!b.n_residues || !b.residue_sasa) return;
    hipLaunchKernelGGL(k_residue_sums, dim3(cdiv(b.n_residues, 256)), dim3(256), 0, stream, b);
}

}  // namespace rsasa
