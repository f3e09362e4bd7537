// Device-side helpers shared by the engine's kernels (gfx950 only).
#pragma once
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
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

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
// value and the workgroup total.  `smem` holds NW words.  Every wave scans the NW wave totals
// itself (lane = wave): a handful of registers, where a loop over the totals keeps NW values and
// NW lane masks alive around the caller's loops.
template <int NW>
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t *smem, uint32_t &total)
{
    static_assert(NW <= kWave, "one lane per wave");
    const uint32_t l = lane_id();
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const uint32_t inc = wave_incl_scan(v);
    __syncthreads();
    if (l == kWave - 1) smem[w] = inc;
    __syncthreads();
    const uint32_t s = l < (uint32_t)NW ? smem[l] : 0u;
    const uint32_t sc = wave_incl_scan(s);
    total = (uint32_t)__builtin_amdgcn_readlane((int)sc, kWave - 1);
    return inc + (uint32_t)__builtin_amdgcn_readlane((int)(sc - s), (int)w);
}

__device__ __forceinline__ bool batch_aborted(const BatchStatus *st)
{
    return (st->overflow | st->grid_too_large) != 0;
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


// Entry `idx` of the batch-wide cell array (StructGrid::cell_base + a cell of the structure): the
// cell-sorted position of the cell's first atom - relative to the structure's first sorted atom
// (16-bit entries) for LDS-binned structures, absolute (32-bit entries) for the others.
__device__ __forceinline__ uint32_t load_cell_start(const uint32_t *cells, uint32_t idx, bool rel16)
{
    return rel16 ? (uint32_t)reinterpret_cast<const uint16_t *>(cells)[idx] : cells[idx];
}

// Two entries at once, both loads in flight together (one branch on the entry width instead of one per load,
// which also serialised them).
__device__ __forceinline__ void load_cell_start2(const uint32_t *cells, uint32_t idx0, uint32_t idx1, bool rel16,
                                                 uint32_t &v0, uint32_t &v1)
{
    if (rel16) {
        const uint16_t *c16 = reinterpret_cast<const uint16_t *>(cells);
        const uint16_t a = c16[idx0], b = c16[idx1];
        v0 = a;
        v1 = b;
    } else {
        const uint32_t a = cells[idx0], b = cells[idx1];
        v0 = a;
        v1 = b;
    }
}

// 64-bit atom id -> 32 bits.  Only used as a filter: ids whose folds differ are different; equal
// folds are decided on the full ids (the general kernel).
__device__ __forceinline__ uint32_t fold_id(uint64_t id) { return (uint32_t)id ^ ((uint32_t)(id >> 32) * 0x9E3779B1u); }

// The id of input atom i as the binning kernels see it: a host-folded id stands for itself (fold_id of a value
// below 2^32 is that value).
__device__ __forceinline__ uint64_t load_id(const uint64_t *id, const uint32_t *id32, uint32_t i)
{
    return id32 ? (uint64_t)id32[i] : id[i];
}

// The radius of input atom i: its f32, or the table entry its one-byte code names (the same bits: the host built
// the table from the batch's own values).
__device__ __forceinline__ float load_radius(const float *radius, const uint8_t *radius8, const float *table, uint32_t i)
{
    return radius8 ? table[radius8[i]] : radius[i];
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace
}  // namespace rsasa
