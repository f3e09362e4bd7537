// Microbenchmark: cost of a 64-lane gather instruction from an L2-resident table on gfx950, as a
// function of the bytes per lane (4 / 8 / 16) and the number of distinct 128-byte lines one
// instruction touches (1 .. 64) -- the access pattern of the occlusion kernel's sweep.
// Build: hipcc --offload-arch=gfx950 -O3 -o mbg microbench_gather.hip ; run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T>
__global__ __launch_bounds__(256) void k(const T *__restrict__ table, uint32_t n_lines, uint32_t lines_per_instr,
                                         int iters, float *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t per_line = 64u / lines_per_instr;           // lanes sharing a line
    const uint32_t grp = lane / per_line, sub = lane % per_line;
    const uint32_t elems_per_line = 128u / sizeof(T);
    uint32_t seed = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u + grp * 40503u;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            seed = seed * 1664525u + 1013904223u;
            const uint32_t line = (seed >> 8) % n_lines;
            const T v = table[line * elems_per_line + (sub % elems_per_line)];
            acc += *reinterpret_cast<const float *>(&v);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <typename T>
void run(const char *name, const void *table, uint32_t n_lines, float *d)
{
    for (uint32_t lines : {1u, 8u, 16u, 32u, 64u}) {
        const int iters = 200, blocks = 256 * 8;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, (const T *)table, n_lines, lines, 5, d);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, (const T *)table, n_lines, lines, iters, d);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
        printf("%-10s lines/instr %2u : %7.2f ns per gather instruction per CU\n", name, lines, ms * 1e6 / instr_per_cu);
    }
}

int main()
{
    const uint32_t n_lines = 16384;  // 2 MB: stays in every XCD's L2, far larger than the 32 KB L1
    void *table; hipMalloc(&table, (size_t)n_lines * 128);
    hipMemset(table, 0, (size_t)n_lines * 128);
    float *d; hipMalloc(&d, 64);
    run<float>("4 B/lane", table, n_lines, d);
    run<float2>("8 B/lane", table, n_lines, d);
    run<float4>("16 B/lane", table, n_lines, d);
    return 0;
}
