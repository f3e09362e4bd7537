// Microbenchmark: what the matrix instructions a filter could use cost per SIMD on gfx950, alone and
// beside vector instructions of the same wave / of other waves (8 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_mfma2 microbench_mfma2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

// MODE: which instruction; FILL: independent v_fma_f32 per matrix instruction in the same wave
template <int MODE, int FILL>
__global__ __launch_bounds__(256) void k(int iters, float *out)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, e = 0.25f;
    f4 z = {0.f, 0.f, 0.f, 0.f};
    f16v z16 = {0.f};
    f4 acc = z;
    f16v acc16 = z16;
    unsigned mx = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            // operands change every iteration (no hoisting): the f16 ones are bit patterns of the running floats
            const h4 ha = __builtin_bit_cast(h4, (float2v){a, b}), hb = __builtin_bit_cast(h4, (float2v){c, e});
            const h8 ha8 = __builtin_bit_cast(h8, (f4){a, b, c, e});
            if (MODE == 1) { f4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(d[0])), __float_as_uint(d[1])); mx = max(max(mx, __float_as_uint(d[2])), __float_as_uint(d[3])); }
            if (MODE == 2) { f4 d = __builtin_amdgcn_mfma_f32_4x4x4f16(ha, hb, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(d[0])), __float_as_uint(d[1])); mx = max(max(mx, __float_as_uint(d[2])), __float_as_uint(d[3])); }
            if (MODE == 3) { f16v d = __builtin_amdgcn_mfma_f32_32x32x8f16(ha, hb, z16, 0, 0, 0);
                for (int q = 0; q < 16; q += 2) mx = max(max(mx, __float_as_uint(d[q])), __float_as_uint(d[q + 1])); }
            if (MODE == 4) { f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha8, ha8, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(d[0])), __float_as_uint(d[1])); mx = max(max(mx, __float_as_uint(d[2])), __float_as_uint(d[3])); }
            if (MODE == 5) { f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(d[0])), __float_as_uint(d[1])); mx = max(max(mx, __float_as_uint(d[2])), __float_as_uint(d[3])); }
            if (MODE == 6) { f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(d[0])), __float_as_uint(d[1])); mx = max(max(mx, __float_as_uint(d[2])), __float_as_uint(d[3])); }
            if (MODE == 7) { acc = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, acc, 0, 0, 0); }
            if (MODE == 8) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }
            for (int f = 0; f < FILL; f += 4)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %3\n v_fma_f32 %3, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
        }
    }
    acc[0] += (float)mx;
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int i = 0; i < 16; i++) s += acc16[i];
    if (s + a + b + c + e == 12345.678f) out[0] = s;
}

template <int MODE, int FILL>
void run(const char *name, float *d)
{
    const int iters = 400, blocks = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(256), 0, 0, 10, d);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(256), 0, 0, iters, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double groups_per_simd = (double)blocks * 4 / 1024 * iters * 8;  // (matrix instruction + FILL fillers) per SIMD
    printf("%-40s + %2d v_fma: %8.3f ms -> %6.2f ns = %6.1f cycles @2.4 per group per SIMD\n", name, FILL, ms, ms * 1e6 / groups_per_simd,
           ms * 1e6 / groups_per_simd * 2.4);
}

int main()
{
    float *d; (void)hipMalloc(&d, 64);
    run<0, 4>("no matrix instruction", d);
    run<0, 8>("no matrix instruction", d);
    run<0, 16>("no matrix instruction", d);
    run<1, 0>("16x16x16_f16 C=0 + 2 max3", d); run<1, 4>("16x16x16_f16 C=0 + 2 max3", d); run<1, 8>("16x16x16_f16 C=0 + 2 max3", d); run<1, 16>("16x16x16_f16 C=0 + 2 max3", d);
    run<2, 0>("4x4x4_16B_f16 C=0 + 2 max3", d); run<2, 4>("4x4x4_16B_f16 C=0 + 2 max3", d); run<2, 8>("4x4x4_16B_f16 C=0 + 2 max3", d); run<2, 16>("4x4x4_16B_f16 C=0 + 2 max3", d);
    run<3, 0>("32x32x8_f16 C=0 + 8 max3", d); run<3, 8>("32x32x8_f16 C=0 + 8 max3", d); run<3, 16>("32x32x8_f16 C=0 + 8 max3", d);
    run<4, 0>("16x16x32_f16 C=0 + 2 max3", d); run<4, 8>("16x16x32_f16 C=0 + 2 max3", d);
    run<5, 0>("16x16x4_f32 C=0 + 2 max3", d); run<5, 4>("16x16x4_f32 C=0 + 2 max3", d); run<5, 8>("16x16x4_f32 C=0 + 2 max3", d); run<5, 16>("16x16x4_f32 C=0 + 2 max3", d);
    run<6, 0>("4x4x1_16B_f32 C=0 + 2 max3", d); run<6, 8>("4x4x1_16B_f32 C=0 + 2 max3", d);
    run<7, 0>("16x16x16_f16 chained acc", d); run<7, 8>("16x16x16_f16 chained acc", d);
    run<8, 0>("16x16x4_f32 chained acc", d); run<8, 8>("16x16x4_f32 chained acc", d);
    return 0;
}
