// Microbenchmark: issue costs in REAL shader cycles.  microbench_issue / microbench_mfma2 divide wall time by an
// assumed 2.4 GHz; this one stamps s_memtime (shader clock) and s_memrealtime (100 MHz) inside every wave, so a
// kernel that pulls the clock down (dense v_fma streams do) is not mistaken for a slower pipe.
// W resident waves per SIMD are set through the dynamic LDS size of a 256-thread workgroup (one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_clock microbench_clock.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long c0, c1, r0, r1; };

// MODE 0: FILL x v_fma_f32 (4 independent chains)
// MODE 1: v_mfma_f32_16x16x16_f16 (C = 0) + 2 dependent maxima + FILL x v_fma_f32
// MODE 2: v_mfma_f32_16x16x4_f32  (C = 0) + 2 dependent maxima + FILL x v_fma_f32
// MODE 3: FILL x s_add_u32 / s_xor_b32
// MODE 4: FILL x v_cmp_lt_f32 -> sgpr pair
// MODE 5: FILL/2 x v_fma_f32 + FILL/2 x s_add_u32 interleaved
// MODE 7: v_mfma_f32_16x16x4_4b_f16 (C = 0, four 16x16x4 blocks: 1024 results) + 8 dependent maxima + FILL x v_fma_f32
// MODE 8: v_mfma_f32_32x32x8_f16 (C = 0, 1024 results) + 8 dependent maxima + FILL x v_fma_f32
// MODE 9: FILL x v_pk_mul_f32 / v_pk_add_f32 (two f32 per lane and instruction)
// MODE 6: as 1, but the maxima are taken on the result of the PREVIOUS matrix instruction (software pipelined)
template <int MODE, int FILL>
__global__ __launch_bounds__(256) void k(int iters, Stamp *out, float *sink)
{
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, e = 0.25f;
    unsigned s0 = blockIdx.x, s1 = 12345u;
    unsigned long long m0 = 0, m1 = 0;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    unsigned mx = 0;
    f4 prev = z;
    float2v pa = {a, b}, pb = {b, c}, pc = {c, e};
    if (iters < 0) lds[threadIdx.x] = a;  // (keeps the allocation)
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const h4 ha = __builtin_bit_cast(h4, (float2v){a, b}), hb = __builtin_bit_cast(h4, (float2v){c, e});
            if (MODE == 1) { f4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(d[0])), __float_as_uint(d[1])); mx = max(max(mx, __float_as_uint(d[2])), __float_as_uint(d[3])); }
            if (MODE == 2) { f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(d[0])), __float_as_uint(d[1])); mx = max(max(mx, __float_as_uint(d[2])), __float_as_uint(d[3])); }
            if (MODE == 6) { f4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, z, 0, 0, 0); mx = max(max(mx, __float_as_uint(prev[0])), __float_as_uint(prev[1])); mx = max(max(mx, __float_as_uint(prev[2])), __float_as_uint(prev[3])); prev = d; }
            if (MODE == 7) { const f16v z16 = {0.f}; f16v d = __builtin_amdgcn_mfma_f32_16x16x4f16(ha, hb, z16, 2, 0, 0);
                for (int q = 0; q < 16; q += 2) mx = max(max(mx, __float_as_uint(d[q])), __float_as_uint(d[q + 1])); }
            if (MODE == 8) { const f16v z16 = {0.f}; f16v d = __builtin_amdgcn_mfma_f32_32x32x8f16(ha, hb, z16, 0, 0, 0);
                for (int q = 0; q < 16; q += 2) mx = max(max(mx, __float_as_uint(d[q])), __float_as_uint(d[q + 1])); }
            if (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 6 || MODE == 7 || MODE == 8)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %3\n v_fma_f32 %3, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
            if (MODE == 9)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %0\n v_pk_add_f32 %0, %0, %1" : "+v"(pa), "+v"(pb), "+v"(pc));
            // MODE 10 / 11: the same addition in its 4-byte (VOP2) and 8-byte (VOP3) encoding; MODE 12: 4-byte with a 32-bit
            // literal behind it (8 bytes): is an instruction's cost its issue slot, or the bytes the CU has to fetch?
            if (MODE == 10)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_add_f32_e32 %0, %0, %1\n v_add_f32_e32 %1, %1, %2\n v_add_f32_e32 %2, %2, %0\n v_add_f32_e32 %3, %3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
            if (MODE == 11)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_add_f32_e64 %0, %0, %1\n v_add_f32_e64 %1, %1, %2\n v_add_f32_e64 %2, %2, %0\n v_add_f32_e64 %3, %3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
            if (MODE == 12)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_add_f32_e32 %0, 0x3f8ccccd, %0\n v_add_f32_e32 %1, 0x3f8ccccd, %1\n v_add_f32_e32 %2, 0x3f8ccccd, %2\n v_add_f32_e32 %3, 0x3f8ccccd, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
            if (MODE == 3)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1) :: "scc");
            if (MODE == 4)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %1, %3, %2\n v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %1, %3, %2" : "=s"(m0), "=s"(m1) : "v"(a), "v"(b));
            if (MODE == 5)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n s_add_u32 %3, %3, %4\n v_fma_f32 %1, %1, %2, %0\n s_xor_b32 %4, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+s"(s0), "+s"(s1) :: "scc");
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};
    mx = max(max(mx, __float_as_uint(prev[0])), __float_as_uint(prev[1]));
    if ((float)mx + pa.x + pb.y + pc.x + a + b + c + e + (float)(s0 + s1) + (float)(m0 + m1) == 12345.678f) sink[0] = a;
}

template <int MODE, int FILL>
void run(const char *name, int waves_per_simd, Stamp *d, float *sink)
{
    const int iters = 300;
    const int blocks = 256 * waves_per_simd;
    const unsigned lds = waves_per_simd >= 8 ? 0u : (160u * 1024u / (unsigned)waves_per_simd) - 512u;  // W workgroups per CU
    (void)hipFuncSetAttribute((const void *)k<MODE, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++)  // the last of three back-to-back launches is read (clock settled)
        hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(256), lds, 0, iters, d, sink);
    (void)hipDeviceSynchronize();
    std::vector<Stamp> h(blocks * 4);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (auto &s : h) {
        cyc.push_back((double)(s.c1 - s.c0));
        if (s.r1 > s.r0) ghz.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 0.1);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    const double groups = (double)iters * 8;  // (matrix instruction + FILL fillers) groups per wave
    const double per_group_simd = cyc[cyc.size() / 2] / groups / waves_per_simd;
    printf("%-34s fill %2d waves/SIMD %d: %7.1f cycles per group per wave, %6.2f per group per SIMD, clock %.2f GHz\n", name, FILL,
           waves_per_simd, cyc[cyc.size() / 2] / groups, per_group_simd, ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
}

int main()
{
    Stamp *d; float *sink;
    (void)hipMalloc(&d, sizeof(Stamp) * 256 * 8 * 4);
    (void)hipMalloc(&sink, 64);
    const int ws[] = {1, 2, 4, 7, 8};
    for (int w : ws) run<10, 16>("v_add_f32 (4 bytes) x16", w, d, sink);
    for (int w : ws) run<11, 16>("v_add_f32 (8 bytes, VOP3) x16", w, d, sink);
    for (int w : ws) run<12, 16>("v_add_f32 + literal (8 bytes) x16", w, d, sink);
    for (int w : ws) run<0, 16>("v_fma_f32 x16", w, d, sink);
    for (int w : ws) run<9, 16>("v_pk_mul/add_f32 x16", w, d, sink);
    for (int w : ws) run<3, 16>("s_add/s_xor x16", w, d, sink);
    for (int w : ws) run<4, 16>("v_cmp->sgpr x16", w, d, sink);
    for (int w : ws) run<5, 16>("v_fma x8 + s_add x8", w, d, sink);
    for (int w : ws) { run<1, 4>("mfma16x16x16f16 + 2 max", w, d, sink); run<1, 8>("mfma16x16x16f16 + 2 max", w, d, sink); run<1, 16>("mfma16x16x16f16 + 2 max", w, d, sink); }
    for (int w : ws) { run<6, 8>("mfma16x16x16f16 pipelined max", w, d, sink); }
    for (int w : ws) { run<7, 4>("mfma16x16x4_4b_f16 + 8 max", w, d, sink); run<7, 8>("mfma16x16x4_4b_f16 + 8 max", w, d, sink); run<7, 16>("mfma16x16x4_4b_f16 + 8 max", w, d, sink); }
    for (int w : ws) { run<8, 4>("mfma32x32x8f16 + 8 max", w, d, sink); run<8, 8>("mfma32x32x8f16 + 8 max", w, d, sink); run<8, 16>("mfma32x32x8f16 + 8 max", w, d, sink); }
    for (int w : ws) { run<2, 8>("mfma16x16x4f32 + 2 max", w, d, sink); run<2, 16>("mfma16x16x4f32 + 2 max", w, d, sink); }
    return 0;
}
