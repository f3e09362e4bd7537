// Microbenchmark: issue cost of VALU / SALU / LDS-broadcast / ballot instructions on gfx950
// at full occupancy (8 waves per SIMD).  Build: hipcc --offload-arch=gfx950 -O3 -o mb microbench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float *out)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    unsigned s0 = blockIdx.x, s1 = 12345u;
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(a, b, c, a);
    __syncthreads();
    unsigned long long acc = 0, m0 = 0, m1 = 0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {a, b}, pb = {b, c}, pc = {c, a};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 64 dependent-free-ish VALU
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %0, %0, %2, %1" : "+v"(a), "+v"(b), "+v"(c));)
        } else if (MODE == 1) {  // 64 SALU
            REP16(asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1) :: "scc");)
        } else if (MODE == 2) {  // 32 VALU + 32 SALU interleaved
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n s_add_u32 %3, %3, %4\n v_fma_f32 %1, %1, %2, %0\n s_xor_b32 %4, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+s"(s0), "+s"(s1) :: "scc");)
        } else if (MODE == 3) {  // 16 x (LDS broadcast read b128 + 3 VALU + v_cmp->sgpr + s_or)
            REP16({ float4 v = lds[(i + s0) & 255]; float d = __builtin_fmaf(a, v.x, __builtin_fmaf(b, v.y, c * v.z)); acc |= __ballot(d < v.w); s0++; })
        } else if (MODE == 4) {  // 64 VALU + 16 SALU
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n s_add_u32 %3, %3, %4\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %0, %0, %2, %1" : "+v"(a), "+v"(b), "+v"(c), "+s"(s0), "+s"(s1) :: "scc");)
        } else if (MODE == 7) {  // 16 x LDS broadcast read b128 only
            REP16({ float4 v = lds[(i + s0) & 255]; asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); s0++; })
        } else if (MODE == 8) {  // 16 x LDS per-lane read b32
            REP16({ float v = ((float *)lds)[(threadIdx.x + i + s0) & 1023]; asm volatile("" :: "v"(v)); s0++; })
        } else if (MODE == 9) {  // 16 x LDS per-lane write b128
            REP16({ lds[(threadIdx.x + s0) & 255] = make_float4(a, b, c, a); s0++; asm volatile("" ::: "memory"); })
        } else if (MODE == 10) {  // 64 v_pk_fma_f32
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %0\n v_pk_fma_f32 %2, %2, %0, %1\n v_pk_fma_f32 %0, %0, %2, %1" : "+v"(pa), "+v"(pb), "+v"(pc));)
        } else if (MODE == 11) {  // 64 v_cmp_lt_f32 to sgpr pair
            REP16(asm volatile("v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %1, %3, %2\n v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %1, %3, %2" : "=s"(m0), "=s"(m1) : "v"(a), "v"(b));)
        } else if (MODE == 5) {  // 64 v_readlane
            REP16(asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %2, 5\n v_readlane_b32 %0, %2, 7\n v_readlane_b32 %1, %2, 9" : "+s"(s0), "+s"(s1) : "v"(a));)
        }
    }
    if (pa.x + pb.y + pc.x == 4242.f) out[1] = 1.f;
    if (m0 + m1 == 77ull) out[2] = 1.f;
    if (a + b + c == 12345.678f || s0 + s1 == 7u || acc == 3ull) out[0] = a + s0 + (float)acc;
}

template <int MODE>
void run(const char *name, int per_iter, float *d)
{
    const int iters = 500, blocks = 256 * 8;  // 8 blocks of 256 threads per CU = 8 waves/SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, 10, d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr_per_simd = (double)blocks * 4 / 1024 * iters * per_iter;
    printf("%-44s %8.3f ms  -> %.2f ns per wave-instr-group per SIMD (%.2f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
}

int main()
{
    float *d; hipMalloc(&d, 64);
    run<0>("VALU v_fma x64", 64, d);
    run<1>("SALU x64", 64, d);
    run<2>("VALU x32 + SALU x32 (count 64)", 64, d);
    run<4>("VALU x64 + SALU x16 (count VALU=64)", 64, d);
    run<3>("LDS bcast b128 + 4 VALU + ballot (count 16)", 16, d);
    run<5>("v_readlane x64", 64, d);
    run<10>("v_pk_fma_f32 x64", 64, d);
    run<11>("v_cmp_lt_f32 -> sgpr x64", 64, d);
    run<7>("LDS bcast read b128 (count 16)", 16, d);
    run<8>("LDS per-lane read b32 (count 16)", 16, d);
    run<9>("LDS per-lane write b128 (count 16)", 16, d);
    return 0;
}
