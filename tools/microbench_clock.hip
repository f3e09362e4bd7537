// Microbenchmark: what an instruction costs a SIMD, in REAL shader cycles and with OBSERVED residency.
//
// Every wave stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its instruction stream and reads
// HW_REG_HW_ID (wave slot, SIMD, CU, shader array, shader engine) and HW_REG_XCC_ID: the host groups the waves by the
// SIMD they really ran on.  Per SIMD:
//     price = (latest c1 - earliest c0 of the waves that ran there) / (instructions those waves issued)
// and, as a cross-check that is not diluted by the launch's ramp, the same over the window in which ALL of the SIMD's
// waves were running (latest c0 .. earliest c1, instructions prorated).  The residency printed is what was seen -
// waves per SIMD, and how many of them overlapped in time - not what the launch geometry was meant to give
// (round 4's version divided a wave's time by an ASSUMED number of residents and published 1.3 cycles per v_fma_f32,
// which is 1.54 x the chip's datasheet rate).  A price that implies more than 64 FLOP / clk / SIMD for v_fma_f32 is
// wrong by construction: the program says so itself.
//
// W resident waves per SIMD are requested through the dynamic LDS size of a 256-thread workgroup (one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench_clock microbench_clock.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long c0, c1, r0, r1; unsigned hw_id, xcc_id, pad0, pad1; };

// MODE 0: FILL x v_fma_f32 (a chain through four registers)        MODE 13: FILL x v_fma_f32, four INDEPENDENT chains
// MODE 1: v_mfma_f32_16x16x16_f16 (C = 0) + 2 dependent maxima + FILL x v_fma_f32
// MODE 2: v_mfma_f32_16x16x4_f32  (C = 0) + 2 dependent maxima + FILL x v_fma_f32
// MODE 3: FILL x s_add_u32 / s_xor_b32
// MODE 4: FILL x v_cmp_lt_f32 -> sgpr pair
// MODE 5: FILL/2 x v_fma_f32 + FILL/2 x s_add_u32 interleaved
// MODE 7: v_mfma_f32_16x16x4_4b_f16 (C = 0, four 16x16x4 blocks: 1024 results) + 8 dependent maxima + FILL x v_fma_f32
// MODE 8: v_mfma_f32_32x32x8_f16 (C = 0, 1024 results) + 8 dependent maxima + FILL x v_fma_f32
// MODE 9: FILL x v_pk_mul_f32 / v_pk_add_f32 (two f32 per lane and instruction)
// MODE 6: as 1, but the maxima are taken on the result of the PREVIOUS matrix instruction (software pipelined)
// MODE 14: FILL x ds_read_b128 (all lanes one address: broadcast) + wait      MODE 15: FILL x ds_write_b128 + wait
// MODE 16: FILL/4 x (v_cmp -> s_and_saveexec -> v_mbcnt x2 -> s_or exec -> s_bcnt1): the sweep's compaction idiom
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
    f4 ld = z;
    if (iters < 0) lds[threadIdx.x] = a;  // (keeps the allocation)
    const unsigned hw_id = (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    const unsigned xcc_id = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID, bits 3:0
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
            if (MODE == 13)
                for (int f = 0; f < FILL; f += 4)
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
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
            if (MODE == 14) {
                for (int f = 0; f < FILL; f++) {
                    f4 t;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(f * 16)));
                    ld += t;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (MODE == 15) {
                for (int f = 0; f < FILL; f++)
                    asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(threadIdx.x * 16)), "v"(ld) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (MODE == 16)
                for (int f = 0; f < FILL; f += 4) {
                    unsigned long long sv;
                    unsigned r;
                    asm volatile("v_cmp_lt_f32 vcc, %2, %3\n s_and_saveexec_b64 %0, vcc\n v_mbcnt_lo_u32_b32 %1, vcc_lo, 0\n v_mbcnt_hi_u32_b32 %1, vcc_hi, %1\n"
                                 " s_or_b64 exec, exec, %0\n s_bcnt1_i32_b64 %4, vcc\n s_add_u32 %5, %5, %4"
                                 : "=&s"(sv), "=&v"(r), "+v"(a), "+v"(b), "=&s"(s1), "+s"(s0) :: "vcc", "scc");
                    mx += r;
                }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1, hw_id, xcc_id, 0u, 0u};
    mx = max(max(mx, __float_as_uint(prev[0])), __float_as_uint(prev[1]));
    if ((float)mx + pa.x + pb.y + pc.x + a + b + c + e + ld[0] + ld[3] + (float)(s0 + s1) + (float)(m0 + m1) == 12345.678f) sink[0] = a;
}

// instructions one group issues (what the price divides by): the FILL instructions + the matrix instruction and its maxima
static int group_insts(int mode, int fill)
{
    switch (mode) {
    case 1: case 2: case 6: return fill + 1 + 4;
    case 7: case 8: return fill + 1 + 16;
    case 14: case 15: return fill + 1;
    case 16: return fill / 4 * 7;
    default: return fill;
    }
}

template <int MODE, int FILL>
void run(const char *name, int waves_per_simd, Stamp *d, float *sink)
{
    const int iters = 300;
    const int blocks = 256 * waves_per_simd;
    const unsigned lds = waves_per_simd >= 8 ? 4608u : (160u * 1024u / (unsigned)waves_per_simd) - 512u;  // W workgroups per CU (8: the wave limit; 4.5 KB for the LDS modes)
    (void)hipFuncSetAttribute((const void *)k<MODE, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++)  // the last of three back-to-back launches is read (clock settled)
        hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(256), lds, 0, iters, d, sink);
    (void)hipDeviceSynchronize();
    std::vector<Stamp> h(blocks * 4);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    // HW_ID (gfx9 layout): wave slot [3:0], SIMD [5:4], pipe [7:6], CU [11:8], shader array [12], shader engine [15:13]
    std::map<unsigned, std::vector<const Stamp *>> by_simd;
    for (auto &s : h) by_simd[(s.xcc_id & 15u) << 16 | (s.hw_id & 0xFF30u)].push_back(&s);
    const double insts_per_wave = (double)iters * 8 * group_insts(MODE, FILL);
    std::vector<double> span_price, overlap_price, resident, overlapping, ghz, wave_price;
    for (auto &kv : by_simd) {
        unsigned long long first = ~0ull, last = 0, c0_max = 0, c1_min = ~0ull;
        for (auto *s : kv.second) {
            first = std::min(first, s->c0); last = std::max(last, s->c1);
            c0_max = std::max(c0_max, s->c0); c1_min = std::min(c1_min, s->c1);
        }
        span_price.push_back((double)(last - first) / (insts_per_wave * kv.second.size()));
        resident.push_back((double)kv.second.size());
        // the window in which every wave of this SIMD ran: instructions prorated by each wave's own rate
        int n_over = 0;
        double in_window = 0;
        if (c1_min > c0_max)
            for (auto *s : kv.second) { in_window += insts_per_wave * (double)(c1_min - c0_max) / (double)(s->c1 - s->c0); n_over++; }
        if (n_over) overlap_price.push_back((double)(c1_min - c0_max) / in_window);
        overlapping.push_back(n_over);
        for (auto *s : kv.second) {
            wave_price.push_back((double)(s->c1 - s->c0) / insts_per_wave);
            if (s->r1 > s->r0) ghz.push_back((double)(s->c1 - s->c0) / (double)(s->r1 - s->r0) * 0.1);
        }
    }
    auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto lo = [](std::vector<double> &v) { return v.empty() ? 0.0 : *std::min_element(v.begin(), v.end()); };
    auto hi = [](std::vector<double> &v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
    const double p = med(span_price), po = med(overlap_price);
    const bool fma = MODE == 0 || MODE == 13;
    printf("%-30s fill %2d asked %d/SIMD: SIMDs seen %4zu, waves per SIMD min %.0f med %.0f max %.0f (all overlapping in time on %.0f %% of SIMDs); "
           "cycles per instruction per SIMD: span %.3f, all-resident window %.3f (per group of %d: %.1f); per wave %.2f; clock %.2f GHz%s\n",
           name, FILL, waves_per_simd, by_simd.size(), lo(resident), med(resident), hi(resident),
           100.0 * (double)std::count_if(overlapping.begin(), overlapping.end(), [&](double x) { return x > 0; }) / (double)overlapping.size(),
           p, po, group_insts(MODE, FILL), po * group_insts(MODE, FILL), med(wave_price), med(ghz),
           fma && po > 0 && po < 1.99 ? "   <-- IMPLIES MORE THAN 64 FLOP/clk/SIMD: CHECK" : "");
}

int main()
{
    Stamp *d; float *sink;
    (void)hipMalloc(&d, sizeof(Stamp) * 256 * 8 * 4);
    (void)hipMalloc(&sink, 64);
    const int ws[] = {1, 2, 4, 7, 8};
    for (int w : ws) run<0, 16>("v_fma_f32 x16 (one chain)", w, d, sink);
    for (int w : ws) run<13, 16>("v_fma_f32 x16 (4 chains)", w, d, sink);
    for (int w : ws) run<10, 16>("v_add_f32 (4 bytes) x16", w, d, sink);
    for (int w : ws) run<11, 16>("v_add_f32 (8 bytes, VOP3) x16", w, d, sink);
    for (int w : ws) run<12, 16>("v_add_f32 + literal x16", w, d, sink);
    for (int w : ws) run<9, 16>("v_pk_mul/add_f32 x16", w, d, sink);
    for (int w : ws) run<3, 16>("s_add/s_xor x16", w, d, sink);
    for (int w : ws) run<4, 16>("v_cmp->sgpr x16", w, d, sink);
    for (int w : ws) run<5, 16>("v_fma x8 + s_add x8", w, d, sink);
    for (int w : ws) run<16, 16>("compaction idiom (7 instr) x4", w, d, sink);
    for (int w : ws) run<14, 8>("ds_read_b128 x8 + wait", w, d, sink);
    for (int w : ws) run<15, 8>("ds_write_b128 x8 + wait", w, d, sink);
    for (int w : ws) { run<1, 4>("mfma16x16x16f16 + 4 max", w, d, sink); run<1, 16>("mfma16x16x16f16 + 4 max", w, d, sink); }
    for (int w : ws) { run<7, 4>("mfma16x16x4_4b_f16 + 16 max", w, d, sink); run<7, 16>("mfma16x16x4_4b_f16 + 16 max", w, d, sink); }
    for (int w : ws) { run<8, 4>("mfma32x32x8f16 + 16 max", w, d, sink); run<8, 16>("mfma32x32x8f16 + 16 max", w, d, sink); }
    for (int w : ws) { run<2, 8>("mfma16x16x4f32 + 4 max", w, d, sink); run<2, 16>("mfma16x16x4f32 + 4 max", w, d, sink); }
    return 0;
}
