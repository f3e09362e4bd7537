// v_mfma_f32_16x16x4_4b_f16 (four independent 16x16x4 products per instruction) as phase A's filter instruction:
// operand / result layout with the A block broadcast (cbsz:2), and its results against v_mfma_f32_16x16x16_f16 on
// the same operands (the instruction the filter used before) and against an f32 fma chain on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -o check_mfma4b check_mfma4b.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__global__ void k(const h4 *a, const h4 *b, float *d4b, float *d16)
{
    const int l = threadIdx.x;
    const f16v z = {0.f};
    const f4 z4 = {0.f, 0.f, 0.f, 0.f};
    // four blocks: the candidates of lanes 0..15 (block 0, broadcast) against the points of lanes 16 b + c
    const f16v d = __builtin_amdgcn_mfma_f32_16x16x4f16(a[l], b[l], z, 2, 0, 0);
    for (int i = 0; i < 16; i++) d4b[l * 16 + i] = d[i];
    // the old form, one point tile at a time: candidates in lanes 0..15 (k = 0..3), zeros elsewhere
    const h4 hz = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const h4 av = l < 16 ? a[l] : hz;
    for (int t = 0; t < 4; t++) {
        const f4 e = __builtin_amdgcn_mfma_f32_16x16x16f16(av, b[16 * t + (l & 15)], z4, 0, 0, 0);
        for (int i = 0; i < 4; i++) d16[l * 16 + 4 * t + i] = e[i];
    }
}

int main()
{
    std::vector<h4> a(64), b(64);
    h4 *da, *db;
    float *d4, *d16;
    (void)hipMalloc(&da, 64 * sizeof(h4)); (void)hipMalloc(&db, 64 * sizeof(h4));
    (void)hipMalloc(&d4, 1024 * 4); (void)hipMalloc(&d16, 1024 * 4);
    srand(7);
    size_t n_bad_layout = 0, n_diff16 = 0, n_total = 0;
    double worst = 0;
    for (int rep = 0; rep < 2000; rep++) {
        const int kind = rep % 4;
        for (int l = 0; l < 64; l++)
            for (int k2 = 0; k2 < 4; k2++) {
                float va = (float)rand() / RAND_MAX * 2 - 1, vb = (float)rand() / RAND_MAX * 2 - 1;
                if (kind == 1) { va *= 64.f; }                      // candidate vectors up to 64
                if (kind == 2) { va *= (k2 == 3 ? 8000.f : 64.f); } // limits up to 2^13
                if (kind == 3) { va *= 1e-4f; vb *= 1e-3f; }         // tiny: products below the f16 normal range
                if (k2 == 3 && kind != 3) vb = -1.0f;
                a[l][k2] = (_Float16)va; b[l][k2] = (_Float16)vb;
            }
        (void)hipMemcpy(da, a.data(), 64 * sizeof(h4), hipMemcpyHostToDevice);
        (void)hipMemcpy(db, b.data(), 64 * sizeof(h4), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, d4, d16);
        std::vector<float> h4b(1024), h16(1024);
        (void)hipMemcpy(h4b.data(), d4, 4096, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h16.data(), d16, 4096, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 16; r++) {
                const int blk = r >> 2, row = 4 * (l >> 4) + (r & 3), col = l & 15;
                double ex = 0;
                float ch = 0.f;
                for (int k2 = 0; k2 < 4; k2++) {
                    ex += (double)(float)a[row][k2] * (double)(float)b[16 * blk + col][k2];
                    ch = fmaf((float)a[row][k2], (float)b[16 * blk + col][k2], ch);
                }
                const float got = h4b[l * 16 + r];
                const double scale = fabs(ex) + 1e-30;
                double mag = 0;
                for (int k2 = 0; k2 < 4; k2++) mag += fabs((double)(float)a[row][k2] * (double)(float)b[16 * blk + col][k2]);
                if (fabs(got - ex) > 1e-6 * mag + 1e-30) n_bad_layout++;
                worst = fmax(worst, fabs(got - ex) / (mag + 1e-30));
                if (got != h16[l * 16 + r]) n_diff16++;
                (void)ch; (void)scale;
                n_total++;
            }
    }
    printf("results %zu: outside 1e-6 of the exact sum (layout or precision) %zu, worst relative-to-magnitude error %.3g, "
           "different from v_mfma_f32_16x16x16_f16 on the same operands %zu\n", n_total, n_bad_layout, worst, n_diff16);
    return n_bad_layout != 0;
}
