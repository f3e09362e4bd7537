// Microbenchmark for the f32 matrix instruction on gfx950 (v_mfma_f32_16x16x4_f32):
//   (1) is D = A.B + C bit for bit the k-ordered fmaf chain the occlusion test needs
//       (reference src/lib.rs:143-146: mul_add(sx, vx, mul_add(sy, vy, sz * vz)) < limit), including
//       zeros, subnormal results and infinities, when C = -0.0 and the k slots hold z, y, x, (-1, limit)?
//   (2) what does it cost next to vector instructions at 8 waves per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o microbench_mfma microbench_mfma.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

// one wave per tile: A[16][4] (row-major), B[4][16] (row-major) -> D[16][16]
__global__ __launch_bounds__(64) void k_tile(const float *A, const float *B, float *D, int n_tiles)
{
    const int t = blockIdx.x, l = threadIdx.x;
    if (t >= n_tiles) return;
    const float a = A[(size_t)t * 64 + (l & 15) * 4 + (l >> 4)];   // A[i = l & 15][k = l >> 4]
    const float b = B[(size_t)t * 64 + (l >> 4) * 16 + (l & 15)];  // B[k = l >> 4][j = l & 15]
    const float nz = __int_as_float(0x80000000);
    f4 c = {nz, nz, nz, nz};
    f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++)  // D[i = 4 * (l >> 4) + r][j = l & 15]
        D[(size_t)t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static int check(const char *name, std::vector<float> &A, std::vector<float> &B, int n_tiles)
{
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, (size_t)n_tiles * 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tile, dim3(n_tiles), dim3(64), 0, 0, dA, dB, dD, n_tiles);
    std::vector<float> D((size_t)n_tiles * 256);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dD);
    long bad_chain = 0, bad_sign = 0, bad_rev = 0, subnormal_out = 0, n = 0;
    for (int t = 0; t < n_tiles; t++)
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                const float *a = &A[(size_t)t * 64 + i * 4];
                const float *b = &B[(size_t)t * 64 + j];
                // k-ordered chain from C = -0.0: fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,-0))))
                float acc = -0.0f;
                for (int k = 0; k < 4; k++) acc = fmaf(a[k], b[k * 16], acc);
                float rev = -0.0f;
                for (int k = 3; k >= 0; k--) rev = fmaf(a[k], b[k * 16], rev);
                // the expression of the reference with the first product as a plain multiply
                const float dot = fmaf(a[2], b[32], fmaf(a[1], b[16], a[0] * b[0]));
                const float ref = fmaf(a[3], b[48], dot);
                const float got = D[(size_t)t * 256 + i * 16 + j];
                const bool nan_ok = std::isnan(got) && std::isnan(acc);
                if (!nan_ok && bits(got) != bits(acc)) bad_chain++;
                if (!(std::isnan(got) && std::isnan(rev)) && bits(got) != bits(rev)) bad_rev++;
                if (!(std::isnan(got) && std::isnan(ref)) && bits(got) != bits(ref)) bad_sign++;
                if (got != 0.0f && std::fabs(got) < 1.17549435e-38f) subnormal_out++;
                n++;
            }
    printf("%-34s %9ld results: != k-ordered chain %ld, != reference expression %ld, (!= reversed chain %ld), subnormal outputs %ld\n",
           name, n, bad_chain, bad_sign, bad_rev, subnormal_out);
    return bad_chain != 0 || bad_sign != 0;
}

// ---- timing ----
#define REP4(x) x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k_time(int iters, float *out)
{
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x * 0.001f;
    lds[threadIdx.x + 256] = threadIdx.x * 0.002f;
    __syncthreads();
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, e = 0.25f;
    float pt[6];
    for (int i = 0; i < 6; i++) pt[i] = a + i;
    f4 acc[6];
    float m[6];
    for (int i = 0; i < 6; i++) { acc[i] = (f4){0.f, 0.f, 0.f, 0.f}; m[i] = 0.f; }
    const f4 z = {-0.0f, -0.0f, -0.0f, -0.0f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
            float cand = a;
            if (MODE == 3) cand = lds[(threadIdx.x + it) & 255];
#pragma unroll
            for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(cand, pt[i], MODE == 3 ? z : acc[i], 0, 0, 0);
            if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 6; i++) m[i] = fminf(fminf(fminf(m[i], acc[i][0]), acc[i][1]), fminf(acc[i][2], acc[i][3]));
            }
        }
        if (MODE == 1 || MODE == 2) {
            REP4(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %3, %3, %2, %1\n v_fma_f32 %0, %0, %3, %2\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));)
        }
        if (MODE == 5 || MODE == 6 || MODE == 7) {
            // hand-ordered: six independent MFMAs into VGPRs (C = literal 0), optional independent filler,
            // then the per-lane minimum over the four candidate rows of every tile (no canonicalisation)
            f4 d0, d1, d2, d3, d4, d5;
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %6, %7, 0\n v_mfma_f32_16x16x4_f32 %1, %6, %8, 0\n"
                         "v_mfma_f32_16x16x4_f32 %2, %6, %9, 0\n v_mfma_f32_16x16x4_f32 %3, %6, %10, 0\n"
                         "v_mfma_f32_16x16x4_f32 %4, %6, %11, 0\n v_mfma_f32_16x16x4_f32 %5, %6, %12, 0\n"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5)
                         : "v"(a), "v"(pt[0]), "v"(pt[1]), "v"(pt[2]), "v"(pt[3]), "v"(pt[4]), "v"(pt[5]));
            if (MODE == 5) {
                REP4(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %3, %3, %2, %1\n v_fma_f32 %0, %0, %3, %2\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));)
            }
            if (MODE == 7) {
                REP4(REP4(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %3, %3, %2, %1\n v_fma_f32 %0, %0, %3, %2\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));))
            }
#define MIN4(mm, dd) asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %0, %0, %3, %4" : "+v"(mm) : "v"(dd[0]), "v"(dd[1]), "v"(dd[2]), "v"(dd[3]));
            MIN4(m[0], d0) MIN4(m[1], d1) MIN4(m[2], d2) MIN4(m[3], d3) MIN4(m[4], d4) MIN4(m[5], d5)
        }
        if (MODE == 8 || MODE == 9 || MODE == 10) {
            // f16 matrix instructions (16x16x16, f32 accumulate) next to vector work: do the pipes overlap?
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 ha = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)e};
            h4 hb = {(_Float16)pt[0], (_Float16)pt[1], (_Float16)pt[2], (_Float16)pt[3]};
            if (MODE != 10) {
#pragma unroll
                for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, acc[i], 0, 0, 0);
            }
            if (MODE != 8) {
                REP4(REP4(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %3, %3, %2, %1\n v_fma_f32 %0, %0, %3, %2\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));))
            }
        }
        if (MODE == 11 || MODE == 12) {
            // the same with independent (non-accumulating) f16 matrix instructions
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 ha = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)e};
            h4 hb = {(_Float16)pt[0], (_Float16)pt[1], (_Float16)pt[2], (_Float16)pt[3]};
            f4 zz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 6; i++) { f4 d = __builtin_amdgcn_mfma_f32_16x16x16f16(ha, hb, zz, 0, 0, 0); m[i] = fmaxf(m[i], d[0] + d[3]); }
            if (MODE == 12) {
                REP4(REP4(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %0\n v_fma_f32 %2, %2, %0, %1\n v_fma_f32 %3, %3, %2, %1\n v_fma_f32 %0, %0, %3, %2\n v_fma_f32 %3, %3, %0, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));))
            }
        }
        if (MODE == 4) {  // the vector version of one phase-A trip pair: 8 candidates x 128 points = 54 VALU
            for (int q = 0; q < 2; q++) {
                REP4(asm volatile("v_mul_f32 %3, %0, %1\n v_fma_f32 %3, %1, %2, %3\n v_fma_f32 %3, %2, %0, %3\n v_sub_f32 %3, %1, %3\n v_mul_f32 %0, %3, %1\n v_fma_f32 %0, %1, %2, %0\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));)
            }
        }
    }
    float s = a + b + c + e;
    for (int i = 0; i < 6; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + m[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char *name, float *d, int blocks_per_cu)
{
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time<MODE>, dim3(blocks), dim3(256), 0, 0, 10, d);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_time<MODE>, dim3(blocks), dim3(256), 0, 0, iters, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_iters_per_simd = (double)blocks * 4 / 1024 * iters;
    printf("%-58s %d waves/SIMD %8.3f ms -> %7.1f ns per wave-iteration per SIMD\n", name, blocks_per_cu, ms,
           ms * 1e6 / wave_iters_per_simd);
}

int main()
{
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    int rc = 0;
    const int T = 8192;
    {   // realistic: B = (sz, sy, sx, -1) unit vectors, A = (vz, vy, vx, limit)
        std::vector<float> A((size_t)T * 64), B((size_t)T * 64);
        for (int t = 0; t < T; t++) {
            for (int i = 0; i < 16; i++) {
                for (int k = 0; k < 3; k++) A[(size_t)t * 64 + i * 4 + k] = 7.f * U(rng);
                A[(size_t)t * 64 + i * 4 + 3] = 8.f * U(rng);
            }
            for (int j = 0; j < 16; j++) {
                float x = U(rng), y = U(rng), z = U(rng), n = sqrtf(x * x + y * y + z * z) + 1e-9f;
                B[(size_t)t * 64 + 0 * 16 + j] = z / n;
                B[(size_t)t * 64 + 1 * 16 + j] = y / n;
                B[(size_t)t * 64 + 2 * 16 + j] = x / n;
                B[(size_t)t * 64 + 3 * 16 + j] = -1.0f;
            }
        }
        rc |= check("realistic (unit points, v, limit)", A, B, T);
        // limit == dot exactly for the diagonal: results are zeros / tiny differences
        for (int t = 0; t < T; t++)
            for (int i = 0; i < 16; i++) {
                const float *a = &A[(size_t)t * 64 + i * 4];
                const float *b = &B[(size_t)t * 64 + i];
                A[(size_t)t * 64 + i * 4 + 3] = fmaf(a[2], b[32], fmaf(a[1], b[16], a[0] * b[0]));
            }
        rc |= check("limit == dot on the diagonal", A, B, T);
    }
    {   // random bit patterns (no NaN inputs), zeros and infinities sprinkled in
        std::vector<float> A((size_t)T * 64), B((size_t)T * 64);
        auto rnd = [&]() {
            for (;;) {
                uint32_t u = rng();
                const uint32_t sel = rng() % 16;
                if (sel == 0) u &= 0x80000000u;               // +-0
                else if (sel == 1) u = (u & 0x807FFFFFu);     // subnormal
                else if (sel == 2) u = (u & 0x80000000u) | 0x7F800000u;  // +-inf
                float f = from_bits(u);
                if (!std::isnan(f)) return f;
            }
        };
        for (auto &v : A) v = rnd();
        for (auto &v : B) v = rnd();
        rc |= check("random bit patterns", A, B, T);
    }
    {   // tiny magnitudes: products and sums in the subnormal range
        std::vector<float> A((size_t)T * 64), B((size_t)T * 64);
        for (auto &v : A) v = U(rng) * 1e-19f;
        for (auto &v : B) v = U(rng) * 1e-20f;
        for (int t = 0; t < T; t++)
            for (int j = 0; j < 16; j++) B[(size_t)t * 64 + 48 + j] = -1.0f;
        for (int t = 0; t < T; t++)
            for (int i = 0; i < 16; i++) A[(size_t)t * 64 + i * 4 + 3] = U(rng) * 1e-39f;
        rc |= check("subnormal products / differences", A, B, T);
    }
    float *d; hipMalloc(&d, 64);
    for (int w : {8, 4, 2, 1}) {
        run<0>("6 MFMA 16x16x4 f32 (chained accumulators)", d, w);
        run<3>("LDS read + 6 MFMA (C = -0) + 6 x 4 v_min", d, w);
        run<1>("24 v_fma_f32", d, w);
        run<2>("6 MFMA + 24 v_fma_f32", d, w);
        run<4>("48 VALU (vector phase A: 8 candidates x 128 points)", d, w);
        run<6>("asm: 6 MFMA -> 12 v_min3 at once", d, w);
        run<5>("asm: 6 MFMA, 24 v_fma filler, 12 v_min3", d, w);
        run<7>("asm: 6 MFMA, 96 v_fma filler, 12 v_min3", d, w);
        run<8>("6 f16 MFMA 16x16x16 (chained accumulators)", d, w);
        run<10>("96 v_fma_f32", d, w);
        run<9>("6 f16 MFMA 16x16x16 + 96 v_fma_f32", d, w);
        run<11>("6 independent f16 MFMA + 6 x 2 VALU", d, w);
        run<12>("6 independent f16 MFMA + 6 x 2 VALU + 96 v_fma_f32", d, w);
    }
    printf(rc ? "MFMA_CHECK_FAILED\n" : "MFMA_CHECK_OK\n");
    return rc;
}
