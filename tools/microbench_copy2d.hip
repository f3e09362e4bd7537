// Host -> device copies of a sub-batch's x, y, z (three arrays a fixed distance apart in one pinned block): three
// hipMemcpyAsync against one hipMemcpy2DAsync of three rows, and one contiguous copy of the same bytes.
// Build: hipcc --offload-arch=gfx950 -O2 -o microbench_copy2d microbench_copy2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

int main()
{
    const size_t N = 11720983, sub = N / 8;  // atoms of the proteome batch, of one of eight sub-batches
    float *h = nullptr, *d = nullptr;
    if (hipHostMalloc((void **)&h, 3 * N * 4, hipHostMallocDefault) != hipSuccess) return 1;
    if (hipMalloc((void **)&d, 3 * sub * 4) != hipSuccess) return 1;
    std::memset(h, 1, 3 * N * 4);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto time = [&](const char *name, auto fn) {
        for (int i = 0; i < 3; i++) fn(0);
        (void)hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        const int reps = 8;
        for (int i = 0; i < reps; i++) fn((size_t)i);
        (void)hipStreamSynchronize(s);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("%-44s %8.1f us  %6.1f GB/s\n", name, sec * 1e6, 3.0 * sub * 4 / sec / 1e9);
    };
    time("three copies (x, y, z of a sub-batch)", [&](size_t k) {
        for (int c = 0; c < 3; c++) (void)hipMemcpyAsync(d + c * sub, h + c * N + k * sub, sub * 4, hipMemcpyHostToDevice, s);
    });
    time("one 2D copy, three rows", [&](size_t k) {
        (void)hipMemcpy2DAsync(d, sub * 4, h + k * sub, N * 4, sub * 4, 3, hipMemcpyHostToDevice, s);
    });
    time("one contiguous copy of the same bytes", [&](size_t k) {
        (void)hipMemcpyAsync(d, h + (k % 2) * 3 * sub, 3 * sub * 4, hipMemcpyHostToDevice, s);
    });
    // the pipelined host path's upload sequence of one proteome batch: per sub-batch x, y, z and a block of 5 bytes per atom
    {
        char *hb = nullptr, *db = nullptr;
        if (hipHostMalloc((void **)&hb, N * 5, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&db, sub * 5) != hipSuccess) return 1;
        std::memset(hb, 2, N * 5);
        for (int rep = 0; rep < 3; rep++) {
            (void)hipStreamSynchronize(s);
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t k = 0; k < 8; k++) {
                for (int c = 0; c < 3; c++) (void)hipMemcpyAsync(d + c * sub, h + c * N + k * sub, sub * 4, hipMemcpyHostToDevice, s);
                (void)hipMemcpyAsync(db, hb + k * sub * 5, sub * 5, hipMemcpyHostToDevice, s);
            }
            (void)hipStreamSynchronize(s);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("32 copies of one batch (17 bytes per atom): %8.1f us  %6.1f GB/s\n", sec * 1e6, 8.0 * sub * 17 / sec / 1e9);
        }
    }
    return 0;
}
