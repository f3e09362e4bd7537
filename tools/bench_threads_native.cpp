// Per-structure calls from several host threads, every thread with its own context - the C ABI without Python.
// build: g++ -O2 -std=c++17 -Iinclude tools/bench_threads_native.cpp -Lrustsasa_amd/lib -lrustsasa_amd -Wl,-rpath,$PWD/rustsasa_amd/lib -lpthread -o /tmp/bench_threads_native
#include "rustsasa_amd.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

int main(int argc, char **argv)
{
    const int n_atoms = argc > 1 ? atoi(argv[1]) : 2622, calls = 2000;
    std::mt19937 rng(1);
    const float side = std::cbrt(n_atoms / 0.05f);
    std::uniform_real_distribution<float> u(0.f, side), ur(1.2f, 2.0f);
    std::vector<float> x(n_atoms), y(n_atoms), z(n_atoms), r(n_atoms);
    for (int i = 0; i < n_atoms; i++) { x[i] = u(rng); y[i] = u(rng); z[i] = u(rng); r[i] = ur(rng); }
    for (int nt : {1, 2, 4, 8, 16}) {
        std::vector<rsasa_context_t *> ctxs(nt);
        for (auto &c : ctxs) if (rsasa_context_create(0, &c) != RSASA_OK) { fprintf(stderr, "no context\n"); return 1; }
        std::vector<std::vector<float>> outs(nt, std::vector<float>(n_atoms));
        for (int t = 0; t < nt; t++) rsasa_calculate_sasa_soa(ctxs[t], x.data(), y.data(), z.data(), r.data(), nullptr, n_atoms, 1.4f, 100, outs[t].data());
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> ths;
        for (int t = 0; t < nt; t++)
            ths.emplace_back([&, t] { for (int k = 0; k < calls; k++) rsasa_calculate_sasa_soa(ctxs[t], x.data(), y.data(), z.data(), r.data(), nullptr, n_atoms, 1.4f, 100, outs[t].data()); });
        for (auto &th : ths) th.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d threads: %.0f structures/s (%.1f us per call and thread)\n", nt, nt * calls / dt, dt / calls * 1e6);
        for (auto c : ctxs) rsasa_context_destroy(c);
    }
}
