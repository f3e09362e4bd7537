// The literal drop-in path, timed: rsasa_calculate_sasa_internal (AoS atoms in, per-atom values out - the call
// INTEGRATION.md's Rust shim makes in place of src/lib.rs:249-254) once per structure from T host threads, every thread
// with a context of its own - what the reference's directory mode does from every rayon worker (src/main.rs:375,439).
// bench.py's `per_call` leg runs this as a child process (no interpreter lock between the threads).
//
//   bench_per_call <structures.bin> <n_points> <seconds per leg> <leg> [<leg> ...]
//
// A leg is a thread count T (a context per thread, every call by itself - the figures of round 5), cT (call combining on,
// rsasa_context_set_call_combining: a context per thread) or sT (call combining on, ONE context shared by all threads);
// PER_CALL_COMBINE_WAIT_US (default 0) is the max_wait_us of the combining legs.
//
// structures.bin: u32 n_structures, u32 offsets[n + 1], then rsasa_atom_t records (24 bytes each).  The threads take
// structures from one shared counter (cycling through the list) until the leg's time is up; every call is timed.
// Output: one JSON object per line and leg.
#include "../../../include/rustsasa_amd.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s structures.bin n_points seconds threads...\n", argv[0]); return 64; }
    std::FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 66; }
    uint32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1 || n == 0) return 65;
    std::vector<uint32_t> off(n + 1);
    if (std::fread(off.data(), 4, n + 1, f) != n + 1) return 65;
    std::vector<rsasa_atom_t> atoms(off[n]);
    if (std::fread(atoms.data(), sizeof(rsasa_atom_t), atoms.size(), f) != atoms.size()) return 65;
    std::fclose(f);
    const size_t n_points = (size_t)std::atol(argv[2]);
    const double seconds = std::atof(argv[3]);
    uint32_t longest = 0;
    for (uint32_t s = 0; s < n; s++) longest = std::max(longest, off[s + 1] - off[s]);

    for (int a = 4; a < argc; a++) {
        const char mode = argv[a][0] == 'c' || argv[a][0] == 's' ? argv[a][0] : 'n';
        const int nt = std::max(1, std::atoi(argv[a] + (mode == 'n' ? 0 : 1)));
        const char *wv = std::getenv("PER_CALL_COMBINE_WAIT_US");
        const int wait_us = wv ? std::atoi(wv) : 0;
        std::vector<rsasa_context_t *> ctxs(nt, nullptr);
        for (int t = 0; t < nt; t++) {
            if (mode == 's' && t > 0) { ctxs[t] = ctxs[0]; continue; }
            if (rsasa_context_create(0, &ctxs[t]) != RSASA_OK) { std::fprintf(stderr, "rsasa_context_create failed\n"); return 70; }
            if (mode != 'n' && rsasa_context_set_call_combining(ctxs[t], wait_us) != RSASA_OK) return 70;
        }
        uint64_t cb0 = 0, cc0 = 0, cb1 = 0, cc1 = 0;
        (void)rsasa_call_combining_stats(0, &cb0, &cc0);
        std::vector<std::vector<float>> outs(nt, std::vector<float>(longest));
        std::vector<std::vector<float>> lat(nt);
        std::vector<double> total(nt, 0.0);
        std::atomic<uint64_t> next{0}, atoms_done{0};
        std::atomic<int> failed{0};
        // one warm-up call per context (workspace growth, lattice upload)
        for (int t = 0; t < nt; t++)
            if (rsasa_calculate_sasa_internal(ctxs[t], atoms.data() + off[0], off[1] - off[0], 1.4f, n_points, -1, outs[t].data()) != RSASA_OK) failed++;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> ths;
        for (int t = 0; t < nt; t++)
            ths.emplace_back([&, t] {
                (void)rsasa_context_bind_thread(ctxs[t], nullptr);
                lat[t].reserve(1 << 16);
                for (;;) {
                    const auto c0 = std::chrono::steady_clock::now();
                    if (std::chrono::duration<double>(c0 - t0).count() >= seconds) break;
                    const uint32_t s = (uint32_t)(next.fetch_add(1) % n);
                    const uint32_t na = off[s + 1] - off[s];
                    if (rsasa_calculate_sasa_internal(ctxs[t], atoms.data() + off[s], na, 1.4f, n_points, -1, outs[t].data()) != RSASA_OK) { failed++; break; }
                    lat[t].push_back(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c0).count());
                    total[t] += outs[t][0];
                    atoms_done += na;
                }
            });
        for (auto &th : ths) th.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<float> all;
        for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end());
        (void)rsasa_call_combining_stats(0, &cb1, &cc1);
        for (int t = 0; t < nt; t++)
            if (mode != 's' || t == 0) rsasa_context_destroy(ctxs[t]);
        if (failed || all.empty()) { std::printf("{\"threads\":%d,\"error\":\"%d calls failed\"}\n", nt, failed.load()); continue; }
        auto q = [&](double p) { return all[std::min(all.size() - 1, (size_t)(p * (double)all.size()))]; };
        std::printf("{\"threads\":%d,\"mode\":\"%s\",\"calls\":%zu,\"seconds\":%.3f,\"structures_per_s\":%.1f,\"atoms_per_s\":%.1f,"
                    "\"ms_per_call_p50\":%.4f,\"ms_per_call_p99\":%.4f,\"ms_per_call_max\":%.4f,\"combined_batches\":%llu,\"calls_per_batch\":%.2f}\n",
                    nt, mode == 'n' ? "alone" : mode == 'c' ? "combined" : "combined_shared_context", all.size(), dt, (double)all.size() / dt,
                    (double)atoms_done.load() / dt, q(0.50), q(0.99), all.back(), (unsigned long long)(cb1 - cb0),
                    cb1 > cb0 ? (double)(cc1 - cc0) / (double)(cb1 - cb0) : 0.0);
    }
    return 0;
}
