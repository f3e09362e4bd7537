// Small driver of the C++ host API used by the tests: reads one structure file and
// prints the level result as JSON (shape of the reference's SASAResult serialisation,
// src/structures/atomic.rs:62-70 + src/utils/io.rs:11-13).  Not a port of the
// reference CLI (src/main.rs), which is out of scope.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/rustsasa_amd.hpp"

using namespace rustsasa;

static void print_str(const std::string &s)
{
    std::putchar('"');
    for (char c : s) {
        if (c == '"' || c == '\\') std::putchar('\\');
        std::putchar(c);
    }
    std::putchar('"');
}

template <typename L>
static SASAOptions<L> make(int argc, char **argv)
{
    SASAOptions<L> o;
    for (int i = 3; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--n-points" && i + 1 < argc) o.with_n_points((size_t)std::atol(argv[++i]));
        else if (a == "--probe-radius" && i + 1 < argc) o.with_probe_radius((float)std::atof(argv[++i]));
        else if (a == "--include-hydrogens") o.with_include_hydrogens(true);
        else if (a == "--include-hetatms") o.with_include_hetatms(true);
        else if (a == "--allow-vdw-fallback") o.with_allow_vdw_fallback(true);
        else if (a == "--read-radii-from-occupancy") o.with_read_radii_from_occupancy(true);
        else if (a == "--radii-file" && i + 1 < argc) o.with_radii_file(argv[++i]);
    }
    return o;
}

template <typename T>
static int fail(const Result<T> &r)
{
    std::printf("{\"error\":%d,\"message\":", (int)r.error);
    print_str(r.message);
    std::printf("}\n");
    return 2;
}

// `files <level> <list file>`: directory mode at library level (SASAOptions::process_files).
template <typename L, typename PrintOne>
static int run_files(int argc, char **argv, PrintOne print_one)
{
    std::vector<std::string> paths;
    {
        std::ifstream f(argv[3]);
        for (std::string line; std::getline(f, line);)
            if (!line.empty()) paths.push_back(line);
    }
    unsigned threads = 0;
    size_t batch = 0;
    bool full = false;
    std::string out_dir;
    int workers = 0, devices = 1;  // --workers K: K GPU worker contexts, on devices k % --devices
    int calls = 1;                 // --calls N: process_files N times in this process (the last call's results are printed)
    int simd_width = 0;            // --simd-width W: the pulp lane count of the context(s) the call is given (0: default)
    for (int i = 4; i < argc; i++) {
        if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = (unsigned)std::atoi(argv[i + 1]);
        if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) batch = (size_t)std::atol(argv[i + 1]);
        if (!std::strcmp(argv[i], "--workers") && i + 1 < argc) workers = std::atoi(argv[i + 1]);
        if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) devices = std::max(1, std::atoi(argv[i + 1]));
        if (!std::strcmp(argv[i], "--calls") && i + 1 < argc) calls = std::max(1, std::atoi(argv[i + 1]));
        if (!std::strcmp(argv[i], "--simd-width") && i + 1 < argc) simd_width = std::atoi(argv[i + 1]);
        if (!std::strcmp(argv[i], "--full")) full = true;
        if (!std::strcmp(argv[i], "--out-dir") && i + 1 < argc) out_dir = argv[i + 1];  // per-file JSON (OptionValues::output_dir)
    }
    std::vector<rsasa_context_t *> ctxs;
    for (int k = 0; k < workers; k++) {
        rsasa_context_t *c = nullptr;
        const int rc = rsasa_context_create(k % devices, &c);
        if (rc != RSASA_OK) {
            std::fprintf(stderr, "rsasa_context_create(%d): %s\n", k % devices, rsasa_status_string(rc));
            return 70;
        }
        ctxs.push_back(c);
        if (simd_width && rsasa_context_set_simd_width(c, simd_width) != RSASA_OK) return 71;
    }
    if (simd_width && ctxs.empty() && rsasa_context_set_simd_width(nullptr, simd_width) != RSASA_OK) return 71;  // (the default context)
    FilesTimings t;
    auto opts = make<L>(argc - 1, argv + 1);
    if (!ctxs.empty()) opts.with_contexts(ctxs);
    if (!out_dir.empty()) opts.with_output_dir(out_dir);
    auto res = opts.process_files(paths, threads, batch, &t);
    std::string call_s = std::to_string(t.total_seconds);  // every call's wall time: the first one includes the HIP runtime's start-up
    for (int k = 1; k < calls; k++) {
        res = opts.process_files(paths, threads, batch, &t);
        call_s += "," + std::to_string(t.total_seconds);
    }
    for (rsasa_context_t *c : ctxs) rsasa_context_destroy(c);
    size_t n_ok = 0;
    for (const auto &r : res) n_ok += r.ok();
    std::string widths;
    for (int w : t.worker_simd_widths) widths += (widths.empty() ? "" : ",") + std::to_string(w);
    std::printf("{\"n_files\":%zu,\"n_ok\":%zu,\"n_atoms\":%zu,\"parse_s\":%.6f,\"compute_s\":%.6f,\"total_s\":%.6f,\"calls_s\":[%s],\"worker_simd_widths\":[%s],\"bytes_written\":%llu,\"results\":[",
                t.n_files, n_ok, t.n_atoms, t.parse_seconds, t.compute_seconds, t.total_seconds, call_s.c_str(), widths.c_str(),
                (unsigned long long)t.bytes_written);
    for (size_t i = 0; i < res.size(); i++) {
        std::printf("%s", i ? "," : "");
        if (!res[i].ok()) {
            std::printf("{\"error\":%d}", (int)res[i].error);
        } else if (full) {
            print_one(res[i].value);
        } else {
            std::printf("{}");
        }
    }
    std::printf("]}\n");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && std::string(argv[1]) == "decimal") {
        // reader self-check: for every stdin token print the fast parser's and strtod's bits
        for (std::string tok; std::cin >> tok;) {
            const double a = parse_decimal_text(tok), b = std::strtod(tok.c_str(), nullptr);
            std::printf("%a %a\n", a, b);
        }
        return 0;
    }
    if (argc >= 3 && std::string(argv[1]) == "model-selftest") {
        // the structure model's copy / move rules (its nodes live in a pool the Structure owns): every copy and every
        // moved-to object must stay whole after its source is gone
        const std::string want = Structure::open(argv[2]).to_pdb_text();
        const size_t n = Structure::open(argv[2]).atom_count();
        int bad = 0;
        auto check = [&](const Structure &s, const char *what) {
            if (s.atom_count() != n || s.to_pdb_text() != want) { std::fprintf(stderr, "model-selftest: %s differs\n", what); bad++; }
        };
        {
            auto src = std::make_unique<Structure>(Structure::open(argv[2]));
            Structure copy(*src);
            Structure copy_assigned;
            copy_assigned = *src;
            Structure moved(std::move(*src));
            if (!src->chains.empty()) { std::fprintf(stderr, "model-selftest: moved-from structure is not empty\n"); bad++; }
            src->chains.push_back(Chain{"Z", {}});  // a moved-from structure is usable (and uses its own memory)
            src.reset();
            check(copy, "copy");
            check(copy_assigned, "copy-assigned");
            check(moved, "moved");
            Structure move_assigned = Structure::open(argv[2]);  // (a pooled target)
            move_assigned = std::move(moved);
            check(move_assigned, "move-assigned");
            move_assigned = *&move_assigned;  // self-assignment
            check(move_assigned, "self-assigned");
            std::vector<Structure> many;
            for (int i = 0; i < 9; i++) many.push_back(i % 2 ? Structure(copy) : Structure::open(argv[2]));  // (reallocations move)
            many.erase(many.begin() + 1, many.begin() + 4);
            for (const Structure &s : many) check(s, "vector element");
            Chain one = many.front().chains.front();  // a copied chain is independent of the structure
            many.clear();
            size_t atoms = 0;
            for (const Residue &r : one.residues)
                for (const Conformer &c : r.conformers) atoms += c.atoms.size();
            if (atoms == 0) { std::fprintf(stderr, "model-selftest: copied chain is empty\n"); bad++; }
        }
        std::printf("{\"ok\":%s,\"atoms\":%zu}\n", bad ? "false" : "true", n);
        return bad ? 1 : 0;
    }
    if (argc >= 4 && std::string(argv[1]) == "parse-bench") {
        // reader micro-benchmark: open + parse the file `reps` times on one thread
        const int reps = std::atoi(argv[3]);
        size_t atoms = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) atoms += Structure::open(argv[2]).atom_count();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("{\"reps\":%d,\"atoms\":%zu,\"ms_per_file\":%.4f,\"ns_per_atom\":%.1f}\n", reps,
                    atoms / (size_t)reps, s / reps * 1e3, s / (double)atoms * 1e9);
        return 0;
    }
    if (argc >= 2 && std::string(argv[1]) == "json-floats") {  // test hook: the JSON writer's f32 printing (no GPU)
        std::vector<float> v;
        for (int i = 2; i < argc; i++) {
            const uint32_t bits = (uint32_t)std::strtoul(argv[i], nullptr, 16);
            float f;
            std::memcpy(&f, &bits, 4);
            v.push_back(f);
        }
        std::printf("%s\n", sasa_result_to_json(v).c_str());
        return 0;
    }
    if (argc >= 4 && std::string(argv[1]) == "files") {
        const std::string level = argv[2];
        try {
            if (level == "atom")
                return run_files<AtomLevel>(argc, argv, [](const std::vector<float> &v) {
                    std::printf("[");
                    for (size_t i = 0; i < v.size(); i++) std::printf("%s%.9g", i ? "," : "", v[i]);
                    std::printf("]");
                });
            // --labels: every value with its chain id, ["A", value] (the quality gate sums per chain: tests/quality.rs:60-100)
            bool labels = false;
            for (int i = 4; i < argc; i++) labels |= !std::strcmp(argv[i], "--labels");
            if (level == "residue")
                return run_files<ResidueLevel>(argc, argv, [labels](const std::vector<ResidueResult> &v) {
                    std::printf("[");
                    for (size_t i = 0; i < v.size(); i++) {
                        if (labels) std::printf("%s[\"%s\",%.9g]", i ? "," : "", v[i].chain_id.c_str(), v[i].value);
                        else std::printf("%s%.9g", i ? "," : "", v[i].value);
                    }
                    std::printf("]");
                });
            if (level == "chain")
                return run_files<ChainLevel>(argc, argv, [labels](const std::vector<ChainResult> &v) {
                    std::printf("[");
                    for (size_t i = 0; i < v.size(); i++) {
                        if (labels) std::printf("%s[\"%s\",%.9g]", i ? "," : "", v[i].name.c_str(), v[i].value);
                        else std::printf("%s%.9g", i ? "," : "", v[i].value);
                    }
                    std::printf("]");
                });
            if (level == "protein")
                return run_files<ProteinLevel>(argc, argv, [](const ProteinResult &v) {
                    std::printf("[%.9g,%.9g,%.9g]", v.global_total, v.polar_total, v.non_polar_total);
                });
        } catch (const std::exception &e) {
            std::fprintf(stderr, "%s\n", e.what());
            return 70;
        }
        return 64;
    }
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <atom|residue|chain|protein> <file> [options]\n", argv[0]);
        return 64;
    }
    const std::string level = argv[1];
    Structure pdb;
    try {
        pdb = Structure::open(argv[2]);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 66;
    }
    try {
        const char *bf_out = nullptr;  // --bfactor-out FILE: write the values into b-factors, save as PDB
        for (int i = 3; i + 1 < argc; i++)
            if (!std::strcmp(argv[i], "--bfactor-out")) bf_out = argv[i + 1];
        auto emit = [&](const auto &value) -> int {
            std::printf("%s\n", sasa_result_to_json(value).c_str());
            if (bf_out) {
                std::string err;
                if (!sasa_result_to_protein_object(pdb, value, &err)) {
                    std::fprintf(stderr, "%s\n", err.c_str());
                    return 3;
                }
                pdb.save_pdb(bf_out);
            }
            return 0;
        };
        if (level == "atom") {
            auto r = make<AtomLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            return emit(r.value);
        } else if (level == "residue") {
            auto r = make<ResidueLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            return emit(r.value);
        } else if (level == "chain") {
            auto r = make<ChainLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            return emit(r.value);
        } else if (level == "protein") {
            auto r = make<ProteinLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            return emit(r.value);
        } else if (level == "select") {
            // reader + atom selection + radii, no GPU: what ChainLevel would hand to the hot path, one atom per
            // line (x y z radius as hex floats, id, chain index), preceded by the chains' ids
            auto r = select_atoms_by_chain(pdb, make<ChainLevel>(argc, argv).values());
            if (!r.ok()) return fail(r);
            std::printf("{\"chains\":[");
            for (size_t c = 0; c < r.value.chain_ids.size(); c++) {
                std::printf("%s", c ? "," : "");
                print_str(r.value.chain_ids[c]);
            }
            std::printf("],\"chain_end\":[");
            for (size_t c = 0; c < r.value.chain_end.size(); c++) std::printf("%s%u", c ? "," : "", r.value.chain_end[c]);
            std::printf("],\"atoms\":[");
            for (size_t i = 0; i < r.value.atoms.size(); i++) {
                const rsasa_atom_t &a = r.value.atoms[i];
                std::printf("%s[%.9g,%.9g,%.9g,%.9g,\"%llu\"]", i ? "," : "", a.position[0], a.position[1], a.position[2],
                            a.radius, (unsigned long long)a.id);
            }
            std::printf("]}\n");
        } else if (level == "prepare-fast" || level == "prepare-general") {
            // directory mode's per-file work without a GPU: `--level N` (0 atom, 1 residue, 2 chain, 3 protein)
            int lv = 1;
            for (int i = 3; i + 1 < argc; i++)
                if (!std::strcmp(argv[i], "--level")) lv = std::atoi(argv[i + 1]);
            std::printf("%s\n", detail::debug_prepare_json(argv[2], make<ChainLevel>(argc, argv).values(), lv, level == "prepare-fast").c_str());
        } else if (level == "prepare-bench-fast" || level == "prepare-bench-general") {
            // the same work timed: `--reps N` passes over the file's text (read once), no GPU
            int lv = 1, reps = 20;
            for (int i = 3; i + 1 < argc; i++) {
                if (!std::strcmp(argv[i], "--level")) lv = std::atoi(argv[i + 1]);
                if (!std::strcmp(argv[i], "--reps")) reps = std::atoi(argv[i + 1]);
            }
            size_t atoms = 0;
            bool used_fast = false;
            const double sec = detail::debug_prepare_seconds(argv[2], make<ChainLevel>(argc, argv).values(), lv, level == "prepare-bench-fast", reps, &atoms, &used_fast);
            std::printf("{\"fast\":%s,\"atoms\":%zu,\"seconds_per_pass\":%.6g,\"ns_per_atom\":%.1f}\n", used_fast ? "true" : "false", atoms, sec,
                        atoms ? sec * 1e9 / (double)atoms : 0.0);
        } else if (level == "rewrite") {  // reader -> writer round trip, no GPU
            std::printf("%s", pdb.to_pdb_text().c_str());
        } else if (level == "parse") {  // reader only (no GPU): atom / residue / chain counts
            size_t res = 0;
            for (const auto &c : pdb.chains) res += c.residues.size();
            std::printf("{\"chains\":%zu,\"residues\":%zu,\"atoms\":%zu}\n", pdb.chains.size(), res, pdb.atom_count());
        } else {
            return 64;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 70;
    }
    return 0;
}
