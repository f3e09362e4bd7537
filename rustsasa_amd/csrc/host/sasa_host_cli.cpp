// Small driver of the C++ host API used by the tests: reads one structure file and
// prints the level result as JSON (shape of the reference's SASAResult serialisation,
// src/structures/atomic.rs:62-70 + src/utils/io.rs:11-13).  Not a port of the
// reference CLI (src/main.rs), which is out of scope.
#include <cstdio>
#include <cstring>
#include <string>

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

int main(int argc, char **argv)
{
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
        if (level == "atom") {
            auto r = make<AtomLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            std::printf("{\"Atom\":[");
            for (size_t i = 0; i < r.value.size(); i++) std::printf("%s%.9g", i ? "," : "", r.value[i]);
            std::printf("]}\n");
        } else if (level == "residue") {
            auto r = make<ResidueLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            std::printf("{\"Residue\":[");
            for (size_t i = 0; i < r.value.size(); i++) {
                const auto &v = r.value[i];
                std::printf("%s{\"serial_number\":%lld,\"insertion_code\":", i ? "," : "", (long long)v.serial_number);
                print_str(v.insertion_code);
                std::printf(",\"value\":%.9g,\"name\":", v.value);
                print_str(v.name);
                std::printf(",\"is_polar\":%s,\"chain_id\":", v.is_polar ? "true" : "false");
                print_str(v.chain_id);
                std::printf("}");
            }
            std::printf("]}\n");
        } else if (level == "chain") {
            auto r = make<ChainLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            std::printf("{\"Chain\":[");
            for (size_t i = 0; i < r.value.size(); i++) {
                std::printf("%s{\"name\":", i ? "," : "");
                print_str(r.value[i].name);
                std::printf(",\"value\":%.9g}", r.value[i].value);
            }
            std::printf("]}\n");
        } else if (level == "protein") {
            auto r = make<ProteinLevel>(argc, argv).process(pdb);
            if (!r.ok()) return fail(r);
            std::printf("{\"Protein\":{\"global_total\":%.9g,\"polar_total\":%.9g,\"non_polar_total\":%.9g}}\n",
                        r.value.global_total, r.value.polar_total, r.value.non_polar_total);
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
