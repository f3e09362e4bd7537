// Host API above the C ABI (include/rustsasa_amd.hpp): minimal PDB / mmCIF
// reader, radii lookup, and the four level processors of the reference's
// src/options.rs.  No SASA arithmetic happens here: per-atom values and the
// sequential f32 segment sums come from the GPU (rsasa_calculate_sasa_batch).
#include "../../../include/rustsasa_amd.hpp"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <climits>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <malloc.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include <cmath>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <type_traits>

namespace rsasa { const char *tuning_env(const char *name); }  // context.cpp: measurement switches, read only under RSASA_TUNING=1
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>  // (the mmCIF short cut finds a row's separators sixteen bytes at a time)

#define RSASA_ROW_SSE2 1
#endif

namespace rustsasa {

// ------------------------------------------------------------------ utils --

static std::string trim(const std::string &s)
{
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) b++;
    while (e > b && std::isspace((unsigned char)s[e - 1])) e--;
    return s.substr(b, e - b);
}

static std::string upper(std::string s)
{
    for (auto &c : s) c = (char)std::toupper((unsigned char)c);
    return s;
}

// utils.rs:24-33
std::int64_t serialize_chain_id(const std::string &s)
{
    std::int64_t result = 0;
    for (unsigned char c : s)
        if (std::isalpha(c)) result = result * 10 + ((std::int64_t)std::toupper(c) - 64);
    return result;
}

// utils.rs:83-87 applied to (&str, usize): FNV-1a over the str bytes, the 0xff
// str terminator of Rust's Hash impl, then the little-endian usize.
std::uint64_t fnv_hash_altloc_serial(const std::string &alt, std::size_t serial)
{
    std::uint64_t h = 0xcbf29ce484222325ull;
    auto feed = [&h](unsigned char b) { h ^= b; h *= 0x100000001b3ull; };
    for (unsigned char c : alt) feed(c);
    feed(0xff);
    std::uint64_t v = (std::uint64_t)serial;
    for (int i = 0; i < 8; i++) feed((unsigned char)(v >> (8 * i)));
    return h;
}

// consts.rs:7-16
bool is_polar_residue(const std::string &name)
{
    static const char *polar[] = {"SER", "THR", "CYS", "ASN", "GLN", "TYR"};
    for (const char *p : polar)
        if (name == p) return true;
    return false;
}

// ------------------------------------------------------------------ radii --

struct ProtorEntry { const char *residue, *atom; float radius; };
static const ProtorEntry kProtor[] = {
#include "protor_table.inc"
};

static const RadiiConfig &protor_map()
{
    static const RadiiConfig m = [] {
        RadiiConfig t;
        for (const auto &e : kProtor) t[e.residue][e.atom] = e.radius;
        return t;
    }();
    return m;
}

// The same table as one open-addressing array keyed by the two names packed into 64 bits (residue names of the table
// have at most 3 characters, atom names at most 4): one multiply and usually one probe per atom instead of two string
// hashes.  Names that do not pack (longer) are not in the table.
struct ProtorFlat {
    static constexpr unsigned kBits = 11, kSize = 1u << kBits;  // 2048 slots for about 500 entries
    std::uint64_t key[kSize];
    float radius[kSize];
    static bool pack(const char *res, size_t n_res, const char *atom, size_t n_atom, std::uint64_t *out)
    {
        if (n_res == 0 || n_res > 4 || n_atom == 0 || n_atom > 4) return false;
        std::uint64_t k = 0;
        for (size_t i = 0; i < n_res; i++) k |= (std::uint64_t)(unsigned char)res[i] << (8 * i);
        for (size_t i = 0; i < n_atom; i++) k |= (std::uint64_t)(unsigned char)atom[i] << (32 + 8 * i);
        *out = k;
        return true;
    }
    static unsigned slot(std::uint64_t k) { return (unsigned)((k * 0x9E3779B97F4A7C15ull) >> (64 - kBits)); }
    ProtorFlat()
    {
        for (auto &k : key) k = 0;  // (no packed name pair is 0)
        for (const auto &e : kProtor) {
            std::uint64_t k;
            if (!pack(e.residue, std::strlen(e.residue), e.atom, std::strlen(e.atom), &k)) continue;
            unsigned s = slot(k);
            while (key[s] != 0 && key[s] != k) s = (s + 1) & (kSize - 1);
            key[s] = k;
            radius[s] = e.radius;
        }
    }
    bool find(const std::string &res, const std::string &atom, float *out) const
    {
        return find(res.data(), res.size(), atom.data(), atom.size(), out);
    }
    bool find(const char *res, size_t n_res, const char *atom, size_t n_atom, float *out) const
    {
        std::uint64_t k;
        if (!pack(res, n_res, atom, n_atom, &k)) return false;
        for (unsigned s = slot(k); key[s] != 0; s = (s + 1) & (kSize - 1))
            if (key[s] == k) { *out = radius[s]; return true; }
        return false;
    }
};

static const ProtorFlat &protor_flat()
{
    static const ProtorFlat flat;
    return flat;
}

bool get_protor_radius(const std::string &residue, const std::string &atom, float *out)
{
    const ProtorFlat &flat = protor_flat();
    if (flat.find(residue, atom, out)) return true;
    if (residue.size() <= 4 && atom.size() <= 4 && !residue.empty() && !atom.empty()) return false;  // packs, not there
    const auto &m = protor_map();  // (names the flat table cannot hold: the general lookup)
    auto r = m.find(residue);
    if (r == m.end()) return false;
    auto a = r->second.find(atom);
    if (a == r->second.end()) return false;
    *out = a->second;
    return true;
}

// consts.rs:31-81
RadiiConfig parse_radii_config(const std::string &content)
{
    std::unordered_map<std::string, float> types;
    RadiiConfig atoms;
    bool in_types = false, in_atoms = false;
    std::istringstream is(content);
    std::string raw;
    while (std::getline(is, raw)) {
        std::string line = trim(raw);
        if (line.empty() || line[0] == '#' || line.rfind("name:", 0) == 0) continue;
        if (line == "types:") { in_types = true; in_atoms = false; continue; }
        if (line == "atoms:") { in_types = false; in_atoms = true; continue; }
        std::istringstream ls(line);
        std::vector<std::string> p;
        for (std::string t; ls >> t;) p.push_back(t);
        if (in_types && p.size() >= 2) {
            char *end = nullptr;
            float v = std::strtof(p[1].c_str(), &end);
            if (end && *end == '\0') types[p[0]] = v;
        } else if (in_atoms && p.size() >= 3) {
            auto t = types.find(p[2]);
            if (t != types.end()) atoms[p[0]][p[1]] = t->second;
        }
    }
    return atoms;
}

RadiiConfig load_radii_from_file(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Failed to load radii file: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_radii_config(ss.str());
}

// Element van-der-Waals radii (Alvarez 2013, the table pdbtbx's
// Element::atomic_radius().van_der_waals is built from).  The reference's own
// tests pin N, C, O, S (tests/units.rs:18-43); the rest is unpinned here.
bool vdw_radius(const std::string &element, float *out)
{
    static const std::pair<const char *, float> t[] = {
        {"H", 1.20f},  {"HE", 1.43f}, {"LI", 2.12f}, {"BE", 1.98f}, {"B", 1.91f},  {"C", 1.77f},
        {"N", 1.66f},  {"O", 1.50f},  {"F", 1.46f},  {"NE", 1.58f}, {"NA", 2.50f}, {"MG", 2.51f},
        {"AL", 2.25f}, {"SI", 2.19f}, {"P", 1.90f},  {"S", 1.89f},  {"CL", 1.82f}, {"AR", 1.83f},
        {"K", 2.73f},  {"CA", 2.62f}, {"SC", 2.58f}, {"TI", 2.46f}, {"V", 2.42f},  {"CR", 2.45f},
        {"MN", 2.45f}, {"FE", 2.44f}, {"CO", 2.40f}, {"NI", 2.40f}, {"CU", 2.38f}, {"ZN", 2.39f},
        {"GA", 2.32f}, {"GE", 2.29f}, {"AS", 1.88f}, {"SE", 1.82f}, {"BR", 1.86f}, {"KR", 2.25f},
        {"RB", 3.21f}, {"SR", 2.84f}, {"MO", 2.45f}, {"CD", 2.49f}, {"I", 2.04f},  {"XE", 2.06f},
        {"CS", 3.48f}, {"BA", 3.03f}, {"W", 2.57f},  {"PT", 2.29f}, {"AU", 2.32f}, {"HG", 2.45f},
        {"PB", 2.60f}, {"U", 2.71f},  {"D", 1.20f}};
    for (const auto &e : t)
        if (element == e.first) { *out = e.second; return true; }
    return false;
}

// ---------------------------------------------------------------- Structure --

bool Residue::name(std::string *out) const
{
    if (conformers.empty()) return false;
    for (const auto &c : conformers)
        if (c.name != conformers.front().name) return false;
    *out = conformers.front().name;
    return true;
}

std::size_t Structure::atom_count() const
{
    std::size_t n = 0;
    for (const auto &c : chains)
        for (const auto &r : c.residues)
            for (const auto &f : r.conformers) n += f.atoms.size();
    return n;
}

namespace {

std::string element_from_name(const std::string &name)
{
    for (unsigned char c : name)
        if (std::isalpha(c)) return std::string(1, (char)std::toupper(c));
    return "";
}

// Decimal text -> double.  Plain decimals with at most 15 significant digits (every PDB
// %8.3f field, typical mmCIF Cartn values) take Clinger's exact fast path: an integer below
// 2^53 divided by a power of ten below 10^22 is one correctly rounded IEEE division, so the
// result equals strtod's.  Anything else (exponents, long mantissas, garbage) goes to strtod.
double parse_decimal(const char *p, size_t n)
{
    static const double pow10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *b = p, *e = p + n;
    while (b < e && (*b == ' ' || *b == '\t')) b++;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t')) e--;
    const char *q = b;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) neg = (*q++ == '-');
    unsigned long long mant = 0;
    int digits = 0, frac = 0;
    bool seen_dot = false, ok = q < e;
    for (; q < e; q++) {
        const char c = *q;
        if (c >= '0' && c <= '9') {
            if (mant != 0 || c != '0') digits++;
            mant = mant * 10 + (unsigned)(c - '0');
            if (seen_dot) frac++;
        } else if (c == '.' && !seen_dot) {
            seen_dot = true;
        } else {
            ok = false;
            break;
        }
        if (digits > 15) { ok = false; break; }
    }
    if (ok && frac <= 22) {
        const double v = (double)mant / pow10[frac];
        return neg ? -v : v;
    }
    return std::strtod(std::string(b, (size_t)(e - b)).c_str(), nullptr);
}

// A %w.df field exactly as PDB writers print it - [spaces][-]digits '.' d digits, the point at its column: the same
// integer / power-of-ten division as parse_decimal (so the same double, -0.0 included), without its scanning.  Anything
// else (blank, exponent, shifted point, a sign without digits) returns false and takes the general path.
inline bool fixed_decimal(const char *p, int w, int d, double &out)
{
    static const double pow10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6};
    const int dot = w - d - 1;
    if (p[dot] != '.') return false;
    int i = 0;
    while (i < dot && p[i] == ' ') i++;
    bool neg = false;
    if (i < dot && p[i] == '-') { neg = true; i++; }
    if (i == dot) return false;
    unsigned long long mant = 0;
    for (; i < dot; i++) {
        const unsigned c = (unsigned)(p[i] - '0');
        if (c > 9) return false;
        mant = mant * 10 + c;
    }
    for (i = dot + 1; i < w; i++) {
        const unsigned c = (unsigned)(p[i] - '0');
        if (c > 9) return false;
        mant = mant * 10 + c;
    }
    const double v = (double)mant / pow10[d];
    out = neg ? -v : v;
    return true;
}

// columns [from, to] as a decimal; `decimals` > 0: try the fixed %w.df layout first
template <typename Line>
inline double column_decimal(const Line &line, size_t from, size_t to, double missing, int decimals = 0)
{
    if (line.size() < from) return missing;
    if (decimals > 0 && line.size() >= to && to - from + 1 <= 12) {
        double v;
        if (fixed_decimal(line.data() + from - 1, (int)(to - from + 1), decimals, v)) return v;
    }
    const size_t len = std::min(to, line.size()) - from + 1;
    const char *p = line.data() + from - 1;
    size_t i = 0;
    while (i < len && p[i] == ' ') i++;
    return i == len ? missing : parse_decimal(p + i, len - i);  // (a blank field is "missing")
}

template <typename Line>
inline long column_int(const Line &line, size_t from, size_t to, bool *ok)
{
    long v = 0;
    bool neg = false, any = false, good = true;
    if (line.size() >= from) {
        const size_t end = std::min(to, line.size());
        for (size_t i = from - 1; i < end; i++) {
            const char c = line[i];
            if (c == ' ') { if (any) { for (size_t j = i; j < end; j++) good = good && line[j] == ' '; break; } continue; }
            if (c == '-' && !any && !neg) { neg = true; continue; }
            if (c < '0' || c > '9') { good = false; break; }
            v = v * 10 + (c - '0');
            any = true;
        }
    }
    if (ok) *ok = good && any;
    return neg ? -v : v;
}

}  // namespace

double parse_decimal_text(const std::string &text) { return parse_decimal(text.data(), text.size()); }

namespace {

// A line of the input text, not copied.
struct LineView {
    const char *p;
    size_t n;
    size_t size() const { return n; }
    const char *data() const { return p; }
    char operator[](size_t i) const { return p[i]; }
    bool starts_with(const char *lit) const
    {
        const size_t k = std::strlen(lit);
        return n >= k && std::memcmp(p, lit, k) == 0;
    }
};

// columns [from, to] (1-based, inclusive) without surrounding white space: begin and length
inline std::pair<const char *, size_t> field_view(const LineView &line, size_t from, size_t to)
{
    if (line.n < from) return {line.p, 0};
    const char *b = line.p + from - 1, *e = line.p + std::min(to, line.n);
    auto space = [](char c) { return c == ' ' || (c >= '\t' && c <= '\r'); };  // isspace in the C locale
    while (b < e && space(*b)) b++;
    while (e > b && space(e[-1])) e--;
    return {b, (size_t)(e - b)};
}

using TextView = std::pair<const char *, size_t>;

inline bool same(const std::string &s, const TextView &v)
{
    return s.size() == v.second && (v.second == 0 || std::memcmp(s.data(), v.first, v.second) == 0);
}

// The model add_atom builds (chains by id, residues by (number, insertion code), conformers by
// (name, alt-loc), existing entries searched from the back), from views into the text: no
// per-line strings, and consecutive atoms of one conformer - nearly all of them - skip the searches.
struct ModelBuilder {
    Structure &s;
    std::pmr::memory_resource *const mem = s.chains.get_allocator().resource();  // the structure's pool
    size_t ci = (size_t)-1, ri = (size_t)-1, fi = (size_t)-1;  // chain / residue / conformer of the previous atom

    // a new, default-constructed record at the end of the atom's conformer
    AtomRecord &add(const TextView &chain_id, std::int64_t res_seq, const TextView &icode, const TextView &res_name,
                    const TextView &alt, size_t expect = 16)
    {
        if (ci == (size_t)-1 || !same(s.chains[ci].id, chain_id)) {
            ci = (size_t)-1;
            for (size_t k = s.chains.size(); k-- > 0;)
                if (same(s.chains[k].id, chain_id)) { ci = k; break; }
            if (ci == (size_t)-1) {
                s.chains.push_back(Chain{std::string(chain_id.first, chain_id.second), std::pmr::vector<Residue>(mem)});
                ci = s.chains.size() - 1;
            }
            ri = fi = (size_t)-1;
        }
        Chain &chain = s.chains[ci];
        if (ri == (size_t)-1 || chain.residues[ri].serial_number != res_seq || !same(chain.residues[ri].insertion_code, icode)) {
            ri = (size_t)-1;
            for (size_t k = chain.residues.size(); k-- > 0;)
                if (chain.residues[k].serial_number == res_seq && same(chain.residues[k].insertion_code, icode)) { ri = k; break; }
            if (ri == (size_t)-1) {
                chain.residues.push_back(Residue{res_seq, std::string(icode.first, icode.second), std::pmr::vector<Conformer>(mem)});
                ri = chain.residues.size() - 1;
            }
            fi = (size_t)-1;
        }
        Residue &res = chain.residues[ri];
        if (fi == (size_t)-1 || !same(res.conformers[fi].name, res_name) || !same(res.conformers[fi].alt_loc, alt)) {
            fi = (size_t)-1;
            for (size_t k = 0; k < res.conformers.size(); k++)
                if (same(res.conformers[k].name, res_name) && same(res.conformers[k].alt_loc, alt)) { fi = k; break; }
            if (fi == (size_t)-1) {
                res.conformers.push_back(Conformer{std::string(res_name.first, res_name.second), std::string(alt.first, alt.second),
                                                   std::pmr::vector<AtomRecord>(mem)});
                fi = res.conformers.size() - 1;
            }
        }
        std::pmr::vector<AtomRecord> &atoms = res.conformers[fi].atoms;
        if (atoms.capacity() == atoms.size()) atoms.reserve(std::max<size_t>(expect, 2 * atoms.size()));
        atoms.emplace_back();
        last_atoms = &atoms;
        return atoms.back();
    }
    // one more record in the conformer of the previous add() (the caller knows that chain, residue, conformer are the
    // same: a run of records with identical columns 17-27); no searches, no comparisons
    std::pmr::vector<AtomRecord> *last_atoms = nullptr;
    AtomRecord &add_to_last()
    {
        last_atoms->emplace_back();
        return last_atoms->back();
    }
};

inline void set_element(AtomRecord &rec, const TextView &symbol)
{
    rec.element.assign(symbol.first, symbol.second);
    for (auto &c : rec.element) c = (char)std::toupper((unsigned char)c);
    if (rec.element.empty()) rec.element = element_from_name(rec.name);
}

// Atoms without an alternate location belong to EVERY conformer of a residue that has alternate locations
// (a side chain modelled twice on one backbone): the blank conformer's atoms are appended to each of the
// others and the blank conformer goes.  The reference's `residue.conformers().next()` (src/options.rs:162,255)
// then selects backbone + first alternate location, which is what its quality gate over the FreeSASA set
// (tests/quality.rs:17-18, RMSE 43.99 of the reference itself) implies for pdbtbx - whose source is not in
// the reference tree, so the order inside a conformer (its own atoms, then the shared ones) is our choice.
inline void share_blank_conformers(Structure &s)
{
    for (Chain &chain : s.chains)
        for (Residue &res : chain.residues) {
            if (res.conformers.size() < 2) continue;
            size_t blank = res.conformers.size();
            for (size_t k = 0; k < res.conformers.size(); k++)
                if (res.conformers[k].alt_loc.empty()) { blank = k; break; }
            if (blank == res.conformers.size()) continue;
            const Conformer shared = std::move(res.conformers[blank]);
            res.conformers.erase(res.conformers.begin() + (std::ptrdiff_t)blank);
            for (Conformer &c : res.conformers) c.atoms.insert(c.atoms.end(), shared.atoms.begin(), shared.atoms.end());
        }
}

// the next line of `text` from `cur` (advanced past it), without its line end
inline LineView next_line(const char *&cur, const char *end)
{
    const char *nl = (const char *)std::memchr(cur, '\n', (size_t)(end - cur));
    const char *stop = nl ? nl : end;
    LineView line{cur, (size_t)(stop - cur)};
    cur = nl ? nl + 1 : end;
    if (line.n && line.p[line.n - 1] == '\r') line.n--;
    return line;
}

}  // namespace

// ---- Structure: special members (see the header) ----
Structure::Structure(std::size_t pool_bytes)
    : pool_(new std::pmr::monotonic_buffer_resource(std::max<std::size_t>(pool_bytes, 4096))), chains(pool_.get())
{
}
Structure::Structure(const Structure &other) : chains(other.chains.begin(), other.chains.end()), warnings(other.warnings) {}
Structure::Structure(Structure &&other) noexcept
    : pool_(std::move(other.pool_)), chains(std::move(other.chains)), warnings(std::move(other.warnings))
{
    // the moved-from object must not keep an allocator that points into a pool it no longer owns
    other.chains.~vector();
    new (&other.chains) std::pmr::vector<Chain>();
}
Structure &Structure::operator=(const Structure &other)
{
    if (this != &other) {
        Structure copy(other);
        *this = std::move(copy);
    }
    return *this;
}
Structure &Structure::operator=(Structure &&other) noexcept
{
    if (this != &other) {  // (pmr containers do not take the allocator along on assignment: rebuild in place)
        this->~Structure();
        new (this) Structure(std::move(other));
    }
    return *this;
}

// A pool for the model of `text_bytes` of PDB / mmCIF text: an atom record takes 1.5 bytes per byte of an 80-column
// line, residues and conformers a little more.
static inline std::size_t pool_bytes_for(std::size_t text_bytes) { return text_bytes * 2 + 4096; }

// process_files reads a file, selects its atoms and drops the model: unless radii come from the occupancy column,
// nothing ever looks at occupancies or b-factors there, and two of a record's five decimals need not be converted
// (they keep their defaults, 1.0 and 0.0).  Per thread; every other caller of the readers gets full records.
static thread_local bool t_skip_occupancy_and_bfactor = false;

Structure Structure::from_pdb_text(const std::string &text)
{
    Structure s(pool_bytes_for(text.size()));
    ModelBuilder model{s};
    bool in_first_model = true, seen_model = false;
    std::size_t counter = 0, run_left = 0, run_len = 1;
    const char *cur = text.data(), *const end = text.data() + text.size();
    while (cur < end) {
        const LineView line = next_line(cur, end);
        if (line.starts_with("MODEL")) {
            if (seen_model) in_first_model = false;
            seen_model = true;
            continue;
        }
        if (line.starts_with("ENDMDL")) { in_first_model = false; continue; }
        const bool is_atom = line.starts_with("ATOM  "), is_het = line.starts_with("HETATM");
        if (!(is_atom || is_het) || !in_first_model) continue;
        if (line.n < 54) { s.warnings.push_back("short ATOM record skipped"); continue; }
        counter++;
        // records of one conformer usually follow each other: count them (columns 17-27: alternate location, residue
        // name, chain, number, insertion code) so that the conformer's atoms are allocated once, at their final size
        if (run_left == 0) {
            run_left = 1;
            if (line.n >= 27) {
                const char *scan = cur;
                while (scan < end) {
                    const LineView nl = next_line(scan, end);
                    if (nl.starts_with("ANISOU")) continue;  // (between an atom and the next one)
                    if (nl.n < 54 || !(nl.starts_with("ATOM  ") || nl.starts_with("HETATM")) || std::memcmp(nl.p + 16, line.p + 16, 11) != 0) break;
                    run_left++;
                }
            }
            run_len = run_left;
        }
        const bool run_start = run_left == run_len;
        run_left--;
        const TextView name = field_view(line, 13, 16);
        // (the records of a run share columns 17-27: only its first one looks its conformer up)
        AtomRecord &rec = run_start ? model.add(field_view(line, 22, 22), column_int(line, 23, 26, nullptr), field_view(line, 27, 27),
                                                field_view(line, 18, 20), field_view(line, 17, 17), run_len)
                                    : model.add_to_last();
        rec.hetero = is_het;
        bool serial_ok = false;
        const long sv = column_int(line, 7, 11, &serial_ok);
        rec.serial = serial_ok ? (std::size_t)sv : counter;
        rec.name.assign(name.first, name.second);
        rec.x = column_decimal(line, 31, 38, 0.0, 3);  // %8.3f
        rec.y = column_decimal(line, 39, 46, 0.0, 3);
        rec.z = column_decimal(line, 47, 54, 0.0, 3);
        if (!t_skip_occupancy_and_bfactor) {
            rec.occupancy = column_decimal(line, 55, 60, 1.0, 2);  // %6.2f
            rec.b_factor = column_decimal(line, 61, 66, 0.0, 2);
        }
        set_element(rec, field_view(line, 77, 78));
    }
    share_blank_conformers(s);
    return s;
}

namespace {

// mmCIF tokens of one row as views: white-space separated, '...' and "..." quoting (a quote ends
// at the quote character that is followed by white space or the end of the line).
inline void tokenize_views(const LineView &line, std::vector<TextView> &out)
{
    out.clear();
    auto space = [](char c) { return c == ' ' || (c >= '\t' && c <= '\r'); };
    size_t i = 0;
    const size_t n = line.n;
    const char *p = line.p;
    while (i < n) {
        while (i < n && space(p[i])) i++;
        if (i >= n) break;
        if (p[i] == '\'' || p[i] == '"') {
            const char q = p[i++];
            size_t j = i;
            while (j < n && !(p[j] == q && (j + 1 == n || space(p[j + 1])))) j++;
            out.push_back({p + i, j - i});
            i = j + 1;
        } else {
            size_t j = i;
            while (j < n && !space(p[j])) j++;
            out.push_back({p + i, j - i});
            i = j;
        }
    }
}

// strtol / strtoul on a token: optional sign, then digits up to the first other character
inline long token_long(const TextView &t)
{
    const char *p = t.first, *e = t.first + t.second;
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) neg = (*p++ == '-');
    unsigned long v = 0;
    bool over = false;
    for (; p < e && *p >= '0' && *p <= '9'; p++) {
        if (v > (~0ul - 9) / 10) over = true;
        v = v * 10 + (unsigned long)(*p - '0');
    }
    if (over || v > (unsigned long)LONG_MAX) return neg ? LONG_MIN : LONG_MAX;  // strtol saturates
    return neg ? -(long)v : (long)v;
}

}  // namespace

Structure Structure::from_mmcif_text(const std::string &text)
{
    Structure s(pool_bytes_for(text.size()));
    ModelBuilder model{s};
    std::vector<std::string> cols;
    std::vector<TextView> tok;
    bool in_loop = false, in_atom_site = false;
    std::string first_model;
    auto col = [&](const char *name) -> int {
        for (size_t i = 0; i < cols.size(); i++)
            if (cols[i] == name) return (int)i;
        return -1;
    };
    int c_group = -1, c_id = -1, c_sym = -1, c_atom = -1, c_alt = -1, c_comp = -1, c_lasym = -1,
        c_aasym = -1, c_lseq = -1, c_aseq = -1, c_ins = -1, c_x = -1, c_y = -1, c_z = -1, c_occ = -1,
        c_b = -1, c_model = -1;
    bool resolved = false;
    auto val = [&](int c) -> TextView {  // "." and "?" stand for no value
        if (c < 0 || c >= (int)tok.size()) return {"", 0};
        const TextView &t = tok[c];
        return (t.second == 1 && (t.first[0] == '.' || t.first[0] == '?')) ? TextView{t.first, 0} : t;
    };
    const char *cur = text.data(), *const end = text.data() + text.size();
    while (cur < end) {
        const LineView raw = next_line(cur, end);
        const TextView tv = field_view(raw, 1, raw.n);  // trimmed
        if (tv.second == 0) continue;
        const LineView t{tv.first, tv.second};
        if (t.n == 5 && t.starts_with("loop_")) { in_loop = true; in_atom_site = false; cols.clear(); resolved = false; continue; }
        if (t[0] == '#') { in_loop = false; in_atom_site = false; continue; }
        if (in_loop && t[0] == '_') {
            if (t.starts_with("_atom_site.")) {
                in_atom_site = true;
                std::string name(t.p + 11, t.n - 11);
                name = trim(name.substr(0, name.find_first_of(" \t")));
                cols.push_back(name);
            } else {
                in_atom_site = false;
            }
            continue;
        }
        if (!(in_loop && in_atom_site)) continue;
        if (t[0] == '_') { in_loop = false; continue; }
        if (!resolved) {
            c_group = col("group_PDB"); c_id = col("id"); c_sym = col("type_symbol");
            c_atom = col("label_atom_id"); c_alt = col("label_alt_id"); c_comp = col("label_comp_id");
            c_lasym = col("label_asym_id"); c_aasym = col("auth_asym_id"); c_lseq = col("label_seq_id");
            c_aseq = col("auth_seq_id"); c_ins = col("pdbx_PDB_ins_code"); c_x = col("Cartn_x");
            c_y = col("Cartn_y"); c_z = col("Cartn_z"); c_occ = col("occupancy");
            c_b = col("B_iso_or_equiv"); c_model = col("pdbx_PDB_model_num");
            resolved = true;
            if (c_x < 0 || c_y < 0 || c_z < 0 || c_atom < 0 || c_comp < 0)
                throw std::runtime_error("mmCIF _atom_site loop lacks required columns");
        }
        tokenize_views(t, tok);
        if (tok.size() < cols.size()) { s.warnings.push_back("short _atom_site row skipped"); continue; }
        const TextView model_id = val(c_model);
        if (first_model.empty()) first_model = model_id.second ? std::string(model_id.first, model_id.second) : "1";
        if (model_id.second && !same(first_model, model_id)) continue;
        const TextView chain_id = c_aasym >= 0 && val(c_aasym).second ? val(c_aasym) : val(c_lasym);
        const TextView seq = c_aseq >= 0 && val(c_aseq).second ? val(c_aseq) : val(c_lseq);
        const TextView name = val(c_atom), group = val(c_group);
        AtomRecord &rec = model.add(chain_id, token_long(seq), val(c_ins), val(c_comp), val(c_alt));
        rec.hetero = group.second == 6 && std::memcmp(group.first, "HETATM", 6) == 0;
        rec.serial = (std::size_t)token_long(val(c_id));
        rec.name.assign(name.first, name.second);
        auto num = [&](int c, double missing) {
            const TextView v = val(c);
            return v.second ? parse_decimal(v.first, v.second) : missing;
        };
        rec.x = num(c_x, 0.0);
        rec.y = num(c_y, 0.0);
        rec.z = num(c_z, 0.0);
        if (!t_skip_occupancy_and_bfactor) {
            rec.occupancy = num(c_occ, 1.0);
            rec.b_factor = num(c_b, 0.0);
        }
        set_element(rec, val(c_sym));
    }
    share_blank_conformers(s);
    return s;
}

// the whole file in a buffer the thread keeps (no stream, no copies of the text)
static const std::string &read_whole_file(const std::string &path)
{
    static thread_local std::string text;
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    text.clear();
    if (std::fseek(f, 0, SEEK_END) == 0) {
        const long size = std::ftell(f);
        if (size > 0) text.resize((size_t)size);
        std::rewind(f);
    }
    size_t got = 0;
    for (;;) {
        if (got == text.size()) text.resize(std::max<size_t>(2 * text.size(), 1 << 16));  // unknown or growing size
        const size_t n = std::fread(&text[got], 1, text.size() - got, f);
        got += n;
        if (n == 0) break;
    }
    const bool bad = std::ferror(f) != 0;
    std::fclose(f);
    if (bad) throw std::runtime_error("cannot read " + path);
    text.resize(got);
    return text;
}

// fs::write: create or truncate, write all, close (POSIX; short writes continued)
static bool write_whole_file(const std::string &path, const std::string &text, std::string *err)
{
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) {
        *err = "cannot write " + path + ": " + std::strerror(errno);
        return false;
    }
    size_t done = 0;
    while (done < text.size()) {
        const ssize_t n = ::write(fd, text.data() + done, text.size() - done);
        if (n < 0) {
            if (errno == EINTR) continue;
            *err = "cannot write " + path + ": " + std::strerror(errno);
            ::close(fd);
            return false;
        }
        done += (size_t)n;
    }
    if (::close(fd) != 0) {
        *err = "cannot write " + path + ": " + std::strerror(errno);
        return false;
    }
    return true;
}

static bool is_mmcif_path(const std::string &path)
{
    const std::string ext = upper(path.substr(path.find_last_of('.') == std::string::npos ? path.size() : path.find_last_of('.')));
    return ext == ".CIF" || ext == ".MMCIF";
}

Structure Structure::open(const std::string &path)
{
    const std::string &text = read_whole_file(path);
    if (is_mmcif_path(path)) return from_mmcif_text(text);
    return from_pdb_text(text);
}

// ------------------------------------------------------------------ output --

namespace {

void json_string(std::string &o, const std::string &s)
{
    o += '"';
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
        else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o += (char)c;
    }
    o += '"';
}

// shortest decimal that round-trips the f32, as serde_json (ryu) prints an f32: plain decimals with ".0" on whole numbers
// ("200.0", not "2e+02" - which round 5's %g search printed), an exponent without sign or padding outside [1e-5, 1e16)
// ("1e-7"), null for NaN and the infinities.  std::to_chars: 90 ns per value where the %.*g / strtof search took 2.4 us -
// directory mode with per-file JSON output prints a million and a half of them per proteome.
void json_f32(std::string &o, float v)
{
    if (!std::isfinite(v)) { o += "null"; return; }
    if (v == 0.0f) { o += std::signbit(v) ? "-0.0" : "0.0"; return; }
    char b[64];
    const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);  // shortest digits: [-]d[.ddd]e[+-]XX
    *r.ptr = 0;
    const char *p = b;
    if (*p == '-') { o += '-'; p++; }
    const char *e = std::strchr(p, 'e');
    char dig[16];
    int nd = 0;
    for (const char *q = p; q < e; q++)
        if (*q != '.') dig[nd++] = *q;
    const int ex = std::atoi(e + 1);
    if (ex < -5 || ex >= 16) {  // ryu's exponent form: "1e-7", "1.5e20"
        o += dig[0];
        if (nd > 1) { o += '.'; o.append(dig + 1, (size_t)nd - 1); }
        o += 'e';
        o += std::to_string(ex);
        return;
    }
    if (ex < 0) {
        o += "0.";
        o.append((size_t)(-ex - 1), '0');
        o.append(dig, (size_t)nd);
    } else if (ex >= nd - 1) {
        o.append(dig, (size_t)nd);
        o.append((size_t)(ex - (nd - 1)), '0');
        o += ".0";
    } else {
        o.append(dig, (size_t)ex + 1);
        o += '.';
        o.append(dig + ex + 1, (size_t)(nd - ex - 1));
    }
}

}  // namespace

std::string sasa_result_to_json(const std::vector<float> &v)
{
    std::string o = "{\"Atom\":[";
    for (size_t i = 0; i < v.size(); i++) { if (i) o += ','; json_f32(o, v[i]); }
    return o + "]}";
}

std::string sasa_result_to_json(const std::vector<ResidueResult> &v)
{
    std::string o = "{\"Residue\":[";
    for (size_t i = 0; i < v.size(); i++) {
        if (i) o += ',';
        o += "{\"serial_number\":" + std::to_string(v[i].serial_number) + ",\"insertion_code\":";
        json_string(o, v[i].insertion_code);
        o += ",\"value\":";
        json_f32(o, v[i].value);
        o += ",\"name\":";
        json_string(o, v[i].name);
        o += std::string(",\"is_polar\":") + (v[i].is_polar ? "true" : "false") + ",\"chain_id\":";
        json_string(o, v[i].chain_id);
        o += '}';
    }
    return o + "]}";
}

std::string sasa_result_to_json(const std::vector<ChainResult> &v)
{
    std::string o = "{\"Chain\":[";
    for (size_t i = 0; i < v.size(); i++) {
        if (i) o += ',';
        o += "{\"name\":";
        json_string(o, v[i].name);
        o += ",\"value\":";
        json_f32(o, v[i].value);
        o += '}';
    }
    return o + "]}";
}

std::string sasa_result_to_json(const ProteinResult &v)
{
    std::string o = "{\"Protein\":{\"global_total\":";
    json_f32(o, v.global_total);
    o += ",\"polar_total\":";
    json_f32(o, v.polar_total);
    o += ",\"non_polar_total\":";
    json_f32(o, v.non_polar_total);
    return o + "}}";
}

// io.rs:25-30: every atom of the model, in order, takes v[i]
bool sasa_result_to_protein_object(Structure &pdb, const std::vector<float> &v, std::string *err)
{
    if (pdb.atom_count() != v.size()) {
        if (err) *err = "atom-level result has " + std::to_string(v.size()) + " values for " +
                        std::to_string(pdb.atom_count()) + " atoms (filtered atoms cannot be mapped back)";
        return false;
    }
    size_t i = 0;
    for (auto &c : pdb.chains)
        for (auto &r : c.residues)
            for (auto &f : r.conformers)
                for (auto &a : f.atoms) a.b_factor = (double)v[i++];
    return true;
}

// io.rs:31-43
bool sasa_result_to_protein_object(Structure &pdb, const std::vector<ResidueResult> &v, std::string *err)
{
    size_t i = 0;
    for (auto &c : pdb.chains)
        for (auto &r : c.residues) {
            if (i >= v.size() || v[i].serial_number != r.serial_number) {
                if (err) *err = "residue-level result does not line up with the structure";
                return false;
            }
            for (auto &f : r.conformers)
                for (auto &a : f.atoms) a.b_factor = (double)v[i].value;
            i++;
        }
    return true;
}

// io.rs:44-54
bool sasa_result_to_protein_object(Structure &pdb, const std::vector<ChainResult> &v, std::string *err)
{
    if (v.size() != pdb.chains.size()) {
        if (err) *err = "chain-level result does not line up with the structure";
        return false;
    }
    for (size_t i = 0; i < pdb.chains.size(); i++) {
        if (v[i].name != pdb.chains[i].id) {
            if (err) *err = "chain-level result does not line up with the structure";
            return false;
        }
        for (auto &r : pdb.chains[i].residues)
            for (auto &f : r.conformers)
                for (auto &a : f.atoms) a.b_factor = (double)v[i].value;
    }
    return true;
}

// io.rs:55-61
bool sasa_result_to_protein_object(Structure &pdb, const ProteinResult &v, std::string *)
{
    for (auto &c : pdb.chains)
        for (auto &r : c.residues)
            for (auto &f : r.conformers)
                for (auto &a : f.atoms) a.b_factor = (double)v.global_total;
    return true;
}

std::string Structure::to_pdb_text() const
{
    std::string o;
    char line[96];
    for (const auto &c : chains) {
        for (const auto &r : c.residues)
            for (const auto &f : r.conformers)
                for (const auto &a : f.atoms) {
                    // atom names shorter than four characters start in column 14 unless the
                    // element symbol has two letters
                    std::string name = a.name;
                    if (name.size() < 4 && a.element.size() < 2) name = " " + name;
                    std::snprintf(line, sizeof line,
                                  "%-6s%5zu %-4s%1s%3s %1s%4lld%1s   %8.3f%8.3f%8.3f%6.2f%6.2f          %2s  \n",
                                  a.hetero ? "HETATM" : "ATOM", a.serial % 100000, name.c_str(),
                                  f.alt_loc.substr(0, 1).c_str(), f.name.substr(0, 3).c_str(),
                                  c.id.substr(0, 1).c_str(), (long long)(r.serial_number % 10000),
                                  r.insertion_code.substr(0, 1).c_str(), a.x, a.y, a.z, a.occupancy,
                                  a.b_factor, a.element.c_str());
                    o += line;
                }
        o += "TER\n";
    }
    o += "END\n";
    return o;
}

void Structure::save_pdb(const std::string &path) const
{
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    const std::string t = to_pdb_text();
    f.write(t.data(), (std::streamsize)t.size());
    if (!f) throw std::runtime_error("write failed: " + path);
}

// ------------------------------------------------------------------ levels --

namespace detail {

namespace {

struct BuildError {
    SASACalcError error = SASACalcError::Ok;
    std::string message;
};

// build_atom! (options.rs:81-116)
bool build_atom(const OptionValues &o, const AtomRecord &atom, const std::string &residue_name,
                std::uint64_t id, std::vector<rsasa_atom_t> &atoms, BuildError &err)
{
    float radius = 0.f;
    if (o.read_radii_from_occupancy) {
        radius = (float)atom.occupancy;
    } else {
        bool found = false;
        if (o.radii_config) {  // utils.rs:45-53
            auto r = o.radii_config->find(residue_name);
            if (r != o.radii_config->end()) {
                auto a = r->second.find(atom.name);
                if (a != r->second.end()) { radius = a->second; found = true; }
            }
        }
        if (!found) found = get_protor_radius(residue_name, atom.name, &radius);
        if (!found) {
            if (o.allow_vdw_fallback) {
                if (!vdw_radius(atom.element, &radius)) {
                    err = {SASACalcError::VanDerWaalsMissing, "Van der Waals radius missing for element"};
                    return false;
                }
            } else {
                err = {SASACalcError::RadiusMissing,
                       "Radius not found for residue '" + residue_name + "' atom '" + atom.name +
                           "' of type '" + atom.element +
                           "'. This error can can be ignored, if you are using the CLI pass "
                           "--allow-vdw-fallback or use with_allow_vdw_fallback if you are using the API."};
                return false;
            }
        }
    }
    rsasa_atom_t a;
    a.position[0] = (float)atom.x;  // options.rs:106-110: f64 -> f32
    a.position[1] = (float)atom.y;
    a.position[2] = (float)atom.z;
    a.radius = radius;
    a.id = id;
    atoms.push_back(a);
    return true;
}

// The atom loop shared by all four build_atoms_and_mapping bodies: chains ->
// residues -> FIRST conformer -> atoms, hydrogen / HETATM filters.  `per_residue`
// is told each residue's range of kept atoms.
static const std::string kNoAltLoc;

template <typename F>
bool select_atoms(const Structure &pdb, const OptionValues &o, bool id_uses_altloc,
                  std::vector<rsasa_atom_t> &atoms, BuildError &err, F per_residue)
{
    atoms.reserve(atoms.size() + pdb.atom_count());  // (an upper bound: one allocation instead of a dozen regrowths)
    for (size_t ci = 0; ci < pdb.chains.size(); ci++) {
        const Chain &chain = pdb.chains[ci];
        for (size_t ri = 0; ri < chain.residues.size(); ri++) {
            const Residue &res = chain.residues[ri];
            // Residue::name(): Some(name) iff every conformer has the same name (options.rs:161,248); no copy
            bool same_name = !res.conformers.empty();
            for (size_t k = 1; k < res.conformers.size() && same_name; k++) same_name = res.conformers[k].name == res.conformers[0].name;
            if (!same_name) {
                err = {SASACalcError::FailedToGetResidueName, "Failed to get residue name"};
                return false;
            }
            const std::string &residue_name = res.conformers[0].name;
            const size_t begin = atoms.size();
            if (!res.conformers.empty()) {
                const Conformer &conf = res.conformers.front();  // conformers().next()
                for (const AtomRecord &atom : conf.atoms) {
                    if (atom.element.empty()) {  // options.rs:164
                        err = {SASACalcError::ElementMissing, "Element missing for atom"};
                        return false;
                    }
                    if (!o.include_hydrogens && atom.element.size() == 1 && atom.element[0] == 'H') continue;  // options.rs:166
                    if (atom.hetero && !o.include_hetatms) continue;           // options.rs:169
                    const std::uint64_t id = fnv_hash_altloc_serial(id_uses_altloc ? conf.alt_loc : kNoAltLoc, atom.serial);
                    if (!build_atom(o, atom, residue_name, id, atoms, err)) return false;
                }
            }
            per_residue(ci, ri, begin, atoms.size());
        }
    }
    return true;
}

// What one structure contributes to a batch: its kept atoms and the contiguous atom
// segments (residues or chains, by level) whose sequential f32 sums the level reports.
struct Prepared {
    BuildError err;
    std::vector<rsasa_atom_t> atoms;
    std::vector<uint32_t> seg_end;  // end offset (in `atoms`) of each segment, in output order
};

// build_atoms_and_mapping of each level (options.rs:151-189, 234-286, 317-364, 412-463)
template <typename Level>
Prepared prepare(const Structure &pdb, const OptionValues &o);

template <>
Prepared prepare<AtomLevel>(const Structure &pdb, const OptionValues &o)
{
    Prepared p;
    select_atoms(pdb, o, true, p.atoms, p.err, [](size_t, size_t, size_t, size_t) {});
    return p;
}

template <>
Prepared prepare<ResidueLevel>(const Structure &pdb, const OptionValues &o)
{
    Prepared p;
    select_atoms(pdb, o, true, p.atoms, p.err,
                 [&](size_t, size_t, size_t, size_t end) { p.seg_end.push_back((uint32_t)end); });
    return p;
}

template <>
Prepared prepare<ChainLevel>(const Structure &pdb, const OptionValues &o)
{
    Prepared p;
    std::vector<uint32_t> chain_end(pdb.chains.size(), 0u);
    select_atoms(pdb, o, true, p.atoms, p.err,
                 [&](size_t ci, size_t, size_t, size_t end) { chain_end[ci] = (uint32_t)end; });
    // kept atoms of a chain are contiguous; a chain without residues owns an empty range
    uint32_t prev = 0;
    for (size_t ci = 0; ci < pdb.chains.size(); ci++) {
        prev = std::max(prev, chain_end[ci]);
        p.seg_end.push_back(prev);
    }
    return p;
}

template <>
Prepared prepare<ProteinLevel>(const Structure &pdb, const OptionValues &o)
{
    Prepared p;  // ids ignore the alt-loc (options.rs:453)
    select_atoms(pdb, o, false, p.atoms, p.err,
                 [&](size_t, size_t, size_t, size_t end) { p.seg_end.push_back((uint32_t)end); });
    return p;
}

// ---- directory mode's short cut: PDB text -> kept atoms, without the atom records of the model ----
// process_files reads a file, selects its atoms and throws the model away; building 112-byte atom records with two
// strings each only to walk them once is most of its CPU time.  For PLAIN files - one conformer per residue (no
// alternate-location characters), the records of a chain and of a residue contiguous and in ascending residue order,
// every radius found - this reads the text once and writes what prepare<Level> would have written: the kept atoms
// (same order, same f64 -> f32 coordinates, same radii, same ids) and the segment ends, plus a model WITHOUT atoms
// (chains, residues, one named conformer each) for the result's metadata.  Anything else - an alt-loc, a chain or
// residue that comes back, a short record, a missing radius or element, a custom radii table - returns false and the
// file takes the general reader, which also produces the errors.  tests/test_host_api.py compares the two on every
// fixture and on mutated files (`sasa_host_cli prepare`); RSASA_NO_FAST_READER=1 switches it off.
enum class SegKind { None, Residue, Chain };
template <typename Level> struct seg_kind_of;
template <> struct seg_kind_of<AtomLevel> { static constexpr SegKind value = SegKind::None; };
template <> struct seg_kind_of<ResidueLevel> { static constexpr SegKind value = SegKind::Residue; };
template <> struct seg_kind_of<ChainLevel> { static constexpr SegKind value = SegKind::Chain; };
template <> struct seg_kind_of<ProteinLevel> { static constexpr SegKind value = SegKind::Residue; };

bool fast_pdb_prepare(const std::string &text, const OptionValues &o, SegKind kind, Structure &light, Prepared &p)
{
    if (o.radii_config) return false;
    const ProtorFlat &protor = protor_flat();
    std::pmr::memory_resource *const mem = light.chains.get_allocator().resource();
    p.atoms.clear();
    p.seg_end.clear();
    p.atoms.reserve(text.size() / 80 + 1);
    bool in_first_model = true, seen_model = false;
    std::size_t counter = 0;
    const char *key = nullptr;             // columns 17-27 of the current residue's records
    std::int64_t res_seq = 0;              // the current residue (of the current chain)
    const char *cur = text.data(), *const end = text.data() + text.size();
    while (cur < end) {
        const LineView line = next_line(cur, end);
        const bool is_atom = line.n >= 6 && std::memcmp(line.p, "ATOM  ", 6) == 0;
        const bool is_het = !is_atom && line.n >= 6 && std::memcmp(line.p, "HETATM", 6) == 0;
        if (!(is_atom || is_het)) {
            if (line.starts_with("MODEL")) {
                if (seen_model) in_first_model = false;
                seen_model = true;
            } else if (line.starts_with("ENDMDL")) {
                in_first_model = false;
            }
            continue;
        }
        if (!in_first_model) continue;
        if (line.n < 54 || line.p[16] != ' ') return false;  // short record / alternate location: the general reader
        counter++;
        if (!key || std::memcmp(line.p + 16, key, 11) != 0) {
            // a new residue: it must continue the current chain in ascending order, or open a chain not seen before
            const TextView chain_id = field_view(line, 22, 22), icode = field_view(line, 27, 27), res_name = field_view(line, 18, 20);
            const std::int64_t seq = column_int(line, 23, 26, nullptr);
            const bool same_chain = !light.chains.empty() && same(light.chains.back().id, chain_id);
            if (same_chain) {
                const Residue &prev = light.chains.back().residues.back();
                const int c = prev.insertion_code.compare(0, std::string::npos, icode.first, icode.second);
                if (!(seq > res_seq || (seq == res_seq && c < 0))) return false;
            } else {
                for (const Chain &ch : light.chains)
                    if (same(ch.id, chain_id)) return false;
                if (kind == SegKind::Chain && !light.chains.empty()) p.seg_end.push_back((uint32_t)p.atoms.size());
                light.chains.push_back(Chain{std::string(chain_id.first, chain_id.second), std::pmr::vector<Residue>(mem)});
            }
            if (kind == SegKind::Residue && key) p.seg_end.push_back((uint32_t)p.atoms.size());
            Chain &chain = light.chains.back();
            chain.residues.push_back(Residue{seq, std::string(icode.first, icode.second), std::pmr::vector<Conformer>(mem)});
            chain.residues.back().conformers.push_back(
                Conformer{std::string(res_name.first, res_name.second), std::string(), std::pmr::vector<AtomRecord>(mem)});
            key = line.p + 16;
            res_seq = seq;
        }
        // element (columns 77-78, upper case; else the first letter of the name), hydrogen / HETATM filters
        const TextView name = field_view(line, 13, 16), sym = field_view(line, 77, 78);
        char e0 = 0, e1 = 0;
        if (sym.second) {
            e0 = (char)std::toupper((unsigned char)sym.first[0]);
            if (sym.second > 1) e1 = (char)std::toupper((unsigned char)sym.first[1]);
        } else {
            for (size_t k = 0; k < name.second && !e0; k++)
                if (std::isalpha((unsigned char)name.first[k])) e0 = (char)std::toupper((unsigned char)name.first[k]);
        }
        if (!e0) return false;                                            // options.rs:164 (the general path reports it)
        if (!o.include_hydrogens && e0 == 'H' && !e1) continue;          // options.rs:166
        if (is_het && !o.include_hetatms) continue;                      // options.rs:169
        float radius = 0.f;
        if (o.read_radii_from_occupancy) {
            radius = (float)column_decimal(line, 55, 60, 1.0, 2);        // options.rs:83-84
        } else {
            const Conformer &conf = light.chains.back().residues.back().conformers.front();
            if (!protor.find(conf.name.data(), conf.name.size(), name.first, name.second, &radius)) {
                if (!o.allow_vdw_fallback) return false;
                const std::string el = e1 ? std::string{e0, e1} : std::string(1, e0);
                if (!vdw_radius(el, &radius)) return false;
            }
        }
        bool serial_ok = false;
        const long sv = column_int(line, 7, 11, &serial_ok);
        rsasa_atom_t a;
        a.position[0] = (float)column_decimal(line, 31, 38, 0.0, 3);     // options.rs:106-110: f64 -> f32
        a.position[1] = (float)column_decimal(line, 39, 46, 0.0, 3);
        a.position[2] = (float)column_decimal(line, 47, 54, 0.0, 3);
        a.radius = radius;
        a.id = fnv_hash_altloc_serial(kNoAltLoc, serial_ok ? (std::size_t)sv : counter);
        p.atoms.push_back(a);
    }
    if (kind == SegKind::Residue && key) p.seg_end.push_back((uint32_t)p.atoms.size());
    if (kind == SegKind::Chain && !light.chains.empty()) p.seg_end.push_back((uint32_t)p.atoms.size());
    return true;
}


// The same short cut for mmCIF text (AlphaFold's files: one `_atom_site` loop, a row per line, no alternate locations):
// the rows' tokens are looked at in place, the kept atoms written as prepare<Level> would have written them, and the
// model keeps chains, residues and one named conformer each.  The exits are the PDB short cut's: an alternate location,
// a chain or residue that comes back or goes down, a second name inside a residue, a row that is not exactly the
// header's columns, a missing element or radius, a second `_atom_site` loop - the general reader takes the file.
// white space as tokenize_views sees it, one table look-up per character
struct SpaceTable {
    bool is[256] = {};
    SpaceTable() { is[(unsigned char)' '] = true; for (int c = '\t'; c <= '\r'; c++) is[c] = true; }
};
static const SpaceTable kSpaceTable;

// tokenize_views into a fixed array (no allocation, no bounds growth); returns the number of tokens, cap + 1 if there
// are more than cap
inline size_t tokenize_row(const char *p, size_t n, TextView *out, size_t cap)
{
    const bool *const sp = kSpaceTable.is;
    size_t i = 0, k = 0;
    while (i < n) {
        while (i < n && sp[(unsigned char)p[i]]) i++;
        if (i >= n) break;
        if (k == cap) return cap + 1;
        if (p[i] == '\'' || p[i] == '"') {
            const char q = p[i++];
            size_t j = i;
            while (j < n && !(p[j] == q && (j + 1 == n || sp[(unsigned char)p[j + 1]]))) j++;
            out[k++] = {p + i, j - i};
            i = j + 1;
        } else {
            size_t j = i + 1;
            while (j < n && !sp[(unsigned char)p[j]]) j++;
            out[k++] = {p + i, j - i};
            i = j;
        }
    }
    return k;
}

// The rows of an `_atom_site` loop are written in aligned columns by most writers (AlphaFold's among them): nearly every
// row has its separators where the row before had them.  RowSplitter keeps the last row's separator mask and token
// bounds; a row with the same mask gets its tokens by adding offsets.  The mask is built sixteen bytes at a time (SSE2);
// rows that hold a quote, a tab or another control character, a byte above 127, or more than 256 bytes - and builds
// without SSE2 - go through tokenize_row.  Either way the tokens are tokenize_views'.
struct RowSplitter {
    static constexpr size_t kMaxCols = 64;
    const char *row = nullptr;          // the current row; token k is row + off[k], len[k] bytes
    uint16_t off[kMaxCols], len[kMaxCols];
    size_t n_tok = 0;
#ifdef RSASA_ROW_SSE2
    uint64_t last_mask[4] = {~0ull, ~0ull, ~0ull, ~0ull};  // bit i: byte i is a blank (or beyond the row's end)
    bool have_last = false;
#endif
    TextView token(size_t k) const { return {row + off[k], len[k]}; }

    // false: more than kMaxCols tokens (or a row too long for 16-bit offsets)
    bool split(const char *p, size_t n, const char *text_end)
    {
        row = p;
#ifdef RSASA_ROW_SSE2
        if (n <= 256) {
            uint64_t m[4] = {~0ull, ~0ull, ~0ull, ~0ull};
            const __m128i blank = _mm_set1_epi8(' '), q1 = _mm_set1_epi8('\''), q2 = _mm_set1_epi8('"');
            unsigned odd = 0;  // quotes, control characters, bytes above 127 (signed compare: below ' ')
            for (size_t b = 0; b < n; b += 16) {
                __m128i v;
                if (p + b + 16 <= text_end) {
                    v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + b));
                } else {
                    char tmp[16] = {' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' ', ' '};
                    std::memcpy(tmp, p + b, (size_t)(text_end - (p + b)));
                    v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(tmp));
                }
                uint32_t sp = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(v, blank));
                uint32_t bad = (uint32_t)_mm_movemask_epi8(_mm_or_si128(_mm_cmplt_epi8(v, blank), _mm_or_si128(_mm_cmpeq_epi8(v, q1), _mm_cmpeq_epi8(v, q2))));
                if (n - b < 16) {  // bytes beyond the row: blanks, whatever they are
                    const uint32_t in_row = (1u << (n - b)) - 1u;
                    sp |= ~in_row & 0xFFFFu;
                    bad &= in_row;
                }
                odd |= bad;
                m[b >> 6] = (m[b >> 6] & ~(0xFFFFull << (b & 63))) | ((uint64_t)sp << (b & 63));
            }
            if (!odd) {
                if (have_last && m[0] == last_mask[0] && m[1] == last_mask[1] && m[2] == last_mask[2] && m[3] == last_mask[3])
                    return true;  // the previous row's columns: off / len stand
                // token bounds from the mask: every change of state, in order
                size_t k = 0;
                bool in_tok = false;
                size_t start = 0;
                uint64_t carry = 1;  // (the byte before the row counts as a blank)
                for (size_t w = 0; w * 64 < n; w++) {
                    const uint64_t sp = m[w];
                    uint64_t edges = sp ^ ((sp << 1) | carry);
                    carry = sp >> 63;
                    while (edges) {
                        const size_t pos = w * 64 + (size_t)__builtin_ctzll(edges);
                        edges &= edges - 1;
                        if (!in_tok) {
                            if (k == kMaxCols) return false;
                            start = pos;
                        } else {
                            off[k] = (uint16_t)start; len[k] = (uint16_t)(pos - start);
                            k++;
                        }
                        in_tok = !in_tok;
                    }
                }
                if (in_tok) {  // (n is a multiple of 64 and the last token ends with the row)
                    off[k] = (uint16_t)start; len[k] = (uint16_t)(n - start);
                    k++;
                }
                n_tok = k;
                for (int w = 0; w < 4; w++) last_mask[w] = m[w];
                have_last = true;
                return true;
            }
            have_last = false;
        }
#endif
        (void)text_end;
#ifdef RSASA_ROW_SSE2
        have_last = false;  // (off / len are overwritten below: the remembered mask no longer describes them)
#endif
        if (n > 0xFFFFu) return false;
        TextView tmp[kMaxCols];
        n_tok = tokenize_row(p, n, tmp, kMaxCols);
        if (n_tok > kMaxCols) return false;
        for (size_t k = 0; k < n_tok; k++) { off[k] = (uint16_t)(tmp[k].first - p); len[k] = (uint16_t)tmp[k].second; }
        return true;
    }
};

bool fast_cif_prepare(const std::string &text, const OptionValues &o, SegKind kind, Structure &light, Prepared &p)
{
    if (o.radii_config) return false;
    const ProtorFlat &protor = protor_flat();
    std::pmr::memory_resource *const mem = light.chains.get_allocator().resource();
    p.atoms.clear();
    p.seg_end.clear();
    p.atoms.reserve(text.size() / 88 + 1);
    std::vector<std::string> cols;
    constexpr size_t kMaxCols = RowSplitter::kMaxCols;
    RowSplitter rows;
    bool in_loop = false, in_atom_site = false, resolved = false, had_rows = false;
    std::string first_model;
    int c_group = -1, c_id = -1, c_sym = -1, c_atom = -1, c_alt = -1, c_comp = -1, c_lasym = -1, c_aasym = -1, c_lseq = -1,
        c_aseq = -1, c_ins = -1, c_x = -1, c_y = -1, c_z = -1, c_occ = -1, c_model = -1;
    auto col = [&](const char *name) -> int {
        for (size_t i = 0; i < cols.size(); i++)
            if (cols[i] == name) return (int)i;
        return -1;
    };
    auto val = [&](int c) -> TextView {  // "." and "?" stand for no value
        if (c < 0) return {"", 0};
        const TextView t = rows.token((size_t)c);
        return (t.second == 1 && (t.first[0] == '.' || t.first[0] == '?')) ? TextView{t.first, 0} : t;
    };
    auto eq = [](const TextView &a, const TextView &b) { return a.second == b.second && std::memcmp(a.first, b.first, a.second) == 0; };
    // the current residue, as views of its first row (the model's strings hold the same characters)
    bool have_res = false;
    std::int64_t res_seq = 0;
    TextView res_chain{"", 0}, res_icode{"", 0}, res_comp{"", 0};
    const char *cur = text.data(), *const end = text.data() + text.size();
    while (cur < end) {
        const LineView raw = next_line(cur, end);
        const TextView tv = field_view(raw, 1, raw.n);  // trimmed
        if (tv.second == 0) continue;
        const LineView t{tv.first, tv.second};
        if (t[0] == 'l' && t.n == 5 && t.starts_with("loop_")) { in_loop = true; in_atom_site = false; cols.clear(); resolved = false; continue; }
        if (t[0] == '#') { in_loop = false; in_atom_site = false; continue; }
        if (t[0] == '_') {
            if (!in_loop) continue;
            if (t.starts_with("_atom_site.")) {
                if (had_rows) return false;  // a second loop of atoms
                in_atom_site = true;
                std::string name(t.p + 11, t.n - 11);
                name = trim(name.substr(0, name.find_first_of(" \t")));
                cols.push_back(name);
            } else {
                in_atom_site = false;
            }
            continue;
        }
        if (!(in_loop && in_atom_site)) continue;
        if (!resolved) {
            c_group = col("group_PDB"); c_id = col("id"); c_sym = col("type_symbol");
            c_atom = col("label_atom_id"); c_alt = col("label_alt_id"); c_comp = col("label_comp_id");
            c_lasym = col("label_asym_id"); c_aasym = col("auth_asym_id"); c_lseq = col("label_seq_id");
            c_aseq = col("auth_seq_id"); c_ins = col("pdbx_PDB_ins_code"); c_x = col("Cartn_x");
            c_y = col("Cartn_y"); c_z = col("Cartn_z"); c_occ = col("occupancy"); c_model = col("pdbx_PDB_model_num");
            resolved = true;
            if (c_x < 0 || c_y < 0 || c_z < 0 || c_atom < 0 || c_comp < 0) return false;  // (the general reader reports it)
            if (cols.size() > kMaxCols) return false;
        }
        if (!rows.split(t.p, t.n, end) || rows.n_tok != cols.size()) return false;
        had_rows = true;
        const TextView model_id = val(c_model);
        if (first_model.empty()) first_model = model_id.second ? std::string(model_id.first, model_id.second) : "1";
        if (model_id.second && !same(first_model, model_id)) continue;
        if (val(c_alt).second) return false;
        TextView chain_id = val(c_aasym);
        if (!chain_id.second) chain_id = val(c_lasym);
        TextView seq_t = val(c_aseq);
        if (!seq_t.second) seq_t = val(c_lseq);
        const std::int64_t seq = token_long(seq_t);
        const TextView icode = val(c_ins), comp = val(c_comp), name = val(c_atom), group = val(c_group);
        const bool same_chain = have_res && eq(res_chain, chain_id);
        const bool same_res = same_chain && seq == res_seq && eq(res_icode, icode);
        if (same_res) {
            if (!eq(res_comp, comp)) return false;  // a second conformer
        } else {
            if (same_chain) {
                const int c = std::string_view(res_icode.first, res_icode.second).compare(std::string_view(icode.first, icode.second));
                if (!(seq > res_seq || (seq == res_seq && c < 0))) return false;
            } else {
                for (const Chain &ch : light.chains)
                    if (same(ch.id, chain_id)) return false;
                if (kind == SegKind::Chain && !light.chains.empty()) p.seg_end.push_back((uint32_t)p.atoms.size());
                light.chains.push_back(Chain{std::string(chain_id.first, chain_id.second), std::pmr::vector<Residue>(mem)});
            }
            if (kind == SegKind::Residue && have_res) p.seg_end.push_back((uint32_t)p.atoms.size());
            Chain &chain = light.chains.back();
            chain.residues.push_back(Residue{seq, std::string(icode.first, icode.second), std::pmr::vector<Conformer>(mem)});
            chain.residues.back().conformers.push_back(
                Conformer{std::string(comp.first, comp.second), std::string(), std::pmr::vector<AtomRecord>(mem)});
            have_res = true;
            res_seq = seq;
            res_chain = chain_id; res_icode = icode; res_comp = comp;
        }
        // element: the symbol in upper case, else the first letter of the name (set_element); hydrogen / HETATM filters
        const TextView sym = val(c_sym);
        char e0 = 0;
        bool one_letter = true;
        if (sym.second) {
            e0 = (char)std::toupper((unsigned char)sym.first[0]);
            one_letter = sym.second == 1;
        } else {
            for (size_t k = 0; k < name.second && !e0; k++)
                if (std::isalpha((unsigned char)name.first[k])) e0 = (char)std::toupper((unsigned char)name.first[k]);
        }
        if (!e0) return false;                                            // options.rs:164 (the general path reports it)
        if (!o.include_hydrogens && e0 == 'H' && one_letter) continue;   // options.rs:166
        const bool is_het = group.second == 6 && std::memcmp(group.first, "HETATM", 6) == 0;
        if (is_het && !o.include_hetatms) continue;                      // options.rs:169
        float radius = 0.f;
        if (o.read_radii_from_occupancy) {
            const TextView occ = val(c_occ);
            radius = (float)(occ.second ? parse_decimal(occ.first, occ.second) : 1.0);  // options.rs:83-84
        } else if (!protor.find(comp.first, comp.second, name.first, name.second, &radius)) {
            if (!o.allow_vdw_fallback) return false;
            std::string el = sym.second ? std::string(sym.first, sym.second) : std::string(1, e0);
            for (auto &c : el) c = (char)std::toupper((unsigned char)c);
            if (!vdw_radius(el, &radius)) return false;
        }
        auto num = [&](int c) {
            const TextView v = val(c);
            return v.second ? parse_decimal(v.first, v.second) : 0.0;
        };
        rsasa_atom_t a;
        a.position[0] = (float)num(c_x);                                  // options.rs:106-110: f64 -> f32
        a.position[1] = (float)num(c_y);
        a.position[2] = (float)num(c_z);
        a.radius = radius;
        a.id = fnv_hash_altloc_serial(kNoAltLoc, (std::size_t)token_long(val(c_id)));
        p.atoms.push_back(a);
    }
    if (kind == SegKind::Residue && have_res) p.seg_end.push_back((uint32_t)p.atoms.size());
    if (kind == SegKind::Chain && !light.chains.empty()) p.seg_end.push_back((uint32_t)p.atoms.size());
    return true;
}

}  // namespace

// Measurement hook (sasa_host_cli prepare-bench-*): seconds per pass of directory mode's per-file work - text to kept
// atoms - through the short cut or the general reader, the file read once; no GPU.
double debug_prepare_seconds(const std::string &path, const OptionValues &o, int level, bool fast, int reps, size_t *n_atoms, bool *used_fast)
{
    const std::string text = read_whole_file(path);
    const bool cif = is_mmcif_path(path);
    const SegKind kinds[] = {SegKind::None, SegKind::Residue, SegKind::Chain, SegKind::Residue};
    size_t atoms = 0;
    bool took_fast = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) {
        Prepared p;
        bool ok = false;
        if (fast) {
            Structure light(text.size() / 8 + 4096);
            ok = cif ? fast_cif_prepare(text, o, kinds[level], light, p) : fast_pdb_prepare(text, o, kinds[level], light, p);
        }
        took_fast = ok;
        if (!ok) {
            t_skip_occupancy_and_bfactor = !o.read_radii_from_occupancy;
            const Structure model = cif ? Structure::from_mmcif_text(text) : Structure::from_pdb_text(text);
            t_skip_occupancy_and_bfactor = false;
            p = level == 0 ? prepare<AtomLevel>(model, o) : level == 1 ? prepare<ResidueLevel>(model, o)
              : level == 2 ? prepare<ChainLevel>(model, o) : prepare<ProteinLevel>(model, o);
        }
        atoms = p.atoms.size();
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps > 0 ? reps : 1);
    if (n_atoms) *n_atoms = atoms;
    if (used_fast) *used_fast = took_fast;
    return s;
}

// Test hook (sasa_host_cli prepare): what directory mode hands to the GPU for one file - kept atoms, segment ends - and
// the model its results take their metadata from, through the short cut (`fast`, when the file qualifies) or the
// general reader.  JSON text; no GPU.
std::string debug_prepare_json(const std::string &path, const OptionValues &o, int level, bool fast)
{
    const std::string &text = read_whole_file(path);
    Structure model;
    Prepared p;
    bool used_fast = false;
    const SegKind kinds[] = {SegKind::None, SegKind::Residue, SegKind::Chain, SegKind::Residue};
    if (fast) {
        Structure light(text.size() / 8 + 4096);
        if (is_mmcif_path(path) ? fast_cif_prepare(text, o, kinds[level], light, p) : fast_pdb_prepare(text, o, kinds[level], light, p)) {
            model = std::move(light);
            used_fast = true;
        }
    }
    if (!used_fast) {
        model = is_mmcif_path(path) ? Structure::from_mmcif_text(text) : Structure::from_pdb_text(text);
        p = level == 0 ? prepare<AtomLevel>(model, o) : level == 1 ? prepare<ResidueLevel>(model, o)
          : level == 2 ? prepare<ChainLevel>(model, o) : prepare<ProteinLevel>(model, o);
    }
    char buf[160];
    std::string out = std::string("{\"fast\":") + (used_fast ? "true" : "false") + ",\"error\":" + std::to_string((int)p.err.error) + ",\"atoms\":[";
    for (size_t i = 0; i < p.atoms.size(); i++) {
        const rsasa_atom_t &a = p.atoms[i];
        std::snprintf(buf, sizeof buf, "%s[\"%a\",\"%a\",\"%a\",\"%a\",\"%llu\"]", i ? "," : "", a.position[0], a.position[1], a.position[2],
                      a.radius, (unsigned long long)a.id);
        out += buf;
    }
    out += "],\"seg_end\":[";
    for (size_t i = 0; i < p.seg_end.size(); i++) out += (i ? "," : "") + std::to_string(p.seg_end[i]);
    out += "],\"model\":[";
    for (size_t c = 0; c < model.chains.size(); c++) {
        out += (c ? ",[\"" : "[\"") + model.chains[c].id + "\",[";
        for (size_t r = 0; r < model.chains[c].residues.size(); r++) {
            const Residue &res = model.chains[c].residues[r];
            std::string name;
            const bool named = res.name(&name);
            out += (r ? ",[" : "[") + std::to_string(res.serial_number) + ",\"" + res.insertion_code + "\",\"" + (named ? name : std::string("?")) + "\"]";
        }
        out += "]]";
    }
    out += "]}";
    return out;
}

// (public, see the header) the ChainLevel selection of one structure without any GPU work
Result<SelectedAtoms> select_by_chain(const Structure &pdb, const OptionValues &o)
{
    Result<SelectedAtoms> r;
    Prepared p = prepare<ChainLevel>(pdb, o);
    if (p.err.error != SASACalcError::Ok) {
        r.error = p.err.error;
        r.message = p.err.message;
        return r;
    }
    r.value.atoms = std::move(p.atoms);
    r.value.chain_end = std::move(p.seg_end);
    for (const Chain &c : pdb.chains) r.value.chain_ids.push_back(c.id);
    return r;
}

namespace {

// process_atoms of each level (options.rs:142-149, 195-232, 292-315, 370-410).  `atom` and
// `seg` are this structure's slices of the batch results; `global` is the sequential f32 sum
// over all its atoms (used by ProteinLevel only).
template <typename Level>
typename Level::Output finish(const Structure &pdb, const float *atom, size_t n_atoms,
                              const float *seg, float global);

template <>
std::vector<float> finish<AtomLevel>(const Structure &, const float *atom, size_t n, const float *, float)
{
    return std::vector<float>(atom, atom + n);
}

template <>
std::vector<ResidueResult> finish<ResidueLevel>(const Structure &pdb, const float *, size_t,
                                                const float *seg, float)
{
    std::vector<ResidueResult> out;
    size_t k = 0;
    for (const Chain &chain : pdb.chains)
        for (const Residue &res : chain.residues) {
            std::string name;
            res.name(&name);
            out.push_back(ResidueResult{res.serial_number, res.insertion_code, seg[k++], name,
                                        is_polar_residue(name), chain.id});
        }
    return out;
}

// ChainLevel keeps the serialize_chain_id key: chains whose ids serialise to the same number
// share the map entry of the LAST of them (parent_to_atoms.insert overwrites, options.rs:361).
template <>
std::vector<ChainResult> finish<ChainLevel>(const Structure &pdb, const float *, size_t,
                                            const float *seg, float)
{
    std::map<std::int64_t, size_t> key_to_chain;
    for (size_t ci = 0; ci < pdb.chains.size(); ci++)
        key_to_chain[serialize_chain_id(pdb.chains[ci].id)] = ci;
    std::vector<ChainResult> out;
    for (const Chain &chain : pdb.chains)
        out.push_back(ChainResult{chain.id, seg[key_to_chain[serialize_chain_id(chain.id)]]});
    return out;
}

template <>
ProteinResult finish<ProteinLevel>(const Structure &pdb, const float *, size_t, const float *seg,
                                   float global)
{
    float polar = 0.f, non_polar = 0.f;  // options.rs:376-402
    size_t k = 0;
    for (const Chain &chain : pdb.chains)
        for (const Residue &res : chain.residues) {
            std::string name;
            res.name(&name);
            if (is_polar_residue(name)) polar += seg[k]; else non_polar += seg[k];
            k++;
        }
    return ProteinResult{global, polar, non_polar};  // global_total = simd_sum(atom_sasa), :404
}

std::string engine_message(const OptionValues &o, int rc)
{
    std::string m = std::string("rustsasa_amd engine: ") + rsasa_status_string(rc);
    if (o.context) m += std::string(": ") + rsasa_context_last_error(o.context);
    return m;
}

// CPUs this process can really use: the hardware threads, capped by its cgroup's CPU quota (cgroup v2 cpu.max).  A
// container limited to 16 CPUs on a 256-thread host gets 16 threads' worth of time however many threads it starts;
// a pool sized by hardware_concurrency() there spends its time being throttled (the measurement boxes are such
// containers: that is why 64 parse threads were slower than 32 there).
static unsigned effective_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long long quota = std::atoll(q);
            if (quota > 0) n = std::min(n, (unsigned)std::max(1ll, (quota + period / 2) / period));
        }
        std::fclose(f);
    }
    return n;
}

// Static-chunked parallel loop over [0, n) on up to `threads` std::threads (host glue only).
template <typename F>
void parallel_for(size_t n, unsigned threads, F body)
{
    const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, threads), n);
    if (nt <= 1) {
        for (size_t i = 0; i < n; i++) body(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < n;) body(i);
    };
    std::vector<std::thread> pool;
    for (unsigned k = 1; k < nt; k++) pool.emplace_back(worker);
    worker();
    for (auto &th : pool) th.join();
}

// Runs the GPU once for a set of prepared structures (those without a build error) and turns
// the results into per-structure level outputs.  Build errors stay with their structure and do
// not disturb the others (reference src/main.rs:446-454).
// `out_paths` (process_files with OptionValues::output_dir): the file a structure's result goes to as JSON, written by
// the thread that builds the result (bytes written are added to *bytes_written).
template <typename Level>
void run_batch(const OptionValues &o, const std::vector<const Structure *> &pdbs,
               std::vector<Prepared> &prep, std::vector<Result<typename Level::Output>> &out,
               unsigned host_threads = 1, const std::string *out_paths = nullptr, std::atomic<uint64_t> *bytes_written = nullptr)
{
    const size_t n_files = pdbs.size();
    out.assign(n_files, Result<typename Level::Output>());
    std::vector<uint32_t> s_off{0u}, seg_off{0u};
    std::vector<size_t> members;
    size_t n_total = 0;
    for (size_t f = 0; f < n_files; f++) {
        if (prep[f].err.error != SASACalcError::Ok) {
            out[f].error = prep[f].err.error;
            out[f].message = prep[f].err.message;
            continue;
        }
        members.push_back(f);
        for (uint32_t e : prep[f].seg_end) seg_off.push_back((uint32_t)n_total + e);
        n_total += prep[f].atoms.size();
        s_off.push_back((uint32_t)n_total);
    }
    if (members.empty()) return;
    const bool trace = rsasa::tuning_env("RSASA_FILES_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    std::vector<float> x(n_total), y(n_total), z(n_total), rad(n_total), atom(n_total, 0.f);
    std::vector<std::uint64_t> id(n_total);
    parallel_for(members.size(), host_threads, [&](size_t m) {
        size_t i = s_off[m];
        for (const rsasa_atom_t &a : prep[members[m]].atoms) {
            x[i] = a.position[0]; y[i] = a.position[1]; z[i] = a.position[2];
            rad[i] = a.radius; id[i] = a.id;
            i++;
        }
    });
    const size_t n_seg = seg_off.size() - 1;
    std::vector<float> seg(n_seg, 0.f), global(members.size(), 0.f);
    int rc = RSASA_OK;
    const auto tr1 = std::chrono::steady_clock::now();
    if (n_total) {  // calculate_sasa_internal on an empty slice returns an empty Vec
        rc = rsasa_calculate_sasa_batch(o.context, x.data(), y.data(), z.data(), rad.data(), id.data(),
                                        s_off.data(), members.size(), o.probe_radius, o.n_points,
                                        atom.data(), n_seg ? seg_off.data() : nullptr, n_seg,
                                        n_seg ? seg.data() : nullptr);
        if (rc == RSASA_OK && std::is_same<Level, ProteinLevel>::value)
            rc = rsasa_segment_sums(o.context, atom.data(), n_total, s_off.data(), members.size(),
                                    global.data());
    }
    const auto tr2 = std::chrono::steady_clock::now();
    std::vector<size_t> seg_pos(members.size() + 1, 0);
    for (size_t m = 0; m < members.size(); m++) seg_pos[m + 1] = seg_pos[m] + prep[members[m]].seg_end.size();
    const std::string engine_err = rc != RSASA_OK ? engine_message(o, rc) : std::string();
    parallel_for(members.size(), host_threads, [&](size_t m) {
        const size_t f = members[m];
        if (rc != RSASA_OK) {
            out[f].error = SASACalcError::Engine;
            out[f].message = engine_err;
            return;
        }
        out[f].value = finish<Level>(*pdbs[f], atom.data() + s_off[m], prep[f].atoms.size(),
                                     seg.data() + seg_pos[m], global[m]);
        if (out_paths) {  // reference src/main.rs:208-211: sasa_result_to_json + fs::write, per file
            const std::string text = sasa_result_to_json(out[f].value);
            std::string err;
            if (!write_whole_file(out_paths[f], text, &err)) {
                out[f].error = SASACalcError::Engine;
                out[f].message = err;
            } else if (bytes_written) {
                bytes_written->fetch_add(text.size(), std::memory_order_relaxed);
            }
        }
    });
    if (trace) {
        const auto tr3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "run_batch: %zu files %zu atoms: pack %.2f ms, engine %.2f ms, results %.2f ms\n", n_files, n_total,
                     ms(tr0, tr1), ms(tr1, tr2), ms(tr2, tr3));
    }
}

}  // namespace

Result<std::vector<float>> run_hot_path(const OptionValues &o, const std::vector<rsasa_atom_t> &atoms)
{
    Result<std::vector<float>> out;
    out.value.assign(atoms.size(), 0.f);
    if (atoms.empty()) return out;
    const int rc = rsasa_calculate_sasa_internal(o.context, atoms.data(), atoms.size(),
                                                 o.probe_radius, o.n_points, o.threads,
                                                 out.value.data());
    if (rc != RSASA_OK) {
        out.error = SASACalcError::Engine;
        out.message = engine_message(o, rc);
    }
    return out;
}

template <typename Level>
std::vector<Result<typename Level::Output>> process_many(const std::vector<const Structure *> &pdbs,
                                                         const OptionValues &o)
{
    std::vector<Prepared> prep(pdbs.size());
    for (size_t f = 0; f < pdbs.size(); f++) prep[f] = prepare<Level>(*pdbs[f], o);
    std::vector<Result<typename Level::Output>> out;
    run_batch<Level>(o, pdbs, prep, out);
    return out;
}

// process_files' cached second context per device (see there)
struct Companion {
    rsasa_context_t *ctx = nullptr;
    bool busy = false;
};
static std::mutex &companions_mutex()
{
    static std::mutex *m = new std::mutex();  // (never destroyed: the atexit handler below may run after other statics)
    return *m;
}
static std::map<int, Companion> &companions_map()
{
    static std::map<int, Companion> *m = new std::map<int, Companion>();
    return *m;
}
void release_cached_contexts_impl()
{
    std::lock_guard<std::mutex> lk(companions_mutex());
    auto &m = companions_map();
    for (auto it = m.begin(); it != m.end();) {
        if (it->second.busy) { ++it; continue; }  // (a process_files call is running with it)
        if (it->second.ctx) rsasa_context_destroy(it->second.ctx);
        it = m.erase(it);
    }
}
static void release_cached_contexts() { release_cached_contexts_impl(); }

// Directory mode at library level (reference src/main.rs:342-480 without the CLI): files are
// parsed and filtered on `host_threads` threads, then each chunk of `files_per_batch` structures
// is ONE GPU batch.
template <typename Level>
std::vector<Result<typename Level::Output>> process_files(const std::vector<std::string> &paths,
                                                          const OptionValues &o, unsigned host_threads,
                                                          size_t files_per_batch, FilesTimings *timings)
{
    using Clock = std::chrono::steady_clock;
    // Directory mode allocates and frees a few thousand small blocks per file on every thread.  With
    // glibc's defaults the heaps are trimmed and regrown all the time (brk / mprotect / page faults
    // under the process-wide mapping lock) and parsing stops scaling at a dozen threads; keeping freed
    // memory and growing the heaps in large steps makes the whole mode twice as fast (measured: 5.2 k
    // -> 10-12 k files/s).  Process-wide, once; RSASA_KEEP_MALLOC_DEFAULTS=1 leaves malloc alone.
    static std::once_flag malloc_once;
    std::call_once(malloc_once, [] {
        if (rsasa::tuning_env("RSASA_KEEP_MALLOC_DEFAULTS")) return;
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 256 << 20);
        mallopt(M_MMAP_THRESHOLD, 32 << 20);
    });
    std::vector<Result<typename Level::Output>> all(paths.size());
    // OptionValues::output_dir: every file's result also goes to <output_dir>/<file stem>.json (reference
    // src/main.rs:395-403: file_stem + the format's extension), written beside the GPU workers' result building
    std::vector<std::string> out_paths;
    std::atomic<uint64_t> bytes_written{0};
    if (!o.output_dir.empty()) {
        out_paths.resize(paths.size());
        for (size_t i = 0; i < paths.size(); i++) {
            const size_t slash = paths[i].find_last_of('/');
            std::string stem = paths[i].substr(slash == std::string::npos ? 0 : slash + 1);
            const size_t dot = stem.find_last_of('.');
            if (dot != std::string::npos && dot > 0) stem.resize(dot);
            out_paths[i] = o.output_dir + "/" + stem + ".json";
        }
    }
    // parsing is allocation heavy and stops scaling early (measured: 32 threads are the optimum on a
    // 256-thread host, 64 are slower), so the default is capped
    // (under a CPU quota twice the quota's threads: the parse threads also wait - for the page cache, for the chunk
    // barrier; measured with cpu.max = 16 CPUs: 16 threads 26 k files/s, 24: 33 k, 32: 37 k)
    if (host_threads == 0) host_threads = std::min(32u, 2u * effective_cpus());
    // (chunks of 512 files and two GPU workers: 16.4 k files/s on the 4 363-file set against 13.3 k with 256 and one
    // shared context - fewer per-chunk joins of the parse pool, and one chunk's upload beside the other's kernels)
    if (files_per_batch == 0) files_per_batch = 512;
    const bool fast_reader = rsasa::tuning_env("RSASA_NO_FAST_READER") == nullptr;
    // one worker per given context; with a single context a second, private one on the same device joins it
    // for the duration of the call (calls on one context are serialised)
    std::vector<rsasa_context_t *> contexts = o.contexts;
    if (contexts.empty()) contexts.push_back(o.context);
    // The companion context is kept for the life of the process, one per device (a context is a GPU workspace and a few
    // threads: creating one per call cost more than the call's second chunk); a call that finds it taken creates its
    // own.  It runs with the caller's context's settings: the pulp lane count decides which points take the remainder
    // rule, and chunks go to whichever worker is free.
    // The cached companions are released when the process ends normally, BEFORE the HIP runtime's own static teardown
    // (atexit handlers registered later run earlier; this one is registered after the first context exists, i.e. after
    // the runtime started up); rustsasa::release_cached_contexts() does the same on request - a long-lived host program
    // that is done with directory mode gets its HBM workspaces, pinned staging and coding threads back.
    static std::mutex &companions_mu = companions_mutex();
    static std::map<int, Companion> &companions = companions_map();
    static const bool registered = (std::atexit([] { release_cached_contexts(); }), true);
    (void)registered;
    struct Borrowed {
        rsasa_context_t *ctx = nullptr;
        int device = -1;
        bool cached = false;
        ~Borrowed()
        {
            if (cached) {
                std::lock_guard<std::mutex> lk(companions_mu);
                companions[device].busy = false;
            } else if (ctx) {
                rsasa_context_destroy(ctx);
            }
        }
    } second;
    if (contexts.size() == 1 && paths.size() > files_per_batch) {
        int device = 0;
        if (rsasa_context_get_device(contexts[0], &device) != RSASA_OK) device = 0;
        second.device = device;
        {
            std::lock_guard<std::mutex> lk(companions_mu);
            Companion &c = companions[device];
            if (!c.busy) {
                if (!c.ctx && rsasa_context_create(device, &c.ctx) != RSASA_OK) c.ctx = nullptr;
                if (c.ctx) {
                    c.busy = true;
                    second.ctx = c.ctx;
                    second.cached = true;
                }
            }
        }
        if (!second.ctx && rsasa_context_create(device, &second.ctx) != RSASA_OK) second.ctx = nullptr;
        // (every setting that changes how the caller's context computes: lane count, kernel tuning)
        if (second.ctx && rsasa_context_clone_settings(second.ctx, contexts[0]) == RSASA_OK)
            contexts.push_back(second.ctx);
        else
            contexts.push_back(contexts[0]);  // (no second context: two workers share the one)
    }

    struct Chunk {
        size_t base = 0, n = 0;
        std::vector<Structure> pdbs;
        std::vector<Prepared> prep;
    };
    // bounded queue: the producer (this thread + its parse pool) stays at most one chunk per
    // worker ahead, so memory holds a few chunks of parsed structures, not the whole directory
    std::mutex mu;
    std::condition_variable cv_push, cv_pop;
    std::deque<std::unique_ptr<Chunk>> queue;
    bool closed = false;
    const size_t capacity = contexts.size() + 1;
    FilesTimings t{};
    std::mutex mu_t;

    // Destroying a chunk's structures competes with the parse pool: free() takes the lock of the arena a block came
    // from - the arenas the parse threads are allocating from.  Measured on the 4 363-file set (32 parse threads, one
    // block per vector of the model): freed by 16 threads on the GPU workers' path the parse pool needed 0.17-0.25 s of
    // wall time per call, by 4 threads 0.10 s, by one 0.07 s (but then the freeing itself took 0.25 s).  Hence the
    // per-structure pools (include/rustsasa_amd.hpp) and ONE reaper with a few threads, off everybody's path.
    std::mutex mu_dead;
    std::condition_variable cv_dead, cv_dead_room;
    std::deque<std::unique_ptr<Chunk>> dead;
    bool no_more_dead = false;
    const unsigned free_threads = 4;
    std::thread reaper([&] {
        for (;;) {
            std::unique_ptr<Chunk> c;
            {
                std::unique_lock<std::mutex> lk(mu_dead);
                cv_dead.wait(lk, [&] { return no_more_dead || !dead.empty(); });
                if (dead.empty()) return;
                c = std::move(dead.front());
                dead.pop_front();
            }
            cv_dead_room.notify_one();
            parallel_for(c->n, free_threads, [&](size_t i) {
                Structure s = std::move(c->pdbs[i]);
                Prepared pr = std::move(c->prep[i]);
            });
        }
    });
    auto worker = [&](rsasa_context_t *ctx) {
        OptionValues mine = o;
        mine.context = ctx;
        (void)rsasa_context_bind_thread(ctx, nullptr);  // multi-socket hosts: this worker next to its GPU's link
        {
            int w = 0;
            (void)rsasa_context_get_simd_width(ctx, &w);
            std::lock_guard<std::mutex> lk(mu_t);
            t.worker_simd_widths.push_back(w);
        }
        // the first touch of a context initialises the HIP runtime (a quarter of a second): do it
        // here, while the producer parses the first chunk
        (void)rsasa_segment_sums(ctx, nullptr, 0, nullptr, 0, nullptr);
        for (;;) {
            std::unique_ptr<Chunk> c;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_pop.wait(lk, [&] { return closed || !queue.empty(); });
                if (queue.empty()) return;
                c = std::move(queue.front());
                queue.pop_front();
            }
            cv_push.notify_one();
            const auto t1 = Clock::now();
            std::vector<const Structure *> ptrs(c->n);
            for (size_t i = 0; i < c->n; i++) ptrs[i] = &c->pdbs[i];
            std::vector<Result<typename Level::Output>> out;
            run_batch<Level>(mine, ptrs, c->prep, out, std::max(1u, host_threads / 2), out_paths.empty() ? nullptr : out_paths.data() + c->base,
                             &bytes_written);
            for (size_t i = 0; i < c->n; i++) all[c->base + i] = std::move(out[i]);
            // the parsed structures (millions of small blocks) go to the reaper thread
            {
                std::unique_lock<std::mutex> lk(mu_dead);
                cv_dead_room.wait(lk, [&] { return dead.size() < 4; });  // (memory stays bounded if freeing falls behind)
                dead.push_back(std::move(c));
            }
            cv_dead.notify_one();
            const double dt = std::chrono::duration<double>(Clock::now() - t1).count();
            std::lock_guard<std::mutex> lk(mu_t);
            t.compute_seconds += dt;
        }
    };
    const auto t_begin = Clock::now();
    std::vector<std::thread> workers;
    for (rsasa_context_t *ctx : contexts) workers.emplace_back(worker, ctx);

    for (size_t base = 0; base < paths.size(); base += files_per_batch) {
        auto c = std::make_unique<Chunk>();
        c->base = base;
        c->n = std::min(files_per_batch, paths.size() - base);
        c->pdbs.resize(c->n);
        c->prep.resize(c->n);
        const auto t0 = Clock::now();
        // largest files first: the chunk's parse ends with the pool, not with one thread on a 2 MB file
        std::vector<std::pair<uint64_t, size_t>> order(c->n);
        for (size_t i = 0; i < c->n; i++) {
            struct stat st{};
            order[i] = {::stat(paths[base + i].c_str(), &st) == 0 ? (uint64_t)st.st_size : 0u, i};
        }
        std::sort(order.begin(), order.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
        parallel_for(c->n, host_threads, [&](size_t k) {
            const size_t i = order[k].second;
            try {
                const std::string &text = read_whole_file(paths[base + i]);
                const bool cif = is_mmcif_path(paths[base + i]);
                if (fast_reader) {  // plain files: kept atoms straight from the text (fast_pdb_prepare / fast_cif_prepare)
                    Structure light(text.size() / 8 + 4096);
                    if (cif ? fast_cif_prepare(text, o, seg_kind_of<Level>::value, light, c->prep[i])
                            : fast_pdb_prepare(text, o, seg_kind_of<Level>::value, light, c->prep[i])) {
                        c->pdbs[i] = std::move(light);
                        return;
                    }
                    c->prep[i] = Prepared{};
                }
                t_skip_occupancy_and_bfactor = !o.read_radii_from_occupancy;
                c->pdbs[i] = cif ? Structure::from_mmcif_text(text) : Structure::from_pdb_text(text);
                t_skip_occupancy_and_bfactor = false;
                c->prep[i] = prepare<Level>(c->pdbs[i], o);
            } catch (const std::exception &e) {  // unreadable file: report, keep going (main.rs:446-454)
                t_skip_occupancy_and_bfactor = false;
                c->prep[i] = Prepared{};
                c->prep[i].err = {SASACalcError::Engine, std::string("cannot read structure: ") + e.what()};
            }
        });
        size_t atoms = 0;
        for (size_t i = 0; i < c->n; i++) atoms += c->prep[i].atoms.size();
        {
            std::lock_guard<std::mutex> lk(mu_t);
            t.parse_seconds += std::chrono::duration<double>(Clock::now() - t0).count();
            t.n_atoms += atoms;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv_push.wait(lk, [&] { return queue.size() < capacity; });
        queue.push_back(std::move(c));
        lk.unlock();
        cv_pop.notify_one();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        closed = true;
    }
    cv_pop.notify_all();
    for (auto &th : workers) th.join();
    {
        std::lock_guard<std::mutex> lk(mu_dead);
        no_more_dead = true;
    }
    cv_dead.notify_all();
    reaper.join();
    t.total_seconds = std::chrono::duration<double>(Clock::now() - t_begin).count();
    t.n_files = paths.size();
    t.bytes_written = bytes_written.load();
    if (timings) *timings = t;
    return all;
}

#define RSASA_INSTANTIATE(L)                                                                          \
    template std::vector<Result<L::Output>> process_many<L>(const std::vector<const Structure *> &,  \
                                                            const OptionValues &);                   \
    template std::vector<Result<L::Output>> process_files<L>(const std::vector<std::string> &,       \
                                                             const OptionValues &, unsigned, size_t, \
                                                             FilesTimings *);
RSASA_INSTANTIATE(AtomLevel)
RSASA_INSTANTIATE(ResidueLevel)
RSASA_INSTANTIATE(ChainLevel)
RSASA_INSTANTIATE(ProteinLevel)
#undef RSASA_INSTANTIATE

}  // namespace detail

void release_cached_contexts() { detail::release_cached_contexts_impl(); }

Result<SelectedAtoms> select_atoms_by_chain(const Structure &pdb, const OptionValues &o)
{
    return detail::select_by_chain(pdb, o);
}

template <typename Level>
Result<typename Level::Output> SASAOptions<Level>::process(const Structure &pdb) const
{
    return std::move(detail::process_many<Level>({&pdb}, o_)[0]);
}

template <typename Level>
std::vector<Result<typename Level::Output>> SASAOptions<Level>::process_files(
    const std::vector<std::string> &paths, unsigned host_threads, size_t files_per_batch,
    FilesTimings *timings) const
{
    return detail::process_files<Level>(paths, o_, host_threads, files_per_batch, timings);
}

template class SASAOptions<AtomLevel>;
template class SASAOptions<ResidueLevel>;
template class SASAOptions<ChainLevel>;
template class SASAOptions<ProteinLevel>;

}  // namespace rustsasa
