// rustsasa_amd.hpp -- C++ host API above the C ABI, mirroring RustSASA's level API
// (reference src/options.rs): SASAOptions<Level>::process(), AtomLevel /
// ResidueLevel / ChainLevel / ProteinLevel, the result structs of
// src/structures/atomic.rs:26-70 and the error enum SASACalcError
// (src/options.rs:466-494).
//
// Everything numeric happens on the GPU through include/rustsasa_amd.h
// (rsasa_calculate_sasa_batch); this layer only selects atoms, looks up radii
// and maps per-atom values to residues / chains exactly as the reference's
// build_atoms_and_mapping / process_atoms do.  There is no CPU compute path.
//
// The reference parses files with the `pdbtbx` crate, whose source is not
// available here; `Structure` is this library's own minimal PDB / mmCIF
// atom-record model.  Where its behaviour cannot be pinned against pdbtbx
// (DESIGN.md "Host API") this library makes its own, documented choice:
//   * multi-model files (NMR ensembles): ONLY THE FIRST MODEL is read.  The
//     reference iterates pdb.chains() / pdb.residues(), which in pdbtbx appear
//     to span every model; if they do, the reference sums an ensemble's models
//     into one result where this library reports model 1.  Single-model files
//     (every fixture of the reference's tests, every AlphaFold model) are
//     unaffected (pinned by tests/test_host_api.py::test_first_model_only_is_pinned);
//   * conformers: one per (residue name, alt-loc) pair in file order; in a residue
//     with alternate locations the atoms WITHOUT one belong to every conformer
//     (appended after the conformer's own atoms) and the blank conformer goes, so
//     that a residue's first conformer (`conformers().next()`,
//     src/options.rs:162,255) is backbone + first alternate location.  With this
//     rule the reference's whole quality set (88 files, 71 with alternate
//     locations; tools/check_quality_set.py) gives an RMSE of 43.997 against
//     FreeSASA - the reference's own figure is 43.99 (tests/quality.rs:17); with
//     blank and lettered conformers kept apart it is 142.  The ORDER of atoms
//     inside such a conformer (it decides AtomLevel's order and the f32 order of
//     ResidueLevel's sums on those residues) remains this library's choice.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <memory>
#include <memory_resource>
#include <utility>
#include <vector>

#include "rustsasa_amd.h"

namespace rustsasa {

// ---- structure model (what the reference gets from pdbtbx::PDB) ------------
struct AtomRecord {
    bool hetero = false;      // HETATM record            (atom.hetero())
    std::size_t serial = 0;   // atom serial number       (atom.serial_number())
    std::string name;         // atom name, trimmed       (atom.name())
    std::string element;      // upper-case element symbol, "" if unknown (atom.element())
    double x = 0, y = 0, z = 0;
    double occupancy = 1.0;
    double b_factor = 0.0;
};

struct Conformer {
    std::string name;     // residue name of this conformer
    std::string alt_loc;  // "" when none
    std::pmr::vector<AtomRecord> atoms;
};

struct Residue {
    std::int64_t serial_number = 0;
    std::string insertion_code;  // "" when none
    std::pmr::vector<Conformer> conformers;
    // Some(name) iff every conformer has the same name (pdbtbx Residue::name()).
    bool name(std::string *out) const;
};

struct Chain {
    std::string id;
    std::pmr::vector<Residue> residues;
};

// The model's nodes (the vectors of chains, residues, conformers and atoms) use std::pmr allocators: a structure the
// readers build keeps them in ONE pool that is released in a few blocks (a directory of structures is millions of
// nodes, and giving them back to malloc one by one took longer than parsing them); a default-constructed Structure,
// and every copy, uses the default resource (new / delete).
struct Structure {
    Structure() = default;
    explicit Structure(std::size_t pool_bytes);  // nodes from a pool of about this size (it grows when it has to)
    Structure(const Structure &other);           // deep copy, on the default resource
    Structure(Structure &&other) noexcept;
    Structure &operator=(const Structure &other);
    Structure &operator=(Structure &&other) noexcept;

  private:
    std::unique_ptr<std::pmr::monotonic_buffer_resource> pool_;  // (declared first: released after `chains`)

  public:
    std::pmr::vector<Chain> chains;
    std::vector<std::string> warnings;
    // Reads a PDB (ATOM/HETATM fixed columns) or mmCIF (_atom_site loop) file,
    // chosen by extension (.cif / .mmcif => mmCIF).  Throws std::runtime_error.
    static Structure open(const std::string &path);
    static Structure from_pdb_text(const std::string &text);
    static Structure from_mmcif_text(const std::string &text);
    std::size_t atom_count() const;
    // PDB text of the model (ATOM / HETATM records, b-factors included), e.g. after
    // sasa_result_to_protein_object; save_pdb throws std::runtime_error on I/O failure.
    std::string to_pdb_text() const;
    void save_pdb(const std::string &path) const;
};

// ---- radii -------------------------------------------------------------------
using RadiiConfig = std::unordered_map<std::string, std::unordered_map<std::string, float>>;
// FreeSASA-format config (reference src/utils/consts.rs:31-81).
RadiiConfig parse_radii_config(const std::string &content);
RadiiConfig load_radii_from_file(const std::string &path);  // throws std::runtime_error
// Embedded ProtOr table (reference radii/protor.config via PROTOR_RADII).
bool get_protor_radius(const std::string &residue, const std::string &atom, float *out);
// van-der-Waals radius of an element symbol (fallback of build_atom!, options.rs:89-93).
bool vdw_radius(const std::string &element, float *out);

// ---- results (reference src/structures/atomic.rs:26-70) -----------------------
struct ResidueResult {
    std::int64_t serial_number;
    std::string insertion_code;
    float value;
    std::string name;
    bool is_polar;
    std::string chain_id;
};
struct ChainResult {
    std::string name;
    float value;
};
struct ProteinResult {
    float global_total, polar_total, non_polar_total;
};

// ---- errors (reference src/options.rs:466-494) ---------------------------------
enum class SASACalcError {
    Ok = 0,
    ElementMissing,
    VanDerWaalsMissing,
    RadiusMissing,
    AtomMapToLevelElementFailed,
    FailedToGetResidueName,
    RadiiFileLoad,
    Engine,  // not in the reference: the GPU engine reported an error (see message)
};

template <typename T>
struct Result {
    SASACalcError error = SASACalcError::Ok;
    std::string message;  // Display text of the reference's error
    T value{};
    bool ok() const { return error == SASACalcError::Ok; }
};

// ---- levels ---------------------------------------------------------------------
struct AtomLevel { using Output = std::vector<float>; };
struct ResidueLevel { using Output = std::vector<ResidueResult>; };
struct ChainLevel { using Output = std::vector<ChainResult>; };
struct ProteinLevel { using Output = ProteinResult; };

// Selected atoms in hot-path form plus the parent -> atom-index map
// (reference AtomsMappingResult, options.rs:78).
struct AtomsAndMapping {
    std::vector<rsasa_atom_t> atoms;
    std::map<std::int64_t, std::vector<std::size_t>> parent_to_atoms;
};

struct OptionValues {
    float probe_radius = 1.4f;        // options.rs:500
    std::size_t n_points = 100;       // options.rs:501
    std::ptrdiff_t threads = -1;      // options.rs:502 (no meaning on the GPU; kept for the signature)
    bool include_hydrogens = false;   // options.rs:503
    std::shared_ptr<const RadiiConfig> radii_config;  // options.rs:504
    bool allow_vdw_fallback = false;  // options.rs:505
    bool include_hetatms = false;     // options.rs:506
    bool read_radii_from_occupancy = false;  // options.rs:507
    rsasa_context_t *context = nullptr;      // GPU context; nullptr = the library's default (device 0)
    // process_files only: one GPU worker thread per context (one context per GPU), fed from a queue
    // of parsed chunks.  Empty = {context}.
    std::vector<rsasa_context_t *> contexts;
    // process_files only: when set, every file's result is ALSO written to <output_dir>/<file stem>.json in serde's shape
    // (sasa_result_to_json) - the reference's directory mode end to end (src/main.rs:203-226,395-403: one output file
    // per input) -, by the threads that build the results, beside the GPU workers.  The directory must exist.  A file
    // that cannot be written is that file's error.
    std::string output_dir;
};

// Wall-clock split of SASAOptions::process_files.
struct FilesTimings {
    double parse_seconds = 0;    // reading + parsing + atom selection on the host threads (producer, summed)
    double compute_seconds = 0;  // packing, H2D, GPU hot path, D2H, result mapping (GPU workers, summed; overlaps parsing)
    double total_seconds = 0;
    std::size_t n_files = 0, n_atoms = 0;
    std::uint64_t bytes_written = 0;  // JSON written to OptionValues::output_dir
    std::vector<int> worker_simd_widths;  // the pulp lane count every GPU worker's context ran with (the caller's, on each)
};

// What ChainLevel hands to the hot path for one structure: the kept atoms (build_atoms_and_mapping,
// options.rs:317-364: first conformer per residue, hydrogen / HETATM filters, radii by options) and, per chain
// in file order, its id and the end of its atoms.  No GPU involved: used to check the selection against the
// reference's quality set (tools/check_quality_set.py).
struct SelectedAtoms {
    std::vector<rsasa_atom_t> atoms;
    std::vector<std::string> chain_ids;
    std::vector<std::uint32_t> chain_end;
};
Result<SelectedAtoms> select_atoms_by_chain(const Structure &pdb, const OptionValues &o);

namespace detail {
Result<SelectedAtoms> select_by_chain(const Structure &pdb, const OptionValues &o);
// test hook: directory mode's per-file work (level 0 atom, 1 residue, 2 chain, 3 protein) as JSON, through its short
// cut for plain PDB files (`fast`) or the general reader; no GPU (tests/test_host_api.py)
std::string debug_prepare_json(const std::string &path, const OptionValues &o, int level, bool fast);
double debug_prepare_seconds(const std::string &path, const OptionValues &o, int level, bool fast, int reps, size_t *n_atoms, bool *used_fast);
Result<std::vector<float>> run_hot_path(const OptionValues &o, const std::vector<rsasa_atom_t> &atoms);
template <typename Level>
std::vector<Result<typename Level::Output>> process_many(const std::vector<const Structure *> &pdbs,
                                                         const OptionValues &o);
template <typename Level>
std::vector<Result<typename Level::Output>> process_files(const std::vector<std::string> &paths,
                                                          const OptionValues &o, unsigned host_threads,
                                                          std::size_t files_per_batch,
                                                          FilesTimings *timings);
}  // namespace detail

// SASAOptions::process_files keeps a second context per GPU for the life of the process (its HBM workspaces, pinned
// staging and coding threads: a fresh one per call cost more than a call's second chunk).  They are released when the
// process ends normally; a long-lived program that is done with directory mode calls this to get them back at once
// (contexts a running process_files call is using stay).
void release_cached_contexts();

template <typename Level>
class SASAOptions {
public:
    SASAOptions() = default;                                      // options.rs:498-510
    SASAOptions &with_probe_radius(float r) { o_.probe_radius = r; return *this; }
    SASAOptions &with_include_hetatms(bool v) { o_.include_hetatms = v; return *this; }
    SASAOptions &with_n_points(std::size_t n) { o_.n_points = n; return *this; }
    SASAOptions &with_read_radii_from_occupancy(bool v) { o_.read_radii_from_occupancy = v; return *this; }
    SASAOptions &with_threads(std::ptrdiff_t t) { o_.threads = t; return *this; }
    SASAOptions &with_include_hydrogens(bool v) { o_.include_hydrogens = v; return *this; }
    SASAOptions &with_radii_file(const std::string &path)         // throws std::runtime_error
    {
        o_.radii_config = std::make_shared<const RadiiConfig>(load_radii_from_file(path));
        return *this;
    }
    SASAOptions &with_allow_vdw_fallback(bool v) { o_.allow_vdw_fallback = v; return *this; }
    SASAOptions &with_context(rsasa_context_t *ctx) { o_.context = ctx; return *this; }
    // directory mode over several GPUs: one context per device (rsasa_context_create(device, ...))
    SASAOptions &with_contexts(std::vector<rsasa_context_t *> ctxs) { o_.contexts = std::move(ctxs); return *this; }
    // directory mode end to end: per-file JSON into `dir` (OptionValues::output_dir)
    SASAOptions &with_output_dir(std::string dir) { o_.output_dir = std::move(dir); return *this; }
    const OptionValues &values() const { return o_; }

    // options.rs:606-618
    Result<typename Level::Output> process(const Structure &pdb) const;

    // Directory mode at library level (reference src/main.rs:342-480, `files.par_iter()` :375):
    // chunks of `files_per_batch` files (0 = 256) are parsed + selected on `host_threads` threads
    // (0 = min(16, all)) and queued; one worker thread per GPU context (with_contexts) takes a
    // chunk, runs it as ONE GPU batch and builds the results while the next chunk is being
    // parsed.  Structures are independent, so several GPUs need no exchange.  A file that fails
    // (unreadable, missing radius, ...) gets its own error; the others are unaffected
    // (main.rs:446-454).
    std::vector<Result<typename Level::Output>> process_files(const std::vector<std::string> &paths,
                                                              unsigned host_threads = 0,
                                                              std::size_t files_per_batch = 0,
                                                              FilesTimings *timings = nullptr) const;

private:
    OptionValues o_;
};

extern template class SASAOptions<AtomLevel>;
extern template class SASAOptions<ResidueLevel>;
extern template class SASAOptions<ChainLevel>;
extern template class SASAOptions<ProteinLevel>;

// ---- output (reference src/utils/io.rs) -------------------------------------------
// JSON in the shape serde gives the externally tagged SASAResult enum (io.rs:11-13,
// atomic.rs:62-70): {"Atom":[..]}, {"Residue":[{..}]}, {"Chain":[{..}]}, {"Protein":{..}}.
std::string sasa_result_to_json(const std::vector<float> &atom_level);
std::string sasa_result_to_json(const std::vector<ResidueResult> &residue_level);
std::string sasa_result_to_json(const std::vector<ChainResult> &chain_level);
std::string sasa_result_to_json(const ProteinResult &protein_level);
// Writes the values into the b-factors of `pdb` (io.rs:20-64).  Returns false with a message
// where the reference would panic (result / structure size mismatch).
bool sasa_result_to_protein_object(Structure &pdb, const std::vector<float> &atom_level, std::string *err);
bool sasa_result_to_protein_object(Structure &pdb, const std::vector<ResidueResult> &residue_level, std::string *err);
bool sasa_result_to_protein_object(Structure &pdb, const std::vector<ChainResult> &chain_level, std::string *err);
bool sasa_result_to_protein_object(Structure &pdb, const ProteinResult &protein_level, std::string *err);

// The reader's decimal parser (exact: equals strtod on every input); exposed for tests.
double parse_decimal_text(const std::string &text);

// helpers shared with the reference's utils.rs
std::int64_t serialize_chain_id(const std::string &s);                  // utils.rs:24-33
std::uint64_t fnv_hash_altloc_serial(const std::string &alt, std::size_t serial);  // utils.rs:83-87
bool is_polar_residue(const std::string &name);                        // consts.rs:7-16

}  // namespace rustsasa
