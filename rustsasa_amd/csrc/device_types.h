// Shared host/device plain-data types of the SASA engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rsasa {

// Per-structure uniform cell grid, restating SpatialGrid::new
// (reference src/structures/spatial_grid.rs:28-50).  64 bytes.
struct StructGrid {
    float min_x, min_y, min_z;     // bounding-box minimum minus cell_size (spatial_grid.rs:124-127)
    float inv_cell;                // 1.0 / cell_size                      (spatial_grid.rs:36)
    uint32_t dim_x, dim_y, dim_z;  // ceil(extent * inv_cell) + 1          (spatial_grid.rs:39-43)
    uint32_t cell_base;            // first cell of this structure in the batch-wide cell array: index of a 16-bit
                                   // entry when in_lds (multiple of 8), of a 32-bit entry otherwise
    float max_r;                   // fold(0.0, max) of the radii           (lib.rs:259-262)
    float cell_size;               // probe + max_r                        (lib.rs:76)
    uint32_t n_cells;
    uint32_t atom_begin;           // first atom of the structure in input order
    uint32_t n_atoms;
    uint32_t sorted_base;          // first position of the structure in the cell-sorted arrays
    uint32_t in_lds;               // 1: binned by k_sort_window (fewer than 65536 atoms; 16-bit cell starts relative
                                   // to sorted_base), 0: by the batch-wide kernels (32-bit absolute cell starts)
    uint32_t odd_radii;            // bit 0: some radius of the structure lies outside [0, 64] or is NaN, or some coordinate is
                                   // NaN, infinite or beyond 1e8 (the matrix-core occlusion kernel leaves such structures
                                   // to the general kernel: its padding records must be finite vectors);
                                   // bit 1 (BatchView::ids_check): the structure keeps its ids - two of them are equal, or nobody
                                   // could check (k_bounds, k_ids_distinct); bit 2: its ids are in no order (k_bounds);
                                   // bits 8..9: log2 of the x-cell block its atom groups share (grid_group_shift)
};
static_assert(sizeof(StructGrid) == 64, "StructGrid layout");

// Order-preserving integer images of the running min/max, combined with
// integer atomics by the bounds kernel.
struct StructAcc {
    int min_x, min_y, min_z;
    int max_x, max_y, max_z;
    int max_r;
    uint32_t n_atoms;      // atoms of the structure (sum over its bounds workgroups)
    uint32_t first_atom;   // smallest atom index of the structure
    int odd_radii;         // bit 0: a radius outside [0, 64] or NaN, a coordinate non-finite or beyond 1e8; bits 1, 2: the ids' verdict
                           // (see StructGrid::odd_radii)
};

// A contiguous slice of one structure handled by one bounds workgroup.
struct Segment {
    uint32_t sid;
    uint32_t begin;
    uint32_t end;
    uint32_t continues;  // 1: atom begin - 1 belongs to the same structure (k_bounds' id check looks across the seam)
};

// Deferred status of a batch, written on device, copied to pinned host memory.
// k_occlusion_mx's waves claim blocks of atoms from one counter per XCD piece of the launch's range; every counter has
// a 128-byte line to itself (all of them in one line made every claim of the GPU queue up behind one address), and
// the launch over the tail structures has its own set.
constexpr uint32_t kClaimStride = 32;                       // uint32 entries between two counters
constexpr uint32_t kClaimBytes = 2u * 8u * kClaimStride * 4u;

struct BatchStatus {
    uint32_t overflow;        // total cells exceed the workspace capacity: re-run after growing
    uint32_t grid_too_large;  // some structure needs more than 2^31 cells
    uint32_t bad_input;       // probe + max_r <= 0 or non-finite bounds
    uint32_t deferred;        // atoms k_occlusion_fast left to the general kernel (BatchView::deferred_list)
    uint64_t total_cells;     // 32-bit entries of the batch-wide cell array in use (the 16-bit cell starts of the
                              // LDS-binned structures first, two per entry, then the tail's)
    uint64_t tail_cell_begin; // first entry of the structures binned by the batch-wide kernels (multiple of 1024)
    uint32_t tail_atom_base;  // their first position in the cell-sorted arrays (= atoms of the LDS-binned structures)
    uint32_t n_windows;       // entries of BatchView::windows (work list of k_sort_window)
    uint64_t grid_cells;      // cells of all grids (statistic)
    uint32_t ids_needed;      // BatchView::ids_check: the structures that keep their ids (StructGrid::odd_radii bit 1).  0 = the ids of
                              // every structure are all different, so "another atom with my id" never happens - the batch
                              // runs as one without ids; otherwise each structure runs in the instantiation that is its own
    uint32_t ids_unordered;   // bit 0, the same: some id does not rise above its predecessor's (k_bounds); k_ids_distinct then looks
                              // for equal ids structure by structure.  Bit 1: the one occlusion launch of the batch was the id-less
                              // instantiation and the ids turned out to matter (OcclusionChain::solo): nothing was computed
};
static_assert(sizeof(BatchStatus) == 56, "BatchStatus layout");

struct Lattice {
    const float *x, *y, *z;  // device SoA, padded with zeros to a multiple of 64 entries
    const float4 *xyz4;      // the same points as (x, y, z, 0) records
    const uint4 *patches;    // point counts above 128: one entry per aligned run of 16 points, f16 (cz, cy | cx, -1 | eps, 0 | 0, 0)
    const float *mx_tab;     // at most 128 points: the matrix-core kernel's operand tables (mx_tab_floats(): context.cpp get_lattice);
                             // single-wave workgroups read them from here (L1 / L2 hits), not from a per-workgroup LDS copy
    uint32_t n_patches;      // (see context.cpp bisect_points, occlusion_mx.inc); else null / 0
    uint32_t n_points;
    uint32_t n_fused;        // points [0, n_fused) use the fused-FMA `<` rule (lib.rs:143-146);
                             // the rest the scalar remainder rule (lib.rs:185-186,206-207)
};

// Lattice::mx_tab (point counts up to 128): NT = 6, 7 or 8 tiles of 16 points cover them (the kernel's template
// argument); NPS = 16 NT + 16.  Floats [0, 4 NPS): the points component major, (z | y | x | -1) x NPS, entries
// behind the last point zero (among them the 16 behind the tiles: a column that nothing occludes); then 16 NT
// entries of 8 bytes: the same points as f16 (z, y | x, -1), rounded to nearest, zero behind the last point.
inline uint32_t mx_tab_tiles(uint32_t n_points) { return n_points <= 96u ? 6u : n_points <= 112u ? 7u : 8u; }
inline uint32_t mx_tab_floats(uint32_t n_points) { return n_points > 128u ? 0u : 4u * (16u * mx_tab_tiles(n_points) + 16u) + 2u * 16u * mx_tab_tiles(n_points); }

// Running sums of the grid placement scan: (cells, atoms) of the LDS-binned / tail structures.
struct GridSums {
    unsigned long long cells_s, cells_l, atoms_s, atoms_l;
};

// Everything a batch run needs on the device.  All pointers are device pointers.
struct BatchView {
    // inputs
    const float *x, *y, *z, *radius;
    const uint64_t *id;  // may be null
    const uint32_t *id32;  // nullable: the ids already folded to 32 bits (fold_id) by the host, in input order.  When set,
                           // the binning kernels read these instead of `id`, no sorted copy of the 64-bit ids is kept, and
                           // `id` (device-accessible, possibly mapped host memory) is only read by the general occlusion
                           // kernel for the few atoms whose folds collide
    const uint8_t *radius8;  // nullable (pipelined host path): the radii as one-byte codes into `radius_table`, read instead of
    const float *radius_table;  // `radius` (a batch has few distinct radii: 1 byte per atom crosses the link, not 4)
    const uint32_t *residue_offsets;
    uint32_t n_atoms, n_structures, n_residues, n_segments;
    float probe;
    // Ids only matter where two atoms of a structure share one (lib.rs:127: a neighbour with the atom's own id is
    // skipped).  1: k_bounds also checks that the ids of every structure increase strictly (what atom serials and indices
    // do); where they do not (hashes), k_ids_distinct puts that structure's ids (or the host's 32-bit folds of them) through
    // a hash table in LDS.  The verdict is each STRUCTURE's (StructGrid::odd_radii bit 1): one whose ids are all different
    // is computed as one WITHOUT ids - no id loads, no sorted copies, the occlusion kernel's id-less instantiation (4 %
    // faster) - with identical results; BatchStatus::ids_needed counts the others.  Set for batches with ids that the
    // matrix-core kernel takes.
    uint32_t ids_check;
    const uint32_t *large_sids;    // k_ids_distinct: the structures of more than kIdAtomsSmall (and at most kIdAtomsLarge) atoms
    uint32_t n_large;
    uint32_t ids_tables;           // the k_ids_distinct launches are part of this batch (the context's last checked batch had
                                   // ids in no order); 0: they are not - their workgroups wait for LDS even to return -, and
                                   // ids that do not rise simply stay in play for this batch
    // workspace
    const Segment *segments;
    StructAcc *acc;
    StructGrid *grids;
    uint32_t *sid_sorted;         // structure of the atom at cell-sorted position p
    uint32_t *cell_of, *rank_of;  // binning only.  Batch-wide route: cell index / arrival rank inside the cell;
                                  // k_sort_window: rank_of = sorted position of the atoms a workgroup's registers do not hold
    uint32_t *deferred_list;      // atoms k_occlusion_fast left to the general kernel (BatchStatus::deferred entries)
    uint32_t *claim;              // k_occlusion_mx: block counters of its persistent waves, kClaimBytes (launch_occlusion zeroes them)
    uint32_t *ids_seg;            // BatchView::ids_check: two bitmaps of ids_seg_words words, a bit per 64 cell-sorted atoms - "an atom of a
    uint32_t ids_seg_words;       // structure that keeps its ids is among them" and "... of one that does not" (k_ids_segments): a workgroup of
                                  // k_occlusion_mx whose atoms hold nothing of its instantiation's returns before it has fetched anything else
    uint32_t *cells;              // cell starts (cell_capacity + 1 entries of 32 bits; see StructGrid::cell_base)
    uint64_t cell_capacity;
    uint4 *windows;               // (structure, window, first atom, atoms) of every k_sort_window workgroup
    uint32_t window_capacity;     // entries of `windows` = workgroups launched (the surplus exits)
    uint32_t *scan_block_sums;
    GridSums *grid_sums;   // per 256 structures: (cells, atoms) x (LDS-binned, tail), 4 x u64
    float4 *sorted_xyzr;          // cell-sorted (x, y, z, radius)
    uint32_t *sorted_orig;        // cell-sorted position -> input index
    uint64_t *sorted_id;          // cell-sorted ids (only when id != null; null too when the matrix-core kernel takes
                                  // the batch: it works on the folds, and the general kernel fetches the few ids it
                                  // needs through sorted_orig)
    uint32_t *sorted_id32;        // the same folded to 32 bits (fold_id): different folds => different ids
    BatchStatus *status;
    // outputs
    float *atom_sasa;             // never null (workspace buffer when the caller passed none)
    float *residue_sasa;          // may be null
    uint32_t *neighbor_counts;    // may be null
    // Nullable, pinned host memory (the one-structure call): k_occlusion_fast sets it to 1 when it leaves an
    // atom to the general kernel, and launch_occlusion then does NOT launch that kernel - the caller looks at
    // the flag after the stream has drained and calls launch_occlusion_deferred if it is set (it rarely is).
    uint32_t *defer_flag;
};

// Grid and status of a one-structure batch, computed by the host and handed to k_sort_window<true> as
// kernel arguments.
struct SingleJob {
    StructGrid grid;
    BatchStatus status;
};

// Occlusion kernel selection (RSASA_OCCLUSION_KERNEL / RSASA_ATOMS_PER_WAVE, read once per
// context; for A/B measurements -- every version computes identical results).
struct OcclusionTuning {
    int kernel_version = 6;       // 0 = all-pairs reference kernel, 3 = general culled-sweep kernel for every atom,
                                  // 4 = straight-line per-atom kernel, 5 = group-union sweep + matrix-core point
                                  // tests (4 and 5 leave the atoms they cannot take to the general kernel),
                                  // 6 = 5 for batches of 32 768 atoms or more, 4 below
    uint32_t atoms_per_wave = 0;  // 0 = choose from the batch size
    uint32_t debug_stop = 0;      // RSASA_DEBUG_STOP: skip later kernel stages (WRONG results; timing ablation only)
    uint32_t deferred_hint = 0xFFFFFFFFu;  // atoms the context's last completed batch left to the general kernel (unknown at
                                           // first): sizes the launch that works off the next batch's deferred list - a grid-stride
                                           // loop, so any size is correct; 1 024 workgroups that find an empty list cost 20 us
};

// Launchers implemented in kernels.hip / occlusion.hip.  Each only enqueues on `stream`.
void launch_grid_prepare(const BatchView &b, hipStream_t stream);
void launch_sort_lds(const BatchView &b, hipStream_t stream);
void launch_sort_single(const BatchView &b, const SingleJob &job, hipStream_t stream);
void launch_sort_tail(const BatchView &b, hipStream_t stream);
// Which atoms (cell-sorted positions) an occlusion launch covers: the tail's binning may still be
// running on another stream while the LDS-binned structures are processed.
enum OcclusionPart : uint32_t {
    kOccAll = 0,   // every atom (both binning routes are complete)
    kOccHead = 1,  // positions below BatchStatus::tail_atom_base (LDS-binned structures); may launch nothing
    kOccRest = 2,  // whatever kOccHead did not launch, plus the deferred atoms
};
// Two batches in flight: a batch's occlusion kernel starts when the other batch's has ended, and "has ended" is an event
// recorded right behind THAT kernel - not behind the launches that follow it (the instantiation the id check does not ask
// for, which returns at once, and the general kernel over the deferred list: 25 us that the next occlusion kernel then
// does not wait for).  launch_occlusion waits for `wait` in front of the launch that does the work, records `start` (timing,
// nullable) there and `done` behind it.  With BatchView::ids_check the host does not know which instantiation will work:
// `expect_ids_dropped` (what the context's last batch did) puts the other one FIRST, in front of the wait, where it runs
// beside the neighbour's kernel for nothing; a wrong guess only loses the ordering for that batch.
// `solo` (callers that can run a batch again: rsasa_batch_wait): when the ids are expected to be dropped only the id-less
// instantiation is launched; should the device find that the ids do matter, that launch does nothing but raise bit 1 of
// BatchStatus::ids_unordered, and the caller runs the batch again with its ids (once, at a change of the id pattern -
// against an empty launch of 45 000 workgroups in front of every batch's working kernel: 15 us of every step).
struct OcclusionChain {
    hipEvent_t wait = nullptr, start = nullptr, done = nullptr;
    bool expect_ids_dropped = true;
    bool solo = false;
};
void launch_occlusion(const BatchView &b, const Lattice &lat, const OcclusionTuning &tune,
                      OcclusionPart part, hipStream_t stream, const OcclusionChain *chain = nullptr);
// Whether launch_occlusion gives a batch of n_atoms atoms to the matrix-core kernel.
bool occlusion_uses_mx(const OcclusionTuning &tune, const Lattice &lat, uint32_t n_atoms);
// The general kernel over the atoms the straight-line kernel deferred (see BatchView::defer_flag).
void launch_occlusion_deferred(const BatchView &b, const Lattice &lat, hipStream_t stream);
void launch_residue_sums(const BatchView &b, hipStream_t stream);
// Pinned 24-byte atom records (x, y, z, r, id) -> device columns, and `hdr_bytes` of header beside them (combine.cpp).
void launch_unpack_atoms(const void *records, uint32_t n_atoms, float *x, float *y, float *z, float *r, uint64_t *id,
                         const void *hdr_src, void *hdr_dst, uint32_t hdr_bytes, hipStream_t stream);
void launch_expand_frames(const float *xyz, const float *radius, const uint64_t *id,
                          const uint32_t *res_off, uint32_t n_atoms, uint32_t n_frames, uint32_t res_stride,
                          float *x, float *y, float *z, float *r, uint64_t *id_out,
                          uint32_t *res_out, hipStream_t stream);

constexpr uint32_t kSegmentAtoms = 4096;  // atoms per bounds workgroup
constexpr uint32_t kScanBlocks = 1024;    // workgroups of the cell scan
#ifndef RSASA_WINDOW_CELLS
#define RSASA_WINDOW_CELLS 36864
#endif
#ifndef RSASA_SORT_THREADS
#define RSASA_SORT_THREADS 1024
#endif
constexpr uint32_t kWindowCells = RSASA_WINDOW_CELLS;  // cells one k_sort_window workgroup bins (16-bit counters, 72 KiB: two per CU)
// k_ids_distinct (BatchView::ids_check): a table of 8 192 slots in LDS for structures of up to 4 096 atoms (32 KB: several
// workgroups per CU), one of 36 864 for up to 27 648 (144 KB, one workgroup per such structure)
// (the large table's entries are 16 bits wide - an atom's number within its structure, below 65 536 -: twice the slots in the
// same 144 KB, which takes structures of up to 55 296 atoms; round 5's 32-bit entries stopped at 27 648, and the reference's own
// quality set holds a complex of 32 500)
constexpr uint32_t kIdSlotsSmall = 8192, kIdAtomsSmall = 4096, kIdSlotsLarge = 73728, kIdAtomsLarge = 55296;
constexpr uint32_t kLdsMaxAtoms = 65536;  // structures with fewer atoms are binned in LDS (16-bit positions)

// 16-bit entries a structure of n_cells cells takes in the cell array: its cells, the end marker,
// padding to whole 16-byte vectors.
// k_occlusion_mx groups atoms of one cell row whose x cells fall in the same aligned block of 2^shift cells; the group
// shares the union of the 25 x-runs around it, (2^shift + 4) cells long, which must hold at most 256 atoms or the group
// is narrowed (and the runs looked up again).  Where atoms fill their grid - 1.5 per cell in a dense medium - a block
// of 8 cells never fits: start from the width that does.  (A heuristic: it changes the kernel's speed, not its results.)
__host__ __device__ inline uint32_t grid_group_shift(uint32_t n_atoms, uint32_t n_cells)
{
    const float per_cell = (float)n_atoms / (float)(n_cells ? n_cells : 1u);
    return per_cell >= 1.0f ? 0u : per_cell >= 0.6f ? 1u : per_cell >= 0.3f ? 2u : 3u;
}
__host__ __device__ constexpr uint32_t lds_cell_slots(uint32_t n_cells) { return (n_cells + 1u + 7u) & ~7u; }
__host__ __device__ constexpr uint32_t grid_windows(uint32_t n_cells) { return (n_cells + kWindowCells - 1u) / kWindowCells; }

}  // namespace rsasa
