/*
 * rustsasa_amd.h -- C ABI of the MI355X-native Shrake-Rupley SASA engine.
 *
 * This is the drop-in boundary for RustSASA's hot path.  The reference has no
 * FFI layer of its own; its seam is the free function
 *
 *     pub fn calculate_sasa_internal(atoms: &[Atom], probe_radius: f32,
 *                                    n_points: usize, threads: isize) -> Vec<f32>
 *                                                     (reference src/lib.rs:249-254)
 *
 * called from SASAOptions::<T>::process (reference src/options.rs:615-616).
 * Every entry point below cites the reference interface it replaces.  The
 * Rust-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function
 * returns RSASA_OK (0) or a negative rsasa_status; nothing throws across the
 * boundary; all `out_*` buffers are caller-owned.  The library is built for
 * gfx950 only and has NO CPU fallback: without a usable HIP device every
 * compute entry point fails with RSASA_ERR_NO_DEVICE.
 */
#ifndef RUSTSASA_AMD_H
#define RUSTSASA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: rsasa_batch_wait returns the OLDEST of up to two batches in flight (version 1 had one batch in flight, so
 * "the" batch); rsasa_batch_wait_all, rsasa_context_get_simd_width, rsasa_context_bind_thread added.
 * 3: rsasa_host_batch_enqueue / _wait / _wait_all (a stream of host batches), rsasa_context_clone_settings,
 *    rsasa_context_ids_dropped added;
 * 4: rsasa_context_set_call_combining, rsasa_call_combining_stats, rsasa_context_ids_kept added; rsasa_host_batch_enqueue with eight batches
 *    queued returns RSASA_ERR_QUEUE_FULL at once (it used to wait for the oldest batch and then fail);
 * nothing else changed, nothing removed. */
#define RSASA_ABI_VERSION 4

typedef enum rsasa_status {
    RSASA_OK = 0,
    RSASA_ERR_INVALID_ARGUMENT = -1, /* NULL where data is required, probe+max_radius <= 0, n_points == 0, ... */
    RSASA_ERR_NO_DEVICE = -2,        /* no HIP device / device index out of range */
    RSASA_ERR_HIP = -3,              /* a HIP runtime call failed; see rsasa_context_last_error */
    RSASA_ERR_OUT_OF_MEMORY = -4,
    RSASA_ERR_GRID_TOO_LARGE = -5,   /* a structure's cell grid exceeds 2^31 cells (coordinates too sparse) */
    RSASA_ERR_INTERNAL = -6,
    RSASA_ERR_QUEUE_FULL = -7        /* rsasa_host_batch_enqueue: eight batches are queued and not yet waited for */
} rsasa_status;

/* Mirrors `Atom` (reference src/structures/atomic.rs:13-24) without the
 * parent_id field, which the hot path never reads.  24 bytes, C layout. */
typedef struct rsasa_atom {
    float position[3];
    float radius;
    uint64_t id; /* atoms with equal id never occlude each other (src/lib.rs:124) */
} rsasa_atom_t;

typedef struct rsasa_context rsasa_context_t;

/* ---- library / device ------------------------------------------------- */
int rsasa_abi_version(void);
const char *rsasa_status_string(int status);
/* Number of HIP devices visible to this process (0 is a valid answer). */
int rsasa_device_count(int *out_count);

/* One context = one GPU + two launch streams with a growable HBM workspace
 * each (two batches in flight, see rsasa_batch_enqueue), copy streams of the
 * host-pointer entry points, and the cached sphere lattices.  Calls on one context are serialised by an internal
 * mutex; use one context per host thread (or per rayon worker) for
 * concurrency.  Replaces the reference's global rayon pool
 * (src/utils.rs:63-81) as the unit of parallel resources. */
int rsasa_context_create(int device, rsasa_context_t **out_ctx);
int rsasa_context_destroy(rsasa_context_t *ctx);
const char *rsasa_context_last_error(const rsasa_context_t *ctx);
/* The GPU index the context was created on (directory mode opens its second worker context there). */
int rsasa_context_get_device(const rsasa_context_t *ctx, int *out_device);

/* pulp lane count W the reference host would dispatch to (8 = AVX2+FMA
 * [default], 16 = AVX-512, 4 = NEON, 1 = scalar).  It only selects which of
 * the last (n_points mod W) sphere points use the reference's scalar
 * remainder rule -- unfused dot product and `<=` (src/lib.rs:163-218). */
int rsasa_context_set_simd_width(rsasa_context_t *ctx, int simd_width);
int rsasa_context_get_simd_width(rsasa_context_t *ctx, int *out_simd_width);

/* CALL COMBINING (ABI 4), for the reference's own call pattern left unchanged:
 * one rsasa_calculate_sasa_internal / rsasa_calculate_sasa_soa call per
 * structure from every worker thread (reference src/main.rs:375,439,
 * src/lib.rs:249-254).  With max_wait_us >= 0 the per-structure calls made
 * through this context - from any number of host threads, and together with
 * those of every other context of the same GPU that has it on - are merged into
 * batch launches: a caller that finds one of the GPU's two combining lanes free
 * leads a batch of every call queued at that moment with the same settings
 * (probe radius, point count, lane count W, ids passed or not; calls that
 * differ are never merged), every member copies its own atoms in and its own
 * values out.  While both lanes are busy the arriving calls queue up, so the
 * batches grow with the load and no timer is needed: max_wait_us = 0 is the
 * recommended setting; a positive value lets a leader hold its batch back up
 * to that long for as many calls as the previous batch had.  The values are
 * those of the call made alone, bit for bit (a batch is independent
 * structures).  A call the combiner does not take - more than 32 768 atoms,
 * non-finite input, anything the call alone would report as an error - runs
 * by itself and cannot fail its batch-mates; a device error fails every call
 * of the batch it hit with that status.  A context may be shared by all
 * threads or there may be one per thread.  max_wait_us < 0 switches it off
 * (the default). */
int rsasa_context_set_call_combining(rsasa_context_t *ctx, int max_wait_us);
/* Batches launched and calls merged into them on `device` since the process
 * started (either pointer may be NULL). */
int rsasa_call_combining_stats(int device, uint64_t *out_batches, uint64_t *out_calls);

/* Multi-socket hosts: binds the CALLING thread to the CPUs of the NUMA node the
 * context's GPU hangs off (sysfs numa_node / local_cpulist of its PCI address;
 * the context's own worker threads are bound the same way when they start).
 * A host program with one worker thread per GPU - the reference runs one rayon
 * worker per structure, src/main.rs:375 - calls this once at the top of each
 * worker, before it allocates the buffers it hands to the context.
 * *out_numa_node (nullable) receives the node, or -1 when the machine has one
 * node, hides its topology, or RSASA_NUMA=0 is set; then nothing is bound. */
int rsasa_context_bind_thread(rsasa_context_t *ctx, int *out_numa_node);

/* Copies every setting that changes how `src` computes - the pulp lane count
 * and the kernel tuning a process may have set - to `dst`: for programs that
 * run several contexts side by side (a second context on the same GPU, one
 * context per GPU) and want them to give the same values. */
int rsasa_context_clone_settings(rsasa_context_t *dst, rsasa_context_t *src);

/* ---- the hot path, one structure per call ------------------------------ */

/* Non-finite input (every entry point of the hot path, every kernel).
 * The reference checks nothing; what its arithmetic does with a NaN is
 * reproduced bit for bit: a NaN coordinate is skipped by the bounding box
 * (f32::min / max, spatial_grid.rs:113-121), lands in cell 0 (`as u32`,
 * :139-141) and fails every distance test (:321-335), so the atom is nobody's
 * neighbour, has no neighbours and keeps its whole sphere; a NaN radius makes
 * that atom's own value NaN (src/lib.rs:101-102,220-222) and is skipped by the
 * maximum radius (lib.rs:262).  Other atoms and the other structures of a
 * batch are not affected.  An INFINITE coordinate overflows the reference's
 * grid arithmetic (it panics): here the call - for a batch: the whole batch,
 * whose grids are placed by one scan - returns RSASA_ERR_GRID_TOO_LARGE and
 * the context stays usable.  probe_radius + largest radius not a positive
 * finite number - an infinite radius among them: an infinite cell size,
 * lib.rs:76 - returns RSASA_ERR_INVALID_ARGUMENT. */

/* Drop-in for calculate_sasa_internal (reference src/lib.rs:249-254).
 * `threads` is accepted for signature compatibility and ignored (the
 * reference uses it only to choose sequential vs rayon, src/lib.rs:278).
 * out_sasa[i] is the SASA of atoms[i] in A^2; n_atoms == 0 is valid. */
int rsasa_calculate_sasa_internal(rsasa_context_t *ctx, const rsasa_atom_t *atoms,
                                  size_t n_atoms, float probe_radius, size_t n_points,
                                  ptrdiff_t threads, float *out_sasa);

/* Same computation on struct-of-arrays input.  `id` may be NULL (all atoms
 * distinct). */
int rsasa_calculate_sasa_soa(rsasa_context_t *ctx, const float *x, const float *y,
                             const float *z, const float *radius, const uint64_t *id,
                             size_t n_atoms, float probe_radius, size_t n_points,
                             float *out_sasa);

/* ---- the hot path, many structures per call ---------------------------- */

/* Directory mode (reference src/main.rs:375,439): n_structures independent
 * structures concatenated into one SoA; structure s owns atoms
 * [structure_offsets[s], structure_offsets[s+1]).  Each structure gets its
 * own bounding box, cell grid and max radius exactly as a separate
 * calculate_sasa_internal call would.
 *
 * Optional ResidueLevel aggregation (reference src/options.rs:202-216,
 * src/utils.rs:14-22): residue k owns atoms
 * [residue_offsets[k], residue_offsets[k+1]) of the concatenated arrays and
 * out_residue_sasa[k] is their strictly sequential f32 sum.  Pass
 * residue_offsets = NULL / n_residues = 0 to skip.  out_atom_sasa may be NULL
 * when only residue values are wanted.  All pointers are HOST pointers. */
int rsasa_calculate_sasa_batch(rsasa_context_t *ctx, const float *x, const float *y,
                               const float *z, const float *radius, const uint64_t *id,
                               const uint32_t *structure_offsets, size_t n_structures,
                               float probe_radius, size_t n_points, float *out_atom_sasa,
                               const uint32_t *residue_offsets, size_t n_residues,
                               float *out_residue_sasa);

/* A STREAM of host batches (ABI 3): the same arguments and the same results as
 * rsasa_calculate_sasa_batch, but the call returns once the batch is queued.  A
 * rank that works through its share of a directory (reference
 * src/main.rs:375: files dealt to workers) enqueues batch k + 1 before it
 * waits for batch k: batch k + 1's first atoms cross the link while batch k's
 * last sub-batches compute and download, which one synchronous call after the
 * other cannot do (its first upload hides behind nothing, and nothing hides
 * its last kernels).  Two batches compute at a time - on two private contexts
 * on the caller's GPU, created by the first call (their workspaces and pinned
 * staging are sized like the caller's own would be, and live until
 * rsasa_context_destroy) - with the caller's settings at the time of the
 * enqueue; up to eight may be queued and not yet waited for: a ninth enqueue
 * does not block and does not take the batch - it returns RSASA_ERR_QUEUE_FULL
 * at once (nothing of the call's buffers is touched; call rsasa_host_batch_wait
 * and enqueue again).
 * rsasa_host_batch_wait() returns the OLDEST enqueued batch: it blocks until
 * that batch is complete and returns its status (the message is then the
 * context's last error); with nothing enqueued it returns RSASA_OK at once.
 * Several threads may enqueue and wait on one context at once: a waiting
 * thread takes the oldest batch no other thread is waiting for already.
 * Every buffer of a batch - inputs and outputs - belongs to the library from
 * the enqueue until the wait that returns the batch.  Pinned (page-locked)
 * host memory makes all copies asynchronous, as for the synchronous call.
 * The two worker contexts create their streams on hardware queues of their
 * own (the runtime otherwise multiplexes a process's streams onto four
 * queues, and two streams that share one run in order: nothing overlaps). */
int rsasa_host_batch_enqueue(rsasa_context_t *ctx, const float *x, const float *y,
                             const float *z, const float *radius, const uint64_t *id,
                             const uint32_t *structure_offsets, size_t n_structures,
                             float probe_radius, size_t n_points, float *out_atom_sasa,
                             const uint32_t *residue_offsets, size_t n_residues,
                             float *out_residue_sasa);
int rsasa_host_batch_wait(rsasa_context_t *ctx);
/* Waits for every enqueued host batch, oldest first; returns the first error. */
int rsasa_host_batch_wait_all(rsasa_context_t *ctx);

/* Device-resident form of the batch call: every pointer in the descriptor is
 * a DEVICE pointer on the context's GPU except structure_offsets_host, which
 * stays on the host (the launch geometry is derived from it).  The call only
 * enqueues work on `hip_stream` (a hipStream_t; NULL = one of the context's
 * own two streams) and returns.  Up to TWO batches may be in flight per
 * context, each in its own workspace (a third rsasa_batch_enqueue first
 * waits for the oldest): enqueueing batch k + 1 before waiting for batch k
 * keeps the GPU busy across the batch boundary - what a rank of a sharded
 * run does with its stream of batches (reference src/main.rs:375).
 * rsasa_batch_wait() waits for the OLDEST batch in flight, reports its
 * deferred errors and transparently re-runs it if the cell workspace had to
 * grow; with no batch in flight it returns RSASA_OK at once.  A batch's
 * buffers must stay valid until the rsasa_batch_wait that returns it. */
typedef struct rsasa_device_batch {
    const float *x, *y, *z, *radius;       /* [n_atoms] device */
    const uint64_t *id;                    /* [n_atoms] device, or NULL */
    const uint32_t *structure_offsets_host;/* [n_structures + 1] HOST */
    size_t n_structures;
    size_t n_atoms;
    const uint32_t *residue_offsets;       /* [n_residues + 1] device, or NULL */
    size_t n_residues;
    float *out_atom_sasa;                  /* [n_atoms] device, or NULL */
    float *out_residue_sasa;               /* [n_residues] device, or NULL */
    uint32_t *out_neighbor_counts;         /* [n_atoms] device, or NULL: per-atom
                                              candidate count K (the length of the
                                              reference's neighbour list,
                                              spatial_grid.rs:335-341) */
} rsasa_device_batch_t;

int rsasa_batch_enqueue(rsasa_context_t *ctx, const rsasa_device_batch_t *batch,
                        float probe_radius, size_t n_points, void *hip_stream);
int rsasa_batch_wait(rsasa_context_t *ctx);
/* Waits for EVERY batch in flight (oldest first) and returns the first error. */
int rsasa_batch_wait_all(rsasa_context_t *ctx);

/* MD-trajectory mode (the reference ecosystem's second workload: per-frame SASA of one
 * topology, README.md:98-149 / paper.md:45): n_frames frames of the same n_atoms atoms.
 * `xyz` is frame-major [n_frames][n_atoms][3] (HOST); radius / id / residue_offsets
 * ([n_residues + 1], offsets within one frame) are given once.  Every frame is an
 * independent structure (own bounding box, grid, max radius), exactly as n_frames separate
 * calculate_sasa_internal calls (src/lib.rs:249-254).  out_atom_sasa is [n_frames][n_atoms],
 * out_residue_sasa [n_frames][n_residues]; either may be NULL.  Only 12 bytes per atom and
 * frame cross PCIe. */
int rsasa_calculate_sasa_trajectory(rsasa_context_t *ctx, const float *xyz, size_t n_frames,
                                    size_t n_atoms, const float *radius, const uint64_t *id,
                                    float probe_radius, size_t n_points, float *out_atom_sasa,
                                    const uint32_t *residue_offsets, size_t n_residues,
                                    float *out_residue_sasa);

/* Strictly sequential f32 sums of contiguous segments of a host array, computed
 * on the GPU: out[k] = ((values[o[k]] + values[o[k]+1]) + ...) over
 * [offsets[k], offsets[k+1]).  This is the reference's simd_sum
 * (src/utils.rs:14-22) as used by the level aggregations
 * (src/options.rs:216,308,392,404). */
int rsasa_segment_sums(rsasa_context_t *ctx, const float *values, size_t n_values,
                       const uint32_t *offsets, size_t n_segments, float *out);

/* ---- measurement ------------------------------------------------------- */

/* When enabled, every rsasa_batch_enqueue brackets its kernels with HIP
 * events on the launch stream; the elapsed times of the most recent
 * completed batch are returned by rsasa_context_get_timings. */
typedef struct rsasa_timings {
    float grid_build_ms;   /* bounds + binning + scan + scatter kernels */
    float occlusion_ms;    /* the occlusion kernels alone (fast kernel + general kernel over deferred atoms) */
    float aggregate_ms;    /* residue sums */
    float total_ms;        /* first kernel start -> last kernel end */
    uint64_t n_cells;      /* total grid cells of the batch */
    uint64_t n_atoms;
    uint64_t n_deferred;   /* atoms the fast occlusion kernel left to the general one */
} rsasa_timings_t;

int rsasa_context_enable_timing(rsasa_context_t *ctx, int enable);
int rsasa_context_get_timings(rsasa_context_t *ctx, rsasa_timings_t *out);

/* Ids only matter where two atoms of one structure share one (a neighbour with
 * the atom's own id is skipped: reference src/lib.rs:127).  Large batches are
 * checked, and when the ids of every structure are all different the batch,
 * or the sub-batch of a pipelined host call, runs as one WITHOUT ids: the
 * same values, no id traffic, the id-less kernels (4 % faster).  Ids that
 * increase strictly within every structure (atom serials, indices) are found
 * by one comparison per atom, on the device or by the host's coding threads;
 * 64-bit ids in no order (hashes) that are on the device go through a hash
 * table per structure (up to 55 296 atoms per structure); ids the host has
 * folded to 32 bits (a pipelined host call's pinned ids) through the same
 * tables on their folds.  The verdict is each STRUCTURE's: one with two equal
 * ids (a file whose serial numbers repeat), or one nobody could check, keeps
 * its ids and runs in the kernels' instantiation with ids, the others of the
 * same batch without (ABI 4; a single such structure used to put the whole
 * batch on the slower path).  rsasa_context_ids_dropped counts the
 * (sub-)batches of the context, its stream of host batches included, in which
 * NO structure kept its ids; rsasa_context_ids_kept gives the number of
 * structures that kept theirs in the context's last checked (sub-)batch. */
int rsasa_context_ids_dropped(rsasa_context_t *ctx, uint64_t *out_batches);
int rsasa_context_ids_kept(rsasa_context_t *ctx, uint64_t *out_structures);

/* ---- utilities --------------------------------------------------------- */

/* The golden-section-spiral lattice the engine uploads to the GPU
 * (reference src/lib.rs:43-66), computed on the host with libm. */
int rsasa_sphere_points(size_t n_points, float *out_x, float *out_y, float *out_z);

#ifdef __cplusplus
}
#endif
#endif /* RUSTSASA_AMD_H */
