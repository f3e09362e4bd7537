/*
 * sasa_oracle.h -- CPU restatement of RustSASA's Shrake-Rupley hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rustsasa_amd/ (the product) may
 * include, link, load or call this.  Allowed users: tests/, the smoke check in
 * __graft_entry__.py and the `cpu_baseline` leg of bench.py.
 *
 * Parity status: PINNED against the reference's golden vector
 * FIXED_LOW_RES_ATOMS (tests/common/data.rs:4-238 of the reference; 2 622
 * per-atom values for tests/data/pdbs/example.cif, pdbtbx van-der-Waals radii,
 * probe 1.4, 100 points), the six analytic cases of tests/sanity.rs:20-157 and
 * the neighbour-membership test tests/units.rs:132-209.  The reference itself
 * (Rust; depends on the un-vendored pdbtbx fork, pulp 0.22.3, rayon 1.12.0)
 * cannot be compiled in this image, so there is no oracle/_ref build.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference tree).
 */
#ifndef SASA_ORACLE_H
#define SASA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/structures/atomic.rs:5-10 (NeighborData) */
typedef struct {
    float threshold_squared;
    uint32_t idx;
} oracle_neighbor_t;

/* CSR form of Vec<Vec<NeighborData>> (src/lib.rs:69-84 return value). */
typedef struct {
    size_t n_atoms;
    size_t *offsets;             /* n_atoms + 1 */
    oracle_neighbor_t *entries;  /* offsets[n_atoms] */
} oracle_neighbor_lists_t;

/* src/lib.rs:43-66  generate_sphere_points (golden-section spiral, SoA). */
void oracle_generate_sphere_points(size_t n_points, float *x, float *y, float *z);

/* src/structures/spatial_grid.rs:28-106 + :195-278 driven with explicit
 * cell_size / max_search_radius (what tests/units.rs:132-209 does), or with
 * the values of precompute_neighbors (src/lib.rs:69-84) when cell_size <= 0.
 * ids may be NULL (=> every atom distinct).  Lists are sorted by distance
 * (spatial_grid.rs:438-465).  Returns 0, or -1 on allocation failure. */
int oracle_neighbor_lists(const float *x, const float *y, const float *z,
                          const float *radius, const uint64_t *id, size_t n,
                          float probe_radius, float max_radius,
                          float cell_size, float max_search_radius,
                          oracle_neighbor_lists_t *out);
void oracle_neighbor_lists_free(oracle_neighbor_lists_t *lists);

/* src/lib.rs:249-298  calculate_sasa_internal, sequential (threads == 1).
 * simd_width = pulp lane count W that the reference would run with
 * (8 = AVX2+FMA, 16 = AVX-512, 4 = NEON, 1 = scalar); it only decides which of
 * the last (n_points mod W) points take the scalar remainder rule
 * (src/lib.rs:163-218).  out_points (optional) receives the integer number of
 * accessible points per atom; out_k (optional) the neighbour-list length.
 * Returns 0, or -1 on allocation failure / bad simd_width. */
int oracle_calculate_sasa_internal(const float *x, const float *y, const float *z,
                                   const float *radius, const uint64_t *id, size_t n,
                                   float probe_radius, size_t n_points, int simd_width,
                                   float *out_sasa, uint32_t *out_points, uint32_t *out_k);

/* The same with the reference's `threads` argument (src/lib.rs:278-290): 1 = sequential map over
 * the atoms, anything else a parallel map (rayon there, OpenMP here; < 1 = all cores).  Results
 * do not depend on it. */
int oracle_calculate_sasa_internal_mt(const float *x, const float *y, const float *z,
                                      const float *radius, const uint64_t *id, size_t n,
                                      float probe_radius, size_t n_points, int simd_width,
                                      int threads, float *out_sasa, uint32_t *out_points,
                                      uint32_t *out_k);

/* Directory mode of the reference (src/main.rs:375,439): independent
 * structures, each computed sequentially, spread over `threads` OpenMP
 * threads.  offsets has n_structures + 1 entries into the concatenated SoA. */
int oracle_calculate_sasa_batch(const float *x, const float *y, const float *z,
                                const float *radius, const uint64_t *id,
                                const uint32_t *offsets, size_t n_structures,
                                float probe_radius, size_t n_points, int simd_width,
                                int threads, float *out_sasa);

/* src/utils.rs:14-22 simd_sum applied per residue as in
 * src/options.rs:202-216: strictly sequential f32 sums of the atoms
 * [residue_offsets[k], residue_offsets[k+1]). */
void oracle_residue_sums(const float *atom_sasa, const uint32_t *residue_offsets,
                         size_t n_residues, float *out);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
