/*
 * sasa_oracle.c -- CPU restatement of RustSASA's Shrake-Rupley hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see sasa_oracle.h).  Parity status: PINNED against
 * the reference's golden vector FIXED_LOW_RES_ATOMS and its analytic tests;
 * see tests/test_oracle.py.
 *
 * Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp (oracle/Makefile).
 * Contraction is OFF so that the only fused operations are the explicit
 * fmaf() calls that restate pulp's mul_add_f32s (src/lib.rs:143-144).
 * All arithmetic is IEEE-754 binary32, evaluated in the reference's order.
 */
#include "sasa_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/utils/consts.rs:18-19: GOLDEN_RATIO = 1.618_034, ANGLE_INCREMENT =
 * 2.0 * PI * GOLDEN_RATIO, const-folded left to right in f32. */
static const float PI_F32 = 3.14159274101257324219f; /* std::f32::consts::PI */
static const float GOLDEN_RATIO = 1.618034f;

/* Rust `x as u32` on f32: saturating, NaN -> 0 (spatial_grid.rs:40-42,139-141). */
static uint32_t f32_as_u32(float v)
{
    if (!(v > 0.0f))
        return 0u; /* negatives, -0, NaN */
    if (v >= 4294967296.0f)
        return 4294967295u;
    return (uint32_t)v;
}

/* Rust `x as i32` on f32 (spatial_grid.rs:47). */
static int32_t f32_as_i32(float v)
{
    if (v != v)
        return 0;
    if (v >= 2147483648.0f)
        return INT32_MAX;
    if (v <= -2147483648.0f)
        return INT32_MIN;
    return (int32_t)v;
}

/* ------------------------------------------------------------------------ */
/* Scratch memory.  A single call allocates and frees with malloc like the    */
/* reference's Vecs.  The batch path (one structure per worker, main.rs:375)   */
/* gives every worker thread an arena that it rewinds between structures:      */
/* the grid, the 640-byte-per-atom neighbour slab and the list headers of a    */
/* structure are megabytes, and 128 threads that mmap / fault / munmap them    */
/* per structure spend their time under the process's one memory-map lock      */
/* instead of in the algorithm (round 3: 11x on 128 threads).  Results do not  */
/* depend on where scratch lives.                                             */
typedef struct arena_block {
    struct arena_block *next;
    size_t cap, used;
} arena_block_t;

static _Thread_local arena_block_t *tl_arena;
static _Thread_local int tl_arena_on;

static void *xmalloc(size_t bytes)
{
    if (!tl_arena_on)
        return malloc(bytes);
    const size_t hdr = (sizeof(arena_block_t) + 63u) & ~(size_t)63u;
    bytes = (bytes + 63u) & ~(size_t)63u;
    arena_block_t *b = tl_arena;
    if (!b || b->cap - b->used < bytes) {
        size_t cap = b ? 2 * b->cap : ((size_t)8 << 20);
        if (cap < bytes)
            cap = bytes;
        arena_block_t *nb = aligned_alloc(64, hdr + cap);
        if (!nb)
            return NULL;
        nb->next = b;
        nb->cap = cap;
        nb->used = 0;
        tl_arena = b = nb;
    }
    void *p = (char *)b + hdr + b->used;
    b->used += bytes;
    return p;
}

static void *xcalloc(size_t n, size_t size)
{
    void *p = xmalloc(n * size);
    if (p)
        memset(p, 0, n * size);
    return p;
}

static void xfree(void *p)
{
    if (!tl_arena_on)
        free(p);
}

/* grow a buffer whose old contents are not needed */
static void *xregrow(void *old, size_t bytes)
{
    if (!tl_arena_on)
        return realloc(old, bytes);
    return xmalloc(bytes);
}

static void arena_release(void)
{
    while (tl_arena) {
        arena_block_t *n = tl_arena->next;
        free(tl_arena);
        tl_arena = n;
    }
}

/* rewind; several blocks are replaced by one of their total size, so the next structure of that size fits */
static void arena_rewind(void)
{
    if (tl_arena && tl_arena->next) {
        size_t total = 0;
        for (arena_block_t *b = tl_arena; b; b = b->next)
            total += b->cap;
        arena_release();
        const size_t hdr = (sizeof(arena_block_t) + 63u) & ~(size_t)63u;
        arena_block_t *nb = aligned_alloc(64, hdr + total);
        if (nb) {
            nb->next = NULL;
            nb->cap = total;
            nb->used = 0;
            tl_arena = nb;
        }
    } else if (tl_arena) {
        tl_arena->used = 0;
    }
}

/* ------------------------------------------------------------------------ */
/* src/lib.rs:43-66 generate_sphere_points                                   */
/* ------------------------------------------------------------------------ */
void oracle_generate_sphere_points(size_t n_points, float *x, float *y, float *z)
{
    const float angle_increment = (2.0f * PI_F32) * GOLDEN_RATIO;
    const float inv_n_points = 1.0f / (float)n_points;         /* lib.rs:48 */
    for (size_t i = 0; i < n_points; i++) {
        float i_f32 = (float)i;                                /* lib.rs:51 */
        float t = i_f32 * inv_n_points;                        /* lib.rs:52 */
        float inclination = acosf(1.0f - 2.0f * t);            /* lib.rs:53 */
        float azimuth = angle_increment * i_f32;               /* lib.rs:54 */
        float sin_azimuth = sinf(azimuth);                     /* lib.rs:57 */
        float cos_azimuth = cosf(azimuth);
        float sin_inclination = sinf(inclination);             /* lib.rs:58 */
        x[i] = sin_inclination * cos_azimuth;                  /* lib.rs:60 */
        y[i] = sin_inclination * sin_azimuth;                  /* lib.rs:61 */
        z[i] = cosf(inclination);                              /* lib.rs:62 */
    }
}

/* ------------------------------------------------------------------------ */
/* src/structures/spatial_grid.rs                                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint32_t *atom_indices;      /* spatial_grid.rs:6  */
    float *px, *py, *pz, *radii; /* spatial_grid.rs:9-14 */
    uint32_t *cell_starts;       /* spatial_grid.rs:17 */
    uint32_t dims[3];            /* spatial_grid.rs:20 */
    size_t num_cells;
    int32_t (*half_shell)[3];    /* spatial_grid.rs:24 */
    size_t n_half_shell;
} grid_t;

typedef struct {
    const float *x, *y, *z, *r;
    const uint64_t *id; /* NULL => id = index */
    size_t n;
} atoms_t;

static inline uint64_t atom_id(const atoms_t *a, size_t i)
{
    return a->id ? a->id[i] : (uint64_t)i;
}

/* spatial_grid.rs:133-143 get_cell_index_static */
static inline size_t cell_index(float x, float y, float z, const float min_b[3],
                                float inv_cell, const uint32_t dims[3])
{
    uint32_t cx = f32_as_u32((x - min_b[0]) * inv_cell);
    uint32_t cy = f32_as_u32((y - min_b[1]) * inv_cell);
    uint32_t cz = f32_as_u32((z - min_b[2]) * inv_cell);
    return (size_t)(cx + cy * dims[0] + cz * dims[0] * dims[1]);
}

static void grid_free(grid_t *g)
{
    xfree(g->atom_indices);
    xfree(g->px);
    xfree(g->py);
    xfree(g->pz);
    xfree(g->radii);
    xfree(g->cell_starts);
    xfree(g->half_shell);
    memset(g, 0, sizeof *g);
}

/* spatial_grid.rs:28-106 SpatialGrid::new (active_indices = 0..n, lib.rs:255) */
static int grid_new(grid_t *g, const atoms_t *a, float cell_size, float max_search_radius)
{
    memset(g, 0, sizeof *g);
    size_t n = a->n;

    /* spatial_grid.rs:108-130 calculate_bounds(padding = cell_size) */
    float min_b[3] = {INFINITY, INFINITY, INFINITY};
    float max_b[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; i++) {
        min_b[0] = fminf(min_b[0], a->x[i]);
        max_b[0] = fmaxf(max_b[0], a->x[i]);
        min_b[1] = fminf(min_b[1], a->y[i]);
        max_b[1] = fmaxf(max_b[1], a->y[i]);
        min_b[2] = fminf(min_b[2], a->z[i]);
        max_b[2] = fmaxf(max_b[2], a->z[i]);
    }
    for (int k = 0; k < 3; k++) {
        min_b[k] -= cell_size;
        max_b[k] += cell_size;
    }
    float inv_cell = 1.0f / cell_size;                                   /* :36 */
    for (int k = 0; k < 3; k++)                                          /* :39-43 */
        g->dims[k] = f32_as_u32(ceilf((max_b[k] - min_b[k]) * inv_cell)) + 1u;
    g->num_cells = (size_t)(g->dims[0] * g->dims[1] * g->dims[2]);       /* :44 */

    /* :47-50, :174-192 half-shell offsets */
    int32_t extent = f32_as_i32(ceilf(max_search_radius / cell_size));
    size_t side = (size_t)(2 * extent + 1);
    g->half_shell = xmalloc(sizeof(int32_t[3]) * side * side * side);
    if (!g->half_shell)
        return -1;
    for (int32_t dz = -extent; dz <= extent; dz++)
        for (int32_t dy = -extent; dy <= extent; dy++)
            for (int32_t dx = -extent; dx <= extent; dx++) {
                int include = (dz > 0) || (dz == 0 && dy > 0) || (dz == 0 && dy == 0 && dx >= 0);
                if (include) {
                    g->half_shell[g->n_half_shell][0] = dx;
                    g->half_shell[g->n_half_shell][1] = dy;
                    g->half_shell[g->n_half_shell][2] = dz;
                    g->n_half_shell++;
                }
            }

    /* :53-68 counts + exclusive prefix */
    g->cell_starts = xcalloc(g->num_cells + 1, sizeof(uint32_t));
    uint32_t *write_pos = xmalloc(sizeof(uint32_t) * (g->num_cells ? g->num_cells : 1));
    uint32_t *cell_of = xmalloc(sizeof(uint32_t) * (n ? n : 1));
    g->atom_indices = xmalloc(sizeof(uint32_t) * (n ? n : 1));
    g->px = xmalloc(sizeof(float) * (n ? n : 1));
    g->py = xmalloc(sizeof(float) * (n ? n : 1));
    g->pz = xmalloc(sizeof(float) * (n ? n : 1));
    g->radii = xmalloc(sizeof(float) * (n ? n : 1));
    if (!g->cell_starts || !write_pos || !cell_of || !g->atom_indices || !g->px || !g->py ||
        !g->pz || !g->radii) {
        xfree(write_pos);
        xfree(cell_of);
        grid_free(g);
        return -1;
    }
    for (size_t i = 0; i < n; i++) {
        size_t c = cell_index(a->x[i], a->y[i], a->z[i], min_b, inv_cell, g->dims);
        cell_of[i] = (uint32_t)c;
        g->cell_starts[c + 1] += 1; /* counts, shifted by one */
    }
    for (size_t c = 0; c < g->num_cells; c++)
        g->cell_starts[c + 1] += g->cell_starts[c];
    /* :70-93 stable scatter into the cell-sorted SoA */
    memcpy(write_pos, g->cell_starts, sizeof(uint32_t) * g->num_cells);
    for (size_t i = 0; i < n; i++) {
        uint32_t wp = write_pos[cell_of[i]]++;
        g->atom_indices[wp] = (uint32_t)i;
        g->px[wp] = a->x[i];
        g->py[wp] = a->y[i];
        g->pz[wp] = a->z[i];
        g->radii[wp] = a->r[i];
    }
    xfree(write_pos);
    xfree(cell_of);
    return 0;
}

/* Vec<NeighborData> with_capacity(80) (spatial_grid.rs:212-213). */
typedef struct {
    oracle_neighbor_t *p;
    uint32_t len, cap;
    int owned;
} nvec_t;

static inline int nvec_push(nvec_t *v, uint32_t idx, float thr)
{
    if (v->len == v->cap) {
        uint32_t ncap = v->cap * 2;
        oracle_neighbor_t *np = xmalloc(sizeof(oracle_neighbor_t) * ncap);
        if (!np)
            return -1;
        memcpy(np, v->p, sizeof(oracle_neighbor_t) * v->len);
        if (v->owned)
            xfree(v->p);
        v->p = np;
        v->cap = ncap;
        v->owned = 1;
    }
    v->p[v->len].idx = idx;
    v->p[v->len].threshold_squared = thr;
    v->len++;
    return 0;
}

/* spatial_grid.rs:282-356 (self cell, j > i) and :360-436 (other cell, all j):
 * the two bodies are identical apart from the j range. */
static inline int process_cell_pair(const grid_t *g, const atoms_t *a, size_t start_a,
                                    size_t end_a, size_t start_b, size_t end_b, int is_self,
                                    float probe, float max_radius, float max_search_sq,
                                    nvec_t *lists)
{
    for (size_t i = start_a; i < end_a; i++) {
        size_t orig_i = g->atom_indices[i];
        float xi = g->px[i], yi = g->py[i], zi = g->pz[i], ri = g->radii[i];
        uint64_t id_i = atom_id(a, orig_i);
        float sr_i = ri + max_radius + 2.0f * probe;                     /* :307 */
        float sr_i_sq = sr_i * sr_i;
        for (size_t j = is_self ? i + 1 : start_b; j < end_b; j++) {
            size_t orig_j = g->atom_indices[j];
            if (atom_id(a, orig_j) == id_i)                              /* :314 */
                continue;
            float dx = xi - g->px[j];
            float dy = yi - g->py[j];
            float dz = zi - g->pz[j];
            float dist_sq = dx * dx + dy * dy + dz * dz;                 /* :321 */
            if (dist_sq > max_search_sq)                                 /* :324 */
                continue;
            float rj = g->radii[j];
            float sr_j = rj + max_radius + 2.0f * probe;                 /* :331 */
            float sr_j_sq = sr_j * sr_j;
            if (dist_sq <= sr_i_sq) {                                    /* :335 */
                float thresh_j = rj + probe;
                if (nvec_push(&lists[orig_i], (uint32_t)orig_j, thresh_j * thresh_j))
                    return -1;
            }
            if (dist_sq <= sr_j_sq) {                                    /* :344 */
                float thresh_i = ri + probe;
                if (nvec_push(&lists[orig_j], (uint32_t)orig_i, thresh_i * thresh_i))
                    return -1;
            }
        }
    }
    return 0;
}

typedef struct {
    float key;
    oracle_neighbor_t e;
} keyed_t;

static int keyed_cmp(const void *pa, const void *pb)
{
    float a = ((const keyed_t *)pa)->key, b = ((const keyed_t *)pb)->key;
    return (a > b) - (a < b); /* partial_cmp(..).unwrap_or(Equal), :462 */
}

/* spatial_grid.rs:438-465 sort_neighbors_by_distance.  The order among equal
 * keys is unspecified in the reference (sort_unstable_by) and has no effect on
 * results: occlusion is an OR over the whole list. */
static int sort_neighbors(const atoms_t *a, size_t center_idx, nvec_t *v, keyed_t **scratch,
                          size_t *scratch_cap)
{
    if (v->len <= 1)
        return 0;
    if (*scratch_cap < v->len) {
        keyed_t *ns = xregrow(*scratch, sizeof(keyed_t) * v->len * 2);
        if (!ns)
            return -1;
        *scratch = ns;
        *scratch_cap = (size_t)v->len * 2;
    }
    keyed_t *k = *scratch;
    float cx = a->x[center_idx], cy = a->y[center_idx], cz = a->z[center_idx];
    for (uint32_t t = 0; t < v->len; t++) {
        size_t j = v->p[t].idx;
        float ex = cx - a->x[j], ey = cy - a->y[j], ez = cz - a->z[j];
        k[t].key = ex * ex + ey * ey + ez * ez; /* powi(2) sums, :455-457 */
        k[t].e = v->p[t];
    }
    if (v->len <= 96) { /* insertion sort: lists are ~45 long */
        for (uint32_t t = 1; t < v->len; t++) {
            keyed_t cur = k[t];
            uint32_t s = t;
            while (s > 0 && k[s - 1].key > cur.key) {
                k[s] = k[s - 1];
                s--;
            }
            k[s] = cur;
        }
    } else {
        qsort(k, v->len, sizeof(keyed_t), keyed_cmp);
    }
    for (uint32_t t = 0; t < v->len; t++)
        v->p[t] = k[t].e;
    return 0;
}

/* spatial_grid.rs:195-278 build_all_neighbor_lists (active = all atoms). */
static int build_all_neighbor_lists(const grid_t *g, const atoms_t *a, float probe,
                                    float max_radius, nvec_t **out_lists,
                                    oracle_neighbor_t **out_slab)
{
    size_t n = a->n;
    const uint32_t INITIAL_CAP = 80;                                     /* :213 */
    nvec_t *lists = xmalloc(sizeof(nvec_t) * (n ? n : 1));
    oracle_neighbor_t *slab = xmalloc(sizeof(oracle_neighbor_t) * INITIAL_CAP * (n ? n : 1));
    if (!lists || !slab) {
        xfree(lists);
        xfree(slab);
        return -1;
    }
    for (size_t i = 0; i < n; i++) {
        lists[i].p = slab + i * INITIAL_CAP;
        lists[i].len = 0;
        lists[i].cap = INITIAL_CAP;
        lists[i].owned = 0;
    }
    float max_search = max_radius + max_radius + 2.0f * probe;           /* :216 */
    float max_search_sq = max_search * max_search;
    int rc = 0;
    uint32_t dxy = g->dims[0] * g->dims[1];
    for (size_t cell_a = 0; cell_a < g->num_cells && !rc; cell_a++) {    /* :220 */
        size_t start_a = g->cell_starts[cell_a], end_a = g->cell_starts[cell_a + 1];
        if (start_a == end_a)
            continue;
        /* :160-167 index_to_cell_coords */
        uint32_t idx = (uint32_t)cell_a;
        int32_t cz = (int32_t)(idx / dxy);
        uint32_t rem = idx % dxy;
        int32_t cy = (int32_t)(rem / g->dims[0]);
        int32_t cx = (int32_t)(rem % g->dims[0]);
        for (size_t h = 0; h < g->n_half_shell; h++) {                   /* :231 */
            int32_t ox = g->half_shell[h][0], oy = g->half_shell[h][1], oz = g->half_shell[h][2];
            int32_t bx = cx + ox, by = cy + oy, bz = cz + oz;
            /* :146-157 cell_coords_to_index */
            if (bx < 0 || by < 0 || bz < 0)
                continue;
            if ((uint32_t)bx >= g->dims[0] || (uint32_t)by >= g->dims[1] ||
                (uint32_t)bz >= g->dims[2])
                continue;
            size_t cell_b = (size_t)((uint32_t)bx + (uint32_t)by * g->dims[0] + (uint32_t)bz * dxy);
            size_t start_b = g->cell_starts[cell_b], end_b = g->cell_starts[cell_b + 1];
            if (start_b == end_b)
                continue;
            int is_self = (ox == 0 && oy == 0 && oz == 0);
            rc = process_cell_pair(g, a, start_a, end_a, is_self ? start_a : start_b,
                                   is_self ? end_a : end_b, is_self, probe, max_radius,
                                   max_search_sq, lists);
            if (rc)
                break;
        }
    }
    keyed_t *scratch = NULL;
    size_t scratch_cap = 0;
    for (size_t i = 0; i < n && !rc; i++)                                /* :275 */
        rc = sort_neighbors(a, i, &lists[i], &scratch, &scratch_cap);
    xfree(scratch);
    if (rc) {
        for (size_t i = 0; i < n; i++)
            if (lists[i].owned)
                xfree(lists[i].p);
        xfree(lists);
        xfree(slab);
        return -1;
    }
    *out_lists = lists;
    *out_slab = slab;
    return 0;
}

static void free_lists(nvec_t *lists, oracle_neighbor_t *slab, size_t n)
{
    if (lists)
        for (size_t i = 0; i < n; i++)
            if (lists[i].owned)
                xfree(lists[i].p);
    xfree(lists);
    xfree(slab);
}

/* src/lib.rs:69-84 precompute_neighbors */
static int precompute_neighbors(const atoms_t *a, float probe, float max_radii, nvec_t **lists,
                                oracle_neighbor_t **slab)
{
    float cell_size = probe + max_radii;                                 /* lib.rs:76 */
    float max_search_radius = max_radii + max_radii + 2.0f * probe;      /* lib.rs:80 */
    grid_t g;
    if (grid_new(&g, a, cell_size, max_search_radius))
        return -1;
    int rc = build_all_neighbor_lists(&g, a, probe, max_radii, lists, slab);
    grid_free(&g);
    return rc;
}

int oracle_neighbor_lists(const float *x, const float *y, const float *z, const float *radius,
                          const uint64_t *id, size_t n, float probe_radius, float max_radius,
                          float cell_size, float max_search_radius,
                          oracle_neighbor_lists_t *out)
{
    atoms_t a = {x, y, z, radius, id, n};
    nvec_t *lists = NULL;
    oracle_neighbor_t *slab = NULL;
    int rc;
    memset(out, 0, sizeof *out);
    if (cell_size <= 0.0f) {
        rc = precompute_neighbors(&a, probe_radius, max_radius, &lists, &slab);
    } else {
        grid_t g;
        if (grid_new(&g, &a, cell_size, max_search_radius))
            return -1;
        rc = build_all_neighbor_lists(&g, &a, probe_radius, max_radius, &lists, &slab);
        grid_free(&g);
    }
    if (rc)
        return -1;
    out->n_atoms = n;
    out->offsets = xmalloc(sizeof(size_t) * (n + 1));
    size_t total = 0;
    for (size_t i = 0; i < n; i++)
        total += lists[i].len;
    out->entries = xmalloc(sizeof(oracle_neighbor_t) * (total ? total : 1));
    if (!out->offsets || !out->entries) {
        free_lists(lists, slab, n);
        oracle_neighbor_lists_free(out);
        return -1;
    }
    size_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        out->offsets[i] = pos;
        memcpy(out->entries + pos, lists[i].p, sizeof(oracle_neighbor_t) * lists[i].len);
        pos += lists[i].len;
    }
    out->offsets[n] = pos;
    free_lists(lists, slab, n);
    return 0;
}

void oracle_neighbor_lists_free(oracle_neighbor_lists_t *l)
{
    xfree(l->offsets);
    xfree(l->entries);
    memset(l, 0, sizeof *l);
}

/* ------------------------------------------------------------------------ */
/* src/lib.rs:86-224 AtomSasaKernel::with_simd, for a lane count W           */
/* ------------------------------------------------------------------------ */
static inline __attribute__((always_inline)) float
atom_kernel(const int W, const atoms_t *a, size_t atom_index, const nvec_t *nb, const float *spx,
            const float *spy, const float *spz, size_t n_points, float probe,
            uint32_t *out_points)
{
    const float cx = a->x[atom_index], cy = a->y[atom_index], cz = a->z[atom_index];
    const uint64_t my_id = atom_id(a, atom_index);
    const float r = a->r[atom_index] + probe;                            /* lib.rs:101 */
    const float r2 = r * r;                                              /* lib.rs:102 */
    const size_t n_chunks = n_points / (size_t)W;   /* S::as_simd_f32s, lib.rs:104-106 */
    const size_t n_rem = n_points - n_chunks * (size_t)W;

    float accessible_points = 0.0f;                                      /* lib.rs:108 */

    for (size_t c = 0; c < n_chunks; c++) {                              /* lib.rs:115 */
        const float *sx = spx + c * (size_t)W, *sy = spy + c * (size_t)W,
                    *sz = spz + c * (size_t)W;
        int mask[16];
        for (int l = 0; l < W; l++)
            mask[l] = 0;                                                 /* lib.rs:121 */
        for (uint32_t k = 0; k < nb->len; k++) {                         /* lib.rs:123 */
            size_t j = nb->p[k].idx;
            if (atom_id(a, j) == my_id)                                  /* lib.rs:124 */
                continue;
            float vx = cx - a->x[j];                                     /* lib.rs:129-131 */
            float vy = cy - a->y[j];
            float vz = cz - a->z[j];
            float v_mag_sq = vx * vx + vy * vy + vz * vz;                /* lib.rs:132-133 */
            float t = nb->p[k].threshold_squared;
            float limit = (t - v_mag_sq - r2) / (2.0f * r);              /* lib.rs:136 */
            int n_occ = 0;
            for (int l = 0; l < W; l++) {
                /* lib.rs:143-144: mul_add(sx,vx, mul_add(sy,vy, sz*vz)) */
                float dot = fmaf(sx[l], vx, fmaf(sy[l], vy, sz[l] * vz));
                mask[l] |= (dot < limit);                                /* lib.rs:146-147 */
                n_occ += mask[l];
            }
            if (n_occ == W)                                              /* lib.rs:149-152 */
                break;
        }
        for (int l = 0; l < W; l++)                                      /* lib.rs:156-159 */
            accessible_points += mask[l] ? 0.0f : 1.0f;
    }

    /* remainder, lib.rs:163-218 */
    {
        const float *sxr = spx + n_chunks * (size_t)W, *syr = spy + n_chunks * (size_t)W,
                    *szr = spz + n_chunks * (size_t)W;
        size_t current_nb = 0;                                           /* lib.rs:163 */
        for (size_t i = 0; i < n_rem; i++) {
            float sx = sxr[i], sy = syr[i], sz = szr[i];
            int occluded = 0;
            if (current_nb < nb->len) {                                  /* lib.rs:171 */
                size_t j = nb->p[current_nb].idx;
                if (atom_id(a, j) != my_id) {                            /* lib.rs:173 */
                    float vx = cx - a->x[j], vy = cy - a->y[j], vz = cz - a->z[j];
                    float v_mag_sq = vx * vx + vy * vy + vz * vz;
                    float t = nb->p[current_nb].threshold_squared;
                    float limit = (t - v_mag_sq - r2) / (2.0f * r);
                    float dot = sx * vx + sy * vy + sz * vz;             /* lib.rs:185 */
                    if (dot <= limit)                                    /* lib.rs:186 */
                        occluded = 1;
                }
            }
            if (!occluded) {                                             /* lib.rs:193 */
                for (uint32_t k = 0; k < nb->len; k++) {
                    size_t j = nb->p[k].idx;
                    if (atom_id(a, j) == my_id)
                        continue;
                    float vx = cx - a->x[j], vy = cy - a->y[j], vz = cz - a->z[j];
                    float v_mag_sq = vx * vx + vy * vy + vz * vz;
                    float t = nb->p[k].threshold_squared;
                    float limit = (t - v_mag_sq - r2) / (2.0f * r);
                    float dot = sx * vx + sy * vy + sz * vz;             /* lib.rs:206 */
                    if (dot <= limit) {                                  /* lib.rs:207 */
                        occluded = 1;
                        current_nb = k;                                  /* lib.rs:209 */
                        break;
                    }
                }
            }
            if (!occluded)
                accessible_points += 1.0f;                               /* lib.rs:216 */
        }
    }

    if (out_points)
        *out_points = (uint32_t)accessible_points;
    float surface_area = 4.0f * PI_F32 * r2;                             /* lib.rs:220 */
    float inv_n_points = 1.0f / (float)n_points;                         /* lib.rs:221 */
    return surface_area * accessible_points * inv_n_points;              /* lib.rs:222 */
}

#define DEFINE_KERNEL(W)                                                                       \
    static float atom_kernel_w##W(const atoms_t *a, size_t i, const nvec_t *nb, const float *px, \
                                  const float *py, const float *pz, size_t np, float probe,    \
                                  uint32_t *op)                                                \
    {                                                                                          \
        return atom_kernel(W, a, i, nb, px, py, pz, np, probe, op);                            \
    }
DEFINE_KERNEL(1)
DEFINE_KERNEL(4)
DEFINE_KERNEL(8)
DEFINE_KERNEL(16)

typedef float (*kernel_fn)(const atoms_t *, size_t, const nvec_t *, const float *, const float *,
                           const float *, size_t, float, uint32_t *);

static kernel_fn pick_kernel(int simd_width)
{
    switch (simd_width) {
    case 1: return atom_kernel_w1;
    case 4: return atom_kernel_w4;
    case 8: return atom_kernel_w8;
    case 16: return atom_kernel_w16;
    default: return NULL;
    }
}

/* src/lib.rs:249-298 calculate_sasa_internal with a caller-provided lattice. */
static int calculate_with_lattice(const atoms_t *a, float probe, size_t n_points,
                                  const float *spx, const float *spy, const float *spz,
                                  kernel_fn kern, float *out_sasa, uint32_t *out_points,
                                  uint32_t *out_k, int atom_threads)
{
    size_t n = a->n;
    if (n == 0)
        return 0;
    float max_radii = 0.0f;                                              /* lib.rs:259-262 */
    for (size_t i = 0; i < n; i++)
        max_radii = fmaxf(max_radii, a->r[i]);
    nvec_t *lists = NULL;
    oracle_neighbor_t *slab = NULL;
    if (precompute_neighbors(a, probe, max_radii, &lists, &slab))       /* lib.rs:264 */
        return -1;
    /* lib.rs:278-290: sequential when threads == 1, otherwise a parallel map over the atoms
     * (rayon par_iter there, OpenMP here); every atom's value is independent of the others */
#pragma omp parallel for schedule(dynamic, 512) num_threads(atom_threads) if (atom_threads > 1)
    for (long i = 0; i < (long)n; i++) {                                 /* lib.rs:278-283 */
        out_sasa[i] = kern(a, i, &lists[i], spx, spy, spz, n_points, probe,
                           out_points ? &out_points[i] : NULL);
        if (out_k)
            out_k[i] = lists[i].len;
    }
    free_lists(lists, slab, n);
    return 0;
}

int oracle_calculate_sasa_internal(const float *x, const float *y, const float *z,
                                   const float *radius, const uint64_t *id, size_t n,
                                   float probe_radius, size_t n_points, int simd_width,
                                   float *out_sasa, uint32_t *out_points, uint32_t *out_k)
{
    return oracle_calculate_sasa_internal_mt(x, y, z, radius, id, n, probe_radius, n_points,
                                             simd_width, 1, out_sasa, out_points, out_k);
}

int oracle_calculate_sasa_internal_mt(const float *x, const float *y, const float *z,
                                      const float *radius, const uint64_t *id, size_t n,
                                      float probe_radius, size_t n_points, int simd_width,
                                      int threads, float *out_sasa, uint32_t *out_points,
                                      uint32_t *out_k)
{
    if (threads < 1)
        threads = oracle_max_threads();
    kernel_fn kern = pick_kernel(simd_width);
    if (!kern)
        return -1;
    atoms_t a = {x, y, z, radius, id, n};
    float *sp = xmalloc(sizeof(float) * 3 * (n_points ? n_points : 1));
    if (!sp)
        return -1;
    oracle_generate_sphere_points(n_points, sp, sp + n_points, sp + 2 * n_points); /* :257 */
    int rc = calculate_with_lattice(&a, probe_radius, n_points, sp, sp + n_points,
                                    sp + 2 * n_points, kern, out_sasa, out_points, out_k, threads);
    xfree(sp);
    return rc;
}

int oracle_calculate_sasa_batch(const float *x, const float *y, const float *z,
                                const float *radius, const uint64_t *id,
                                const uint32_t *offsets, size_t n_structures,
                                float probe_radius, size_t n_points, int simd_width,
                                int threads, float *out_sasa)
{
    kernel_fn kern = pick_kernel(simd_width);
    if (!kern)
        return -1;
    int failed = 0;
    if (threads < 1)
        threads = oracle_max_threads();
    /* each worker regenerates the lattice per structure exactly as the
     * reference does per calculate_sasa_internal call (lib.rs:257); its scratch
     * (grid, neighbour slab, lattice) lives in the worker's arena, rewound per structure */
#pragma omp parallel num_threads(threads)
    {
        tl_arena_on = 1;
#pragma omp for schedule(dynamic, 1)
        for (long s = 0; s < (long)n_structures; s++) {
            arena_rewind();
            size_t b = offsets[s], e = offsets[s + 1];
            atoms_t a = {x + b, y + b, z + b, radius + b, id ? id + b : NULL, e - b};
            float *sp = xmalloc(sizeof(float) * 3 * (n_points ? n_points : 1));
            if (!sp) {
#pragma omp atomic write
                failed = 1;
                continue;
            }
            oracle_generate_sphere_points(n_points, sp, sp + n_points, sp + 2 * n_points);
            if (calculate_with_lattice(&a, probe_radius, n_points, sp, sp + n_points,
                                       sp + 2 * n_points, kern, out_sasa + b, NULL, NULL, 1)) {
#pragma omp atomic write
                failed = 1;
            }
        }
        tl_arena_on = 0;
        arena_release();
    }
    return failed ? -1 : 0;
}

/* src/utils.rs:14-22 + src/options.rs:209-216 */
void oracle_residue_sums(const float *atom_sasa, const uint32_t *residue_offsets,
                         size_t n_residues, float *out)
{
    for (size_t k = 0; k < n_residues; k++) {
        float total = 0.0f;
        for (uint32_t i = residue_offsets[k]; i < residue_offsets[k + 1]; i++)
            total += atom_sasa[i];
        out[k] = total;
    }
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
