// Host-pointer entry points of the C ABI (include/rustsasa_amd.h): the small-batch path, the pipelined host batch
// (rsasa_calculate_sasa_batch), the stream of host batches, the per-structure calls and trajectories.  Host code only.
#include "engine_internal.h"

namespace {

// Host-side restatement of make_grid (kernels.hip) for the small-batch path: the same IEEE f32
// operations (this file is compiled with -ffp-contract=off like the device code).
bool small_grid(const float mn_in[3], const float mx_in[3], float max_r, float probe, uint32_t n_atoms,
                StructGrid *out)
{
    const float cell = probe + max_r;  // lib.rs:76
    const float inv = 1.0f / cell;     // spatial_grid.rs:36
    if (!(cell > 0.0f) || !(inv < INFINITY) || !(cell < INFINITY)) return false;
    StructGrid g{};
    unsigned long long nc = 1;
    uint32_t d[3];
    const float mn[3] = {mn_in[0] - cell, mn_in[1] - cell, mn_in[2] - cell};
    const float mx[3] = {mx_in[0] + cell, mx_in[1] + cell, mx_in[2] + cell};
    for (int k = 0; k < 3; k++) {  // spatial_grid.rs:39-43
        const float e = ceilf((mx[k] - mn[k]) * inv);
        if (!(e >= 0.0f) || e >= 2147483648.0f) return false;
        d[k] = (uint32_t)e + 1u;
        nc *= d[k];
        if (nc > 64ull * kWindowCells) return false;  // (a sparse structure: the general path)
    }
    g.min_x = mn[0]; g.min_y = mn[1]; g.min_z = mn[2];
    g.inv_cell = inv;
    g.dim_x = d[0]; g.dim_y = d[1]; g.dim_z = d[2];
    g.max_r = max_r;
    g.cell_size = cell;
    g.n_cells = (uint32_t)nc;
    g.n_atoms = n_atoms;
    g.in_lds = 1u;  // fewer than kLdsMaxAtoms atoms (kSmallAtoms): binned in LDS
    *out = g;
    return true;
}

constexpr size_t kSmallAtoms = 32768, kSmallStructures = 256;
constexpr size_t kSingleAtoms = 8192;  // one structure up to this size: its atoms are read from pinned host memory

}  // namespace

namespace rsasa {

// Bounding box, largest radius and grid of one structure, computed by the host (the small path and the call combiner:
// a structure of a few thousand atoms is scanned faster than a bounds kernel is launched).  false: anything unusual -
// non-finite input, a probe + radius that is no cell size, a grid too sparse for the LDS windows - which the general
// path validates and reports.  atom_begin / sorted_base / cell_base are the caller's to fill in.
bool small_structure_grid(const SmallSource &src, uint32_t begin, uint32_t end, float probe, StructGrid *out)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mr = 0.0f;
    bool finite = true, odd_r = false;
    for (uint32_t i = begin; i < end; i++) {
        float p[3], r;
        if (src.aos) { p[0] = src.aos[i].position[0]; p[1] = src.aos[i].position[1]; p[2] = src.aos[i].position[2]; r = src.aos[i].radius; }
        else { p[0] = src.x[i]; p[1] = src.y[i]; p[2] = src.z[i]; r = src.radius[i]; }
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(mn[k], p[k]);
            mx[k] = fmaxf(mx[k], p[k]);
            finite &= std::isfinite(p[k]);
        }
        mr = fmaxf(mr, r);
        finite &= std::isfinite(r);
        odd_r |= !(r >= 0.0f && r <= 64.0f) || !(fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2])) <= 1e8f);
    }
    if (!finite || !small_grid(mn, mx, mr, probe, end - begin, out)) return false;
    out->odd_radii = (odd_r ? 1u : 0u) | (grid_group_shift(out->n_atoms, out->n_cells) << 8);
    return true;
}

// Atoms [begin, end) of the source as columns, written at atom `at` of the staging block's columns.
void small_fill(const SmallSource &src, uint32_t begin, uint32_t end, const SmallLayout &lay, char *h, size_t at)
{
    const size_t n = end - begin;
    float *x = (float *)(h + lay.o_x) + at, *y = (float *)(h + lay.o_y) + at, *z = (float *)(h + lay.o_z) + at, *r = (float *)(h + lay.o_r) + at;
    uint64_t *id = lay.has_id ? (uint64_t *)(h + lay.o_id) + at : nullptr;
    if (src.aos) {
        const rsasa_atom_t *a = src.aos + begin;
        for (size_t i = 0; i < n; i++) {
            x[i] = a[i].position[0];
            y[i] = a[i].position[1];
            z[i] = a[i].position[2];
            r[i] = a[i].radius;
        }
        if (id)
            for (size_t i = 0; i < n; i++) id[i] = a[i].id;
        return;
    }
    std::memcpy(x, src.x + begin, n * 4);
    std::memcpy(y, src.y + begin, n * 4);
    std::memcpy(z, src.z + begin, n * 4);
    std::memcpy(r, src.radius + begin, n * 4);
    if (id && src.id) std::memcpy(id, src.id + begin, n * 8);
}

// The same atoms as 24-byte records (rsasa_atom_t) at record `at` of `recs`: what the call combiner's blocks hold.
void small_fill_records(const SmallSource &src, uint32_t begin, uint32_t end, rsasa_atom_t *recs, size_t at)
{
    const size_t n = end - begin;
    rsasa_atom_t *o = recs + at;
    if (src.aos) {
        std::memcpy(o, src.aos + begin, n * sizeof(rsasa_atom_t));
        return;
    }
    for (size_t i = 0; i < n; i++) {
        o[i].position[0] = src.x[begin + i];
        o[i].position[1] = src.y[begin + i];
        o[i].position[2] = src.z[begin + i];
        o[i].radius = src.radius[begin + i];
        o[i].id = src.id ? src.id[begin + i] : 0u;
    }
}

std::atomic<uint64_t> g_small_trace_ns[2];  // merged batches: header + launches queued | waited for (combine.cpp prints them)

// staging layout (16-byte aligned sections): status | grids | windows | x | y | z | r | id | residue offsets; results behind
SmallLayout small_layout(size_t S, size_t N, size_t W, size_t R, bool has_id, unsigned long long total_cells16)
{
    auto up = [](size_t v) { return (v + 15) & ~size_t(15); };
    SmallLayout l;
    l.S = S; l.N = N; l.W = W; l.R = R; l.has_id = has_id;
    l.tail_begin = (total_cells16 / 2ull + 1023ull) & ~1023ull;
    l.o_grid = 64; l.o_win = l.o_grid + up(S * sizeof(StructGrid)); l.o_x = l.o_win + up(W * sizeof(uint4));
    l.o_y = l.o_x + up(N * 4); l.o_z = l.o_y + up(N * 4); l.o_r = l.o_z + up(N * 4); l.o_id = l.o_r + up(N * 4);
    l.o_res = l.o_id + (has_id ? up(N * 8) : 0);
    l.in_bytes = l.o_res + (R ? up((R + 1) * 4) : 0);
    l.o_oa = 0; l.o_or = up(N * 4); l.out_bytes = l.o_or + up(R * 4);
    return l;
}

// The context's pinned staging block and device buffers for a batch of this layout.  On return ctx->h_small holds
// in_bytes of input staging followed by out_bytes of result staging.
int small_reserve(rsasa_context *ctx, const SmallLayout &l, const Lattice &lat, bool own_staging)
{
    int rc;
    const size_t host_bytes = l.in_bytes + l.out_bytes;
    if (own_staging && host_bytes > ctx->h_small_cap) {
        if (ctx->h_small) {
            RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            RS_HIP(ctx, hipHostFree(ctx->h_small));
            ctx->h_small = nullptr;
            ctx->h_small_cap = 0;
        }
        RS_HIP(ctx, hipHostMalloc(&ctx->h_small, host_bytes * 2, hipHostMallocDefault));
        ctx->h_small_cap = host_bytes * 2;
    }
    const size_t N = l.N;
    if ((rc = reserve(ctx, ctx->small_in, l.in_bytes))) return rc;
    if (own_staging && (rc = reserve(ctx, ctx->small_out, l.out_bytes))) return rc;
    if ((rc = reserve(ctx, ctx->sid_sorted, N * 4))) return rc;
    if ((rc = reserve(ctx, ctx->deferred_list, N * 4))) return rc;
    if ((rc = reserve(ctx, ctx->claim, kClaimBytes))) return rc;
    if ((rc = reserve(ctx, ctx->rank_of, N * 4))) return rc;
    if ((rc = reserve(ctx, ctx->cells, (size_t)(l.tail_begin + 8) * 4))) return rc;
    if ((rc = reserve(ctx, ctx->sorted_xyzr, N * 16))) return rc;
    if ((rc = reserve(ctx, ctx->sorted_orig, N * 4))) return rc;
    const bool keep_ids = l.has_id && !occlusion_uses_mx(ctx->tuning, lat, (uint32_t)N);
    if (keep_ids && (rc = reserve(ctx, ctx->sorted_id, N * 8))) return rc;
    if (l.has_id && (rc = reserve(ctx, ctx->sorted_id32, N * 4))) return rc;
    return RSASA_OK;
}

// Runs the staged batch: ONE upload (inputs + grids + status), four launches (LDS binning, the two occlusion kernels,
// residue sums), one download; returns when the results are in the staging block (h_small + in_bytes).  `grids` /
// `windows` are copied into the block here; the columns (small_fill) and residue offsets are already there.
int small_run(rsasa_context *ctx, const SmallLayout &l, const Lattice &lat, const StructGrid *grids, const uint4 *windows, char *h, char *hout,
              const void *records)
{
    const bool zero_copy = records != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t S = l.S, N = l.N, W = l.W, R = l.R;
    BatchStatus stt{};
    stt.total_cells = l.tail_begin;
    stt.tail_cell_begin = l.tail_begin;
    stt.tail_atom_base = (uint32_t)N;
    stt.n_windows = (uint32_t)W;
    std::memcpy(h, &stt, sizeof stt);
    std::memcpy(h + l.o_grid, grids, S * sizeof(StructGrid));
    if (W) std::memcpy(h + l.o_win, windows, W * sizeof(uint4));
    hipStream_t st = ctx->stream;
    char *d = (char *)ctx->small_in.p, *dout = (char *)ctx->small_out.p;
    // One structure of a few thousand atoms - the per-structure call: no upload at all.  The binning
    // kernel gets grid and status as kernel arguments and reads the atoms from the pinned staging
    // block (they cross the link once or twice; an upload costs 15 us before the first kernel starts).
    // `records` (the call combiner's batches): no copy engine either - the atoms are 24-byte records in pinned memory, and
    // one kernel reads them and the header (h: status, grids, work list) across the link into the device block; the
    // occlusion kernels write the values straight into the pinned `hout`.  R must be 0 then.
    const bool single = S == 1 && N <= kSingleAtoms && !zero_copy;
    if (zero_copy)
        launch_unpack_atoms(records, (uint32_t)N, (float *)(d + l.o_x), (float *)(d + l.o_y), (float *)(d + l.o_z), (float *)(d + l.o_r),
                            l.has_id ? (uint64_t *)(d + l.o_id) : nullptr, h, d, (uint32_t)l.o_x, st);
    else if (!single) RS_HIP(ctx, hipMemcpyAsync(d, h, l.in_bytes, hipMemcpyHostToDevice, st));
    const char *src = single ? h : d;
    const bool keep_ids = l.has_id && !occlusion_uses_mx(ctx->tuning, lat, (uint32_t)N);

    BatchView v{};
    v.x = (const float *)(src + l.o_x); v.y = (const float *)(src + l.o_y); v.z = (const float *)(src + l.o_z);
    v.radius = (const float *)(src + l.o_r);
    v.id = l.has_id ? (const uint64_t *)(src + l.o_id) : nullptr;
    v.residue_offsets = R ? (const uint32_t *)(src + l.o_res) : nullptr;
    v.n_atoms = (uint32_t)N; v.n_structures = (uint32_t)S; v.n_residues = (uint32_t)R;
    v.probe = l.probe;
    v.grids = (StructGrid *)(d + l.o_grid);
    v.status = (BatchStatus *)d;
    v.sid_sorted = (uint32_t *)ctx->sid_sorted.p;
    v.deferred_list = (uint32_t *)ctx->deferred_list.p;
    v.claim = (uint32_t *)ctx->claim.p;
    v.cell_of = (uint32_t *)ctx->cell_of.p;
    v.rank_of = (uint32_t *)ctx->rank_of.p;
    v.cells = (uint32_t *)ctx->cells.p;
    v.cell_capacity = l.tail_begin + 8;
    v.windows = (uint4 *)(d + l.o_win);
    v.window_capacity = (uint32_t)W;
    v.sorted_xyzr = (float4 *)ctx->sorted_xyzr.p;
    v.sorted_orig = (uint32_t *)ctx->sorted_orig.p;
    v.sorted_id = keep_ids ? (uint64_t *)ctx->sorted_id.p : nullptr;
    v.sorted_id32 = l.has_id ? (uint32_t *)ctx->sorted_id32.p : nullptr;
    // (the single-structure call also gets its results written straight into the pinned block)
    char *outp = single || zero_copy ? hout : dout;
    v.atom_sasa = (float *)(outp + l.o_oa);
    v.residue_sasa = R ? (float *)(outp + l.o_or) : nullptr;
    // (and the general kernel is only launched if the straight-line one says it left atoms to it:
    // a word of the pinned block's header, looked at after the stream has drained)
    uint32_t *flag = reinterpret_cast<uint32_t *>(h + 56);
    *flag = 0u;
    v.defer_flag = single || zero_copy ? flag : nullptr;  // (k_occlusion_fast reports through it; the matrix-core kernel's launch ignores it)
    if (single) launch_sort_single(v, SingleJob{grids[0], stt}, st);
    else launch_sort_lds(v, st);
    launch_occlusion(v, lat, ctx->tuning, kOccAll, st);
    launch_residue_sums(v, st);
    if (!single && !zero_copy) RS_HIP(ctx, hipMemcpyAsync(hout, dout, l.out_bytes, hipMemcpyDeviceToHost, st));
    RS_HIP(ctx, hipGetLastError());
    const auto t_queued = std::chrono::steady_clock::now();
    RS_HIP(ctx, hipStreamSynchronize(st));
    if (zero_copy) {  // (RSASA_COMBINE_TRACE: a merged batch's launches against the wait for them; two relaxed adds otherwise)
        const auto t_done = std::chrono::steady_clock::now();
        g_small_trace_ns[0].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_queued - t_begin).count(), std::memory_order_relaxed);
        g_small_trace_ns[1].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_done - t_queued).count(), std::memory_order_relaxed);
    }
    if ((single || zero_copy) && *flag) {
        launch_occlusion_deferred(v, lat, st);
        launch_residue_sums(v, st);
        RS_HIP(ctx, hipGetLastError());
        RS_HIP(ctx, hipStreamSynchronize(st));
    }
    return RSASA_OK;
}

// Batches of a few structures handed over in host memory - the literal drop-in use, one
// calculate_sasa_internal call per structure - are latency bound: ~17 kernel launches and half a
// dozen small copies.  Here the host computes the bounding boxes and grids itself (N is small),
// so the device needs ONE upload (inputs + grids + status, through pinned staging), four
// launches (LDS binning, the two occlusion kernels, residue sums) and one download.
// kNotSmall: the batch is the general path's (too large, or anything unusual, which that path validates and reports).
int run_small_host_batch(rsasa_context *ctx, const SmallSource &in, const uint32_t *so, size_t S,
                         float probe, size_t n_points, float *out_atom, const uint32_t *ro, size_t R,
                         float *out_res)
{
    if (S == 0 || S > kSmallStructures || so[0] != 0 || n_points == 0 || n_points > (1u << 24) ||
        !(probe >= 0.0f) || !std::isfinite(probe) || ctx->timing || ctx->tuning.debug_stop)
        return kNotSmall;
    const size_t N = so[S];
    if (N == 0 || N > kSmallAtoms) return kNotSmall;
    // (the single-structure call - most calls - keeps its one grid on the stack)
    StructGrid grid1;
    std::vector<StructGrid> grid_vec;
    StructGrid *grids = &grid1;
    if (S > 1) { grid_vec.resize(S); grids = grid_vec.data(); }
    uint4 win1[4];
    std::vector<uint4> win_vec;  // work list of k_sort_window (the general path builds it on the device)
    size_t W = 0;
    unsigned long long total_cells = 0;  // 16-bit entries of the cell array
    for (size_t s = 0; s < S; s++) {
        if (so[s] > so[s + 1]) return kNotSmall;  // (the general path reports it)
        if (!small_structure_grid(in, so[s], so[s + 1], probe, &grids[s])) return kNotSmall;
        grids[s].atom_begin = so[s];
        grids[s].sorted_base = so[s];
        grids[s].cell_base = (uint32_t)total_cells;
        total_cells += lds_cell_slots(grids[s].n_cells);
        for (uint32_t w = 0; w < grid_windows(grids[s].n_cells); w++) {
            const uint4 e = make_uint4((uint32_t)s, w, grids[s].atom_begin, grids[s].n_atoms);
            if (S == 1 && W < 4) win1[W] = e;
            else {
                if (win_vec.empty() && W) win_vec.assign(win1, win1 + W);
                win_vec.push_back(e);
            }
            W++;
        }
    }
    const uint4 *windows = win_vec.empty() ? win1 : win_vec.data();

    Lattice lat;
    int rc = get_lattice(ctx, n_points, &lat);
    if (rc) return rc;
    SmallLayout lay = small_layout(S, N, W, R, in.has_id(), total_cells);
    lay.probe = probe;
    if ((rc = small_reserve(ctx, lay, lat, true))) return rc;
    char *h = (char *)ctx->h_small;
    small_fill(in, 0, (uint32_t)N, lay, h, 0);
    if (R) std::memcpy(h + lay.o_res, ro, (R + 1) * 4);
    char *hout = h + lay.in_bytes;
    if ((rc = small_run(ctx, lay, lat, grids, windows, h, hout, nullptr))) return rc;
    if (out_atom) std::memcpy(out_atom, hout + lay.o_oa, N * 4);
    if (R) std::memcpy(out_res, hout + lay.o_or, R * 4);
    return RSASA_OK;
}

}  // namespace rsasa

extern "C" {


}  // extern "C"

namespace {

// RSASA_H2H_TRACE=1 (under RSASA_TUNING=1): the host-side phases of a pipelined call and, from HIP events, every sub-batch's
// uploads and kernels, on one clock for all contexts of the process (tools/h2h_stream_trace.py).  Measurement only.
struct H2HTrace {
    static bool on()
    {
        static const bool v = tuning_env("RSASA_H2H_TRACE") != nullptr;
        return v;
    }
    static std::chrono::steady_clock::time_point epoch()
    {
        static const auto e = std::chrono::steady_clock::now();
        return e;
    }
    struct Ref {
        std::mutex mu;
        hipEvent_t ev = nullptr;  // ties the device's clock to the host's
        double host_us = 0;
    };
    static Ref &ref()
    {
        static Ref r;
        return r;
    }
    rsasa_context *ctx;
    std::chrono::steady_clock::time_point t0;
    explicit H2HTrace(rsasa_context *c) : ctx(c), t0(std::chrono::steady_clock::now())
    {
        (void)epoch();
        if (!on()) return;
        Ref &r = ref();
        std::lock_guard<std::mutex> lk(r.mu);
        if (!r.ev && hipEventCreate(&r.ev) == hipSuccess) {
            (void)hipEventRecord(r.ev, ctx->stream);
            (void)hipEventSynchronize(r.ev);
            r.host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - epoch()).count();
        }
        for (auto &row : ctx->tr_ev)
            for (hipEvent_t &e : row)
                if (!e) (void)hipEventCreate(&e);
    }
    void rec(int k, int i, hipStream_t s) const
    {
        if (on() && ctx->tr_ev[k][i]) (void)hipEventRecord(ctx->tr_ev[k][i], s);
    }
    void say(const char *what) const
    {
        if (!on()) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "h2h ctx %p at %9.1f us, %8.1f us into the call: %s\n", (void *)ctx,
                     std::chrono::duration<double, std::micro>(now - epoch()).count(),
                     std::chrono::duration<double, std::micro>(now - t0).count(), what);
    }
    void device_side(size_t n_sub) const  // each sub-batch's uploads and kernels on the host trace's clock
    {
        Ref &r = ref();
        if (!on() || !r.ev || n_sub > (size_t)rsasa_context::kSlots) return;
        for (size_t c = 0; c < n_sub; c++) {
            float t[4] = {};
            for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&t[i], r.ev, ctx->tr_ev[c][i]);
            std::fprintf(stderr, "h2h ctx %p device: sub-batch %zu uploads %9.1f .. %9.1f us, kernels %9.1f .. %9.1f us\n", (void *)ctx, c,
                         r.host_us + t[0] * 1e3, r.host_us + t[1] * 1e3, r.host_us + t[2] * 1e3, r.host_us + t[3] * 1e3);
        }
    }
};

// ONE coding pool per device for all its contexts: two contexts with a stream of host batches between them (or
// process_files' pair) would otherwise run two pools of sixteen threads against each other - under a CPU quota (the
// measurement boxes: 16 CPUs) both are throttled, a sub-batch's coding takes 8 ms instead of 1 and its upload waits for
// it.  Jobs are worked off in the order they were submitted, whoever submitted them.
FoldPool *device_fold_pool(rsasa_context *ctx)
{
    static std::mutex pools_mu;
    static FoldPool *pools[64] = {};
    unsigned nt = std::thread::hardware_concurrency() / 4;
    if (const char *v = tuning_env("RSASA_FOLD_THREADS")) nt = (unsigned)std::atoi(v);
    std::lock_guard<std::mutex> lkp(pools_mu);
    const int d = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if (!pools[d]) pools[d] = new (std::nothrow) FoldPool(std::min(16u, std::max(2u, nt)), ctx->node);  // (lives as long as the process)
    return pools[d];
}

bool is_pinned(const void *p)
{
    hipPointerAttribute_t at{};
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// One rsasa_calculate_sasa_batch call on the general path (whatever the small path does not take), in four steps:
//   plan()     the cut into sub-batches of whole structures and residues
//   stage()    device buffers, pinned blocks, what the host prepares per sub-batch (id folds, radius codes) and how results leave
//   run_one()  a batch that is one sub-batch: upload, kernels, wait, copy out
//   run_piped() several sub-batches on three streams: copy-in (c + 1), compute (c), copy-out (c - 1)
// The context is locked and its device current for the object's lifetime.
struct HostBatch {
    static constexpr int kSlots = rsasa_context::kSlots;
    static constexpr size_t kTableWords = 256;  // a sub-batch's offsets block on the device: radius table | residue offsets
    // the call
    rsasa_context *ctx;
    const float *x, *y, *z, *radius;
    const uint64_t *id;
    const uint32_t *structure_offsets;
    size_t n_structures;
    float probe_radius;
    size_t n_points;
    float *out_atom_sasa;
    const uint32_t *residue_offsets;
    size_t n_residues;
    float *out_residue_sasa;
    size_t N = 0;
    bool want_res = false;
    H2HTrace tr;
    // plan
    std::vector<size_t> cut{0};  // structure indices where sub-batches begin / end
    std::vector<size_t> res_cut;
    size_t max_atoms = 1, max_res = 1, n_sub = 1;
    bool piped = false;
    int n_slots = 1;
    // stage
    DeviceBuffer *bx[kSlots], *by[kSlots], *bz[kSlots], *br[kSlots], *bi[kSlots], *bo[kSlots], *oa[kSlots], *orr[kSlots];
    const float *dev_x[kSlots] = {}, *dev_y[kSlots] = {}, *dev_z[kSlots] = {};
    const uint64_t *id_mapped = nullptr;
    bool fold_ids = false, code_radii = false, check_ids = false, host_check = false;
    struct Pack { size_t base = 0, o_res = 0, o_id = 0, o_r8 = 0, bytes = 0; };  // a sub-batch's pinned block: radius table |
    std::vector<Pack> pack;                                                      // residue offsets | radius codes | folded ids
    std::vector<unsigned long long> fold_job;
    bool atoms_direct = true, res_direct = true;
    size_t stage_atoms = 0, stage_bytes = 0;
    std::vector<uint32_t> so[kSlots];  // host copies of the rebased offsets stay alive until their sub-batch has been waited for
    std::vector<char> use_codes;       // sub-batch c's radii travel as codes (decided once its coding job is done)
    std::vector<char> drop_ids;        // sub-batch c's ids stay on the host: those of each of its structures increase strictly, so
                                       // they are all different and change nothing (IdOrder; found by the workers that fold them)
    std::unique_ptr<std::atomic<int>[]> ids_matter;
    struct Staged { bool active = false; size_t a0 = 0, na = 0, r0 = 0, nr = 0; } staged[kSlots];  // results of output slot k that
                                                                                                   // still have to be moved to the caller's arrays
    // No coding job may outlive the call: the workers read the caller's arrays and write the flags above.  (The LAST member:
    // destroyed first, so it waits while everything a job touches is still there - as a local of the old one-function form
    // it was destroyed after the flags, which only an error return with jobs still running could have noticed.)
    struct FoldDrain {
        FoldPool *pool = nullptr;
        unsigned long long last = 0;
        ~FoldDrain() { if (pool) pool->wait(last); }
    } fold_drain;
    HostBatch(rsasa_context *c, const float *x_, const float *y_, const float *z_, const float *r_, const uint64_t *id_, const uint32_t *so_,
              size_t ns, float probe, size_t np, float *oa_, const uint32_t *ro_, size_t nr, float *or_)
        : ctx(c), x(x_), y(y_), z(z_), radius(r_), id(id_), structure_offsets(so_), n_structures(ns), probe_radius(probe), n_points(np),
          out_atom_sasa(oa_), residue_offsets(ro_), n_residues(nr), out_residue_sasa(or_), tr(c)
    {
        N = n_structures ? structure_offsets[n_structures] : 0;
        want_res = residue_offsets && n_residues;
    }

    size_t atoms_of(size_t c) const { return structure_offsets[cut[c + 1]] - structure_offsets[cut[c]]; }
    size_t residues_of(size_t c) const { return want_res ? res_cut[c + 1] - res_cut[c] : 0; }

    // Large batches are cut into sub-batches of whole structures (and whole residues) whose host-to-device copies run on
    // a second stream into a second set of input buffers while the previous sub-batch computes: the PCIe transfer hides
    // behind the kernels.
    void plan()
    {
        size_t kSubAtoms = 1500000;  // smallest sub-batch worth its own launch sequence
        if (const char *v = tuning_env("RSASA_SUB_ATOMS")) kSubAtoms = (size_t)std::max(100000, std::atoi(v));
        if (N >= 2 * kSubAtoms && n_structures > 1) {
            // (a worker of a stream of host batches: the NEXT call's uploads hide a call's fill and drain, so two sub-batches
            // - one upload running beside one half's kernels - are enough, and every sub-batch fewer is a grid build fewer
            // between the occlusion kernels: HostStream)
            size_t max_sub = ctx->stream_sub_batches ? ctx->stream_sub_batches : 8;
            if (const char *v = tuning_env("RSASA_SUB_BATCHES")) max_sub = (size_t)std::max(2, std::atoi(v));
            const size_t want = std::min<size_t>(max_sub, N / kSubAtoms);
            // The link is the longest leg.  The first sub-batch's upload is not hidden behind anything, and nothing hides
            // the last one's kernels: each gets half a share (RSASA_H2H_TAIL=0: only the first).
            static const bool half_tail = !(tuning_env("RSASA_H2H_TAIL") && std::atoi(tuning_env("RSASA_H2H_TAIL")) == 0);
            const size_t first = half_tail ? N / (2 * want - 2) : N / (2 * want - 1);
            const size_t share = half_tail ? 2 * first : (N - first) / (want - 1);
            auto boundary = [&](size_t k) { return first + (k - 1) * share; };  // first atom of sub-batch k >= 1
            size_t next = 1;
            for (size_t sidx = 1; sidx < n_structures && next < want; sidx++) {
                const size_t a0 = structure_offsets[sidx];
                if (a0 < boundary(next)) continue;
                if (want_res && !std::binary_search(residue_offsets, residue_offsets + n_residues + 1, (uint32_t)a0))
                    continue;  // a residue spans this structure boundary: cut later
                cut.push_back(sidx);
                while (next < want && boundary(next) <= a0) next++;
            }
        }
        cut.push_back(n_structures);
        res_cut.assign(cut.size(), 0);
        for (size_t c = 0; c + 1 < cut.size(); c++) {
            max_atoms = std::max<size_t>(max_atoms, atoms_of(c));
            if (want_res) {
                res_cut[c + 1] = c + 2 == cut.size()
                                     ? n_residues
                                     : (size_t)(std::lower_bound(residue_offsets, residue_offsets + n_residues + 1,
                                                                 structure_offsets[cut[c + 1]]) - residue_offsets);
                max_res = std::max(max_res, res_cut[c + 1] - res_cut[c]);
            }
        }
        piped = cut.size() > 2;
        n_sub = cut.size() - 1;
        n_slots = piped ? (int)std::min<size_t>((size_t)kSlots, n_sub) : 1;
    }

    int stage()
    {
        int rc;
        if ((rc = stage_device_buffers())) return rc;
        if ((rc = stage_host_preparation())) return rc;
        if ((rc = stage_outputs())) return rc;
        tr.say("setup done");
        use_codes.assign(cut.size(), 0);
        drop_ids.assign(cut.size(), 0);
        ids_matter.reset(new (std::nothrow) std::atomic<int>[cut.size()]);
        if (!ids_matter) return fail(ctx, RSASA_ERR_OUT_OF_MEMORY, "id flags");
        return RSASA_OK;
    }

    int stage_device_buffers()
    {
        int rc;
        bx[0] = &ctx->in_x; by[0] = &ctx->in_y; bz[0] = &ctx->in_z; br[0] = &ctx->in_r;
        bi[0] = &ctx->in_id; bo[0] = &ctx->in_res; oa[0] = &ctx->atom_sasa; orr[0] = &ctx->out_res;
        for (int k = 1; k < kSlots; k++) {
            rsasa_context::MoreSlot &m = ctx->more[k - 1];
            bx[k] = &m.x; by[k] = &m.y; bz[k] = &m.z; br[k] = &m.r; bi[k] = &m.id; bo[k] = &m.res; oa[k] = &m.atom_sasa; orr[k] = &m.out_res;
        }
        for (int k = 0; k < n_slots; k++) {
            if ((rc = reserve(ctx, *bx[k], max_atoms * 4))) return rc;
            if ((rc = reserve(ctx, *by[k], max_atoms * 4))) return rc;
            if ((rc = reserve(ctx, *bz[k], max_atoms * 4))) return rc;
            dev_x[k] = (const float *)bx[k]->p; dev_y[k] = (const float *)by[k]->p; dev_z[k] = (const float *)bz[k]->p;
            if ((rc = reserve(ctx, *br[k], max_atoms * 4))) return rc;
            if (id && !piped && (rc = reserve(ctx, *bi[k], max_atoms * 8))) return rc;  // (pipelined: stage_host_preparation, unless the ids are folded)
            if (want_res && !piped && (rc = reserve(ctx, *bo[k], (max_res + 1) * 4))) return rc;
            if ((rc = reserve(ctx, *oa[k], max_atoms * 4))) return rc;
            if (want_res && (rc = reserve(ctx, *orr[k], max_res * 4))) return rc;
        }
        return RSASA_OK;
    }

    // What the host prepares for a sub-batch, and the pinned blocks it travels in.
    int stage_host_preparation()
    {
        int rc;
        // Ids on the pipelined path: the link is the longest leg, and the matrix-core kernel only looks at 32-bit folds
        // of the ids.  With the caller's ids in pinned memory the host folds them (a few worker threads, one sub-batch
        // ahead of the uploads) and 4 bytes per atom cross the link instead of 8; the general kernel reads the few full
        // ids it needs (atoms whose folds collide) straight from the caller's array, mapped into the device's
        // address space.  Pageable ids, or a sub-batch the per-atom kernels take: the 64-bit ids are uploaded.
        if (piped && id && !tuning_env("RSASA_NO_ID_FOLD")) {
            Lattice lat_probe;
            void *dp = nullptr;
            if (n_points >= 1 && n_points <= (1u << 24) && get_lattice(ctx, n_points, &lat_probe) == RSASA_OK &&
                hipHostGetDevicePointer(&dp, const_cast<uint64_t *>(id), 0) == hipSuccess && dp) {
                fold_ids = true;
                for (size_t c = 0; c + 1 < cut.size(); c++) fold_ids &= occlusion_uses_mx(ctx->tuning, lat_probe, (uint32_t)atoms_of(c));
                id_mapped = (const uint64_t *)dp;
            } else {
                (void)hipGetLastError();
            }
        }
        // Radii on the pipelined path: one-byte codes into the table of the batch's distinct radii (RadiusCodec), coded by
        // the same worker threads.
        code_radii = piped && !tuning_env("RSASA_NO_RADIUS_CODES");
        // the sub-batches' pinned blocks: radius table | residue offsets | radius codes | folded ids, 16-byte aligned parts
        pack.assign(cut.size(), Pack());
        if (piped) {
            auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
            size_t total = 0, largest = 0;
            for (size_t c = 0; c + 1 < cut.size(); c++) {
                const size_t na = atoms_of(c), nr = residues_of(c);
                Pack &pk = pack[c];
                pk.base = total;
                pk.o_res = kTableWords * 4;
                // (the folded ids last: a sub-batch whose ids turn out not to matter is uploaded without them)
                pk.o_r8 = pk.o_res + up16(want_res ? (nr + 1) * 4 : 0);
                pk.o_id = pk.o_r8 + up16(code_radii ? na : 0);
                pk.bytes = pk.o_id + up16(fold_ids ? na * 4 : 0);
                total += pk.bytes;
                largest = std::max(largest, pk.bytes);
            }
            for (int k = 0; k < n_slots; k++)
                if ((rc = reserve(ctx, ctx->in_pack[k], largest))) return rc;
            if (total > ctx->h_pack_cap) {
                if (ctx->h_pack) {
                    RS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
                    RS_HIP(ctx, hipHostFree(ctx->h_pack));
                    ctx->h_pack = nullptr;
                    ctx->h_pack_cap = 0;
                }
                const size_t cap = total + total / 4;
                RS_HIP(ctx, hipHostMalloc((void **)&ctx->h_pack, cap, hipHostMallocDefault));
                ctx->h_pack_cap = cap;
            }
        }
        // Ids that are all different within their structure change nothing (BatchView::ids_check).  The pipelined path's
        // coding workers look while they fold; one large sub-batch is checked by the same workers while its coordinates
        // cross the link (then its 8 bytes of id per atom stay on the host); anything smaller is checked on the device.
        check_ids = id && !tuning_env("RSASA_NO_ID_CHECK");
        if (!piped && check_ids && cut.size() == 2 && structure_offsets[n_structures] >= 262144u && n_points >= 1 && n_points <= (1u << 24)) {
            Lattice lat_probe;
            host_check = get_lattice(ctx, n_points, &lat_probe) == RSASA_OK &&
                         occlusion_uses_mx(ctx->tuning, lat_probe, structure_offsets[n_structures]);
        }
        if ((fold_ids || code_radii || host_check) && !ctx->fold_pool) {
            ctx->fold_pool = device_fold_pool(ctx);
            if (!ctx->fold_pool) return fail(ctx, RSASA_ERR_OUT_OF_MEMORY, "fold pool");
        }
        fold_job.assign(cut.size(), 0);
        for (int k = 0; k < n_slots && want_res && !piped; k++) {
            rsasa_context::HostSlot &hs = ctx->slot[k];
            if (max_res + 1 <= hs.h_res_cap) continue;
            if (hs.h_res) {
                RS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
                RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
                RS_HIP(ctx, hipHostFree(hs.h_res));
                hs.h_res = nullptr;
                hs.h_res_cap = 0;
            }
            const size_t cap = max_res + 1 + max_res / 4;
            RS_HIP(ctx, hipHostMalloc((void **)&hs.h_res, cap * sizeof(uint32_t), hipHostMallocDefault));
            hs.h_res_cap = cap;
        }
        for (int k = 0; k < n_slots && piped && id && !fold_ids; k++)
            if ((rc = reserve(ctx, *bi[k], max_atoms * 8))) return rc;
        return RSASA_OK;
    }

    // Results leave on their own stream while the next sub-batch computes.  A destination in pinned (page-locked) host
    // memory takes the copy directly; a pageable one gets it through pinned staging, moved to its place by this thread
    // once the copy has landed.
    int stage_outputs()
    {
        atoms_direct = !out_atom_sasa || is_pinned(out_atom_sasa);
        res_direct = !want_res || is_pinned(out_residue_sasa);
        stage_atoms = (out_atom_sasa && !atoms_direct) ? max_atoms * 4 : 0;
        stage_bytes = stage_atoms + ((want_res && !res_direct) ? max_res * 4 : 0);
        for (int k = 0; k < n_slots && stage_bytes; k++) {
            if (stage_bytes <= ctx->h_out_cap[k]) continue;
            if (ctx->h_out[k]) {
                RS_HIP(ctx, hipStreamSynchronize(ctx->d2h_stream));
                RS_HIP(ctx, hipHostFree(ctx->h_out[k]));
                ctx->h_out[k] = nullptr;
                ctx->h_out_cap[k] = 0;
            }
            RS_HIP(ctx, hipHostMalloc(&ctx->h_out[k], stage_bytes + stage_bytes / 4, hipHostMallocDefault));
            ctx->h_out_cap[k] = stage_bytes + stage_bytes / 4;
        }
        return RSASA_OK;
    }

    // the coordinates of a sub-batch do not wait for its coding job (ids, radius codes): they are queued first, and the
    // first sub-batch's start crossing the link while its block is still being written
    int upload_xyz(size_t c, hipStream_t st)
    {
        const int k = (int)(c % kSlots);
        const size_t a0 = structure_offsets[cut[c]], na = atoms_of(c);
        if (na) {
            // (one hipMemcpy2DAsync of three rows for x, y, z a fixed distance apart runs at the link's rate by itself -
            // tools/microbench_copy2d.hip - but is a kernel: behind the occlusion kernels it waits for CUs, and the call
            // took 6.5 instead of 5.1 ms)
            RS_HIP(ctx, hipMemcpyAsync(bx[k]->p, x + a0, na * 4, hipMemcpyHostToDevice, st));
            RS_HIP(ctx, hipMemcpyAsync(by[k]->p, y + a0, na * 4, hipMemcpyHostToDevice, st));
            RS_HIP(ctx, hipMemcpyAsync(bz[k]->p, z + a0, na * 4, hipMemcpyHostToDevice, st));
        }
        return RSASA_OK;
    }

    // everything else of sub-batch c: radii (or their codes), ids (or their folds), rebased offsets
    int upload(size_t c, hipStream_t st)
    {
        const int k = (int)(c % kSlots);
        const size_t s0 = cut[c], s1 = cut[c + 1], a0 = structure_offsets[s0], na = structure_offsets[s1] - a0;
        so[k].resize(s1 - s0 + 1);
        for (size_t i = s0; i <= s1; i++) so[k][i - s0] = structure_offsets[i] - (uint32_t)a0;
        if (na) {
            if (!use_codes[c]) RS_HIP(ctx, hipMemcpyAsync(br[k]->p, radius + a0, na * 4, hipMemcpyHostToDevice, st));
            if (!fold_ids && id && !drop_ids[c]) RS_HIP(ctx, hipMemcpyAsync(bi[k]->p, id + a0, na * 8, hipMemcpyHostToDevice, st));
        }
        if (piped) {
            // the sub-batch's pinned block (the workers have filled in ids and radius codes): table and offsets, one copy
            char *blk = ctx->h_pack + pack[c].base;
            if (use_codes[c]) std::memcpy(blk, ctx->radius_codec.table, kTableWords * 4);
            if (want_res) {
                const size_t r0 = res_cut[c], r1 = res_cut[c + 1];
                uint32_t *ro = reinterpret_cast<uint32_t *>(blk + pack[c].o_res);
                for (size_t i = r0; i <= r1; i++) ro[i - r0] = residue_offsets[i] - (uint32_t)a0;
            }
            RS_HIP(ctx, hipMemcpyAsync(ctx->in_pack[k].p, blk, drop_ids[c] ? pack[c].o_id : pack[c].bytes, hipMemcpyHostToDevice, st));
        } else if (want_res) {
            const size_t r0 = res_cut[c], r1 = res_cut[c + 1];
            uint32_t *ro = ctx->slot[k].h_res;
            for (size_t i = r0; i <= r1; i++) ro[i - r0] = residue_offsets[i] - (uint32_t)a0;
            RS_HIP(ctx, hipMemcpyAsync(bo[k]->p, ro, (r1 - r0 + 1) * 4, hipMemcpyHostToDevice, st));
        }
        return RSASA_OK;
    }

    int drain(int k)  // output slot k's staged results to the caller's (pageable) arrays
    {
        if (!staged[k].active) return RSASA_OK;
        RS_HIP(ctx, hipEventSynchronize(ctx->ev_d2h[k]));
        const char *h = (const char *)ctx->h_out[k];
        if (out_atom_sasa && !atoms_direct && staged[k].na)
            std::memcpy(out_atom_sasa + staged[k].a0, h, staged[k].na * 4);
        if (want_res && !res_direct && staged[k].nr)
            std::memcpy(out_residue_sasa + staged[k].r0, h + stage_atoms, staged[k].nr * 4);
        staged[k].active = false;
        return RSASA_OK;
    }

    int copy_out(size_t c)  // sub-batch c's results (all its kernels have been waited for or ordered before)
    {
        hipStream_t dn = ctx->d2h_stream;
        const int k = (int)(c % kSlots);
        const size_t a0 = structure_offsets[cut[c]], na = atoms_of(c);
        const size_t r0 = res_cut[c], nr = residues_of(c);
        char *h = (char *)ctx->h_out[k];
        if (out_atom_sasa && na)
            RS_HIP(ctx, hipMemcpyAsync(atoms_direct ? (void *)(out_atom_sasa + a0) : (void *)h, oa[k]->p, na * 4,
                                       hipMemcpyDeviceToHost, dn));
        if (nr)
            RS_HIP(ctx, hipMemcpyAsync(res_direct ? (void *)(out_residue_sasa + r0) : (void *)(h + stage_atoms),
                                       orr[k]->p, nr * 4, hipMemcpyDeviceToHost, dn));
        RS_HIP(ctx, hipEventRecord(ctx->ev_d2h[k], dn));
        staged[k].active = stage_bytes != 0;
        staged[k].a0 = a0; staged[k].na = na; staged[k].r0 = r0; staged[k].nr = nr;
        return RSASA_OK;
    }

    // one sub-batch: upload, kernels, wait (rsasa_batch_wait re-runs with a larger cell array if needed), copy out
    int run_one()
    {
        int rc;
        hipStream_t st = ctx->stream;
        unsigned long long order_job = 0;
        if (host_check) {
            IdOrder order;
            ids_matter[0].store(0);
            order.starts = structure_offsets;
            order.n_starts = n_structures;
            order.ids_matter = &ids_matter[0];
            order_job = ctx->fold_pool->submit(id, nullptr, structure_offsets[n_structures], nullptr, nullptr, nullptr, order);
            fold_drain.pool = ctx->fold_pool;
            fold_drain.last = order_job;
        }
        if ((rc = upload_xyz(0, st))) return rc;
        if (host_check) {
            ctx->fold_pool->wait(order_job);
            drop_ids[0] = !ids_matter[0].load();
            if (drop_ids[0]) ctx->ids_dropped.fetch_add(1, std::memory_order_relaxed);
        }
        if ((rc = upload(0, st))) return rc;
        const size_t na = atoms_of(0), nr = residues_of(0);
        if (na || nr) {
            rsasa_device_batch_t bt{};
            bt.x = dev_x[0];
            bt.y = dev_y[0];
            bt.z = dev_z[0];
            bt.radius = (const float *)br[0]->p;
            bt.id = id && !drop_ids[0] ? (const uint64_t *)bi[0]->p : nullptr;
            bt.structure_offsets_host = so[0].data();
            bt.n_structures = n_structures;
            bt.n_atoms = na;
            bt.residue_offsets = nr ? (const uint32_t *)bo[0]->p : nullptr;
            bt.n_residues = nr;
            bt.out_atom_sasa = (float *)oa[0]->p;
            bt.out_residue_sasa = nr ? (float *)orr[0]->p : nullptr;
            bt.out_neighbor_counts = nullptr;
            // (ids that do not rise are not yet ids that matter: the device's own check - its hash tables for ids in no order -
            // still runs on them; the host's pass only ever proves the droppable case)
            if ((rc = batch_enqueue(ctx, &bt, probe_radius, n_points, nullptr, false))) return rc;
            if ((rc = rsasa_batch_wait(ctx))) return rc;
        }
        if ((rc = copy_out(0))) return rc;
        if ((rc = drain(0))) return rc;
        RS_HIP(ctx, hipStreamSynchronize(ctx->d2h_stream));
        return RSASA_OK;
    }

    // ---- several sub-batches ----
    struct Attempt {  // what one pass over the sub-batches finds out
        uint64_t need_cells = 0;
        int err = RSASA_OK;
        bool used[kSlots] = {};
    };

    void check(int k, Attempt &at)  // status of the sub-batch that used host slot k (its event has been waited for)
    {
        const BatchStatus stt = *ctx->slot[k].h_status;
        if (stt.grid_too_large && !at.err)
            at.err = fail(ctx, RSASA_ERR_GRID_TOO_LARGE, "a structure's cell grid exceeds 2^31 cells (coordinates too sparse)");
        if (stt.bad_input && !at.err)
            at.err = fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "probe_radius + max radius must be a positive finite number");
        if (stt.overflow) at.need_cells = std::max<uint64_t>(at.need_cells, stt.total_cells);
        else ctx->tuning.deferred_hint = stt.deferred;
        if (!stt.overflow && ctx->slot[k].ids_check) {
            ctx->ids_drop_hint = !stt.ids_needed;
            ctx->ids_unordered_hint = (stt.ids_unordered & 1u) != 0u;
            ctx->ids_kept_structures.store(stt.ids_needed, std::memory_order_relaxed);
            if (!stt.ids_needed) ctx->ids_dropped.fetch_add(1, std::memory_order_relaxed);
        }
    }

    // all sub-batches' folds and radius codes, in order, while the uploads follow behind (the previous attempt's copies
    // out of the pinned blocks have all been waited for)
    void submit_coding_jobs()
    {
        if (!(fold_ids || code_radii)) return;
        if (code_radii) ctx->radius_codec.reset();
        for (size_t c = 0; c < n_sub; c++) {
            const size_t a0 = structure_offsets[cut[c]], na = atoms_of(c);
            char *blk = ctx->h_pack + pack[c].base;
            IdOrder order;
            ids_matter[c].store(0);
            if (fold_ids && check_ids) {
                order.starts = structure_offsets + cut[c];
                order.n_starts = cut[c + 1] - cut[c];
                order.first = (uint32_t)a0;
                order.ids_matter = &ids_matter[c];
            }
            fold_job[c] = ctx->fold_pool->submit(fold_ids ? id + a0 : nullptr,
                                                 fold_ids ? reinterpret_cast<uint32_t *>(blk + pack[c].o_id) : nullptr, na,
                                                 radius + a0, code_radii ? reinterpret_cast<uint8_t *>(blk + pack[c].o_r8) : nullptr,
                                                 code_radii ? &ctx->radius_codec : nullptr, order);
        }
        fold_drain.pool = ctx->fold_pool;
        fold_drain.last = fold_job[n_sub - 1];
    }

    // sub-batch c: its uploads on the copy stream, its kernels behind them on one of the two launch streams, its results
    // out behind those
    int launch_sub_batch(size_t c, Attempt &at)
    {
        int rc;
        hipStream_t cp = ctx->copy_stream, dn = ctx->d2h_stream;
        const int k = (int)(c % kSlots);
        if (at.used[k]) {
            // slot k (host segments / status, input and output buffers) was sub-batch c - kSlots's
            RS_HIP(ctx, hipEventSynchronize(ctx->ev_done[k]));
            check(k, at);
            if ((rc = drain(k))) return rc;  // its staged results, if the destination is pageable
        }
        tr.rec(k, 0, cp);
        if ((rc = upload_xyz(c, cp))) return rc;
        if (fold_ids || code_radii) ctx->fold_pool->wait(fold_job[c]);
        use_codes[c] = code_radii && !ctx->radius_codec.failed.load();
        drop_ids[c] = fold_ids && check_ids && !ids_matter[c].load();
        if (drop_ids[c]) ctx->ids_dropped.fetch_add(1, std::memory_order_relaxed);
        if (c == 0) tr.say("first sub-batch coded");
        if ((rc = upload(c, cp))) return rc;
        tr.rec(k, 1, cp);
        RS_HIP(ctx, hipEventRecord(ctx->ev_copy[k], cp));
        // consecutive sub-batches alternate between the context's two workspaces and launch streams: a
        // sub-batch's grid build is then queued beside its predecessor's occlusion kernel and starts in its tail
        // (enqueue_batch chains the occlusion kernels themselves)
        const int w = (int)(c & 1);
        hipStream_t st = w ? ctx->stream2 : ctx->stream;
        RS_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_copy[k], 0));
        if (at.used[k]) RS_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_d2h[k], 0));  // output slot k has left the device
        const size_t s0 = cut[c], s1 = cut[c + 1], na = atoms_of(c), nr = residues_of(c);
        Pending pd;
        pd.batch.x = dev_x[k];
        pd.batch.y = dev_y[k];
        pd.batch.z = dev_z[k];
        pd.batch.radius = (const float *)br[k]->p;
        pd.batch.id = drop_ids[c] ? nullptr : fold_ids ? id_mapped + structure_offsets[s0] : id ? (const uint64_t *)bi[k]->p : nullptr;
        const char *dblk = (const char *)ctx->in_pack[k].p;
        pd.id32 = fold_ids && !drop_ids[c] ? (const uint32_t *)(dblk + pack[c].o_id) : nullptr;
        // (ids that do not rise are not yet ids that matter: the device looks at the folds - a hash table per structure once
        // the context has seen ids in no order - and every structure whose folds all differ runs without its ids)
        pd.ids_needed_known = false;
        pd.batch.structure_offsets_host = so[k].data();
        pd.batch.n_structures = s1 - s0;
        pd.batch.n_atoms = na;
        pd.batch.residue_offsets = nr ? (const uint32_t *)(dblk + pack[c].o_res) : nullptr;
        pd.radius8 = use_codes[c] ? (const uint8_t *)(dblk + pack[c].o_r8) : nullptr;
        pd.radius_table = use_codes[c] ? (const float *)dblk : nullptr;
        pd.batch.n_residues = nr;
        pd.batch.out_atom_sasa = (float *)oa[k]->p;
        pd.batch.out_residue_sasa = nr ? (float *)orr[k]->p : nullptr;
        pd.batch.out_neighbor_counts = nullptr;
        pd.probe = probe_radius;
        pd.n_points = n_points;
        pd.stream = st;
        pd.ws = w;
        tr.rec(k, 2, st);
        if ((rc = enqueue_batch(ctx, pd, ctx->slot[k]))) return rc;
        tr.rec(k, 3, st);
        RS_HIP(ctx, hipEventRecord(ctx->ev_done[k], st));
        RS_HIP(ctx, hipStreamWaitEvent(dn, ctx->ev_done[k], 0));
        if ((rc = copy_out(c))) return rc;
        at.used[k] = true;
        if (H2HTrace::on()) tr.say(c + 1 == n_sub ? "last sub-batch enqueued" : "sub-batch enqueued");
        return RSASA_OK;
    }

    // Several sub-batches on three streams: copy-in (sub-batch c + 1), compute (c), copy-out (c - 1).  kSlots
    // sub-batches are in flight: the host queues the next one (upload, then kernels behind the upload's event) while
    // earlier ones compute and never waits in between - the uploads, which are the longest leg (PCIe), follow each other
    // without a gap.  A sub-batch's status block (host slot c % kSlots) is only read when its slot is needed again or at
    // the end; if one of them reports that the cell array was too small, everything is drained, the array grows to the
    // largest size reported and the call starts over (outputs are simply written again) - that happens on a context's
    // first large call at most.
    int run_piped()
    {
        int rc;
        if (n_points == 0 || n_points > (1u << 24))
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "n_points must be in [1, 2^24]");
        if (!(probe_radius >= 0.0f) || !std::isfinite(probe_radius))
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "probe_radius must be finite and >= 0");
        for (size_t sidx = 0; sidx < n_structures; sidx++)
            if (structure_offsets[sidx] > structure_offsets[sidx + 1])
                return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "structure_offsets must be non-decreasing");
        if (structure_offsets[0] != 0) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "structure_offsets must span [0, n_atoms]");
        hipStream_t cp = ctx->copy_stream;
        if (!ctx->stream2) RS_HIP(ctx, new_stream(ctx, &ctx->stream2, 2));
        for (int attempt = 0;; attempt++) {
            Attempt at;
            LinkHold turn;  // (released without an event on an error return)
            submit_coding_jobs();
            // (the coding jobs run while this call waits for its turn on the link)
            RS_HIP(ctx, turn.take(ctx, cp));
            tr.say("turn on the link taken");
            for (size_t c = 0; c < n_sub; c++)
                if ((rc = launch_sub_batch(c, at))) return rc;
            turn.pass(cp);  // the next call's uploads follow this one's last
            tr.say("turn passed on");
            for (int k = 0; k < kSlots; k++) {
                if (!at.used[k]) continue;
                RS_HIP(ctx, hipEventSynchronize(ctx->ev_done[k]));
                check(k, at);
                if ((rc = drain(k))) return rc;
            }
            RS_HIP(ctx, hipStreamSynchronize(ctx->d2h_stream));
            tr.say("all done");
            tr.device_side(n_sub);
            if (at.err) return at.err;
            if (!at.need_cells) return RSASA_OK;
            if (at.need_cells >= 0xFFFFFFF0ull || attempt >= 3)
                return fail(ctx, RSASA_ERR_GRID_TOO_LARGE, "batch needs more than 2^32 grid cells; split it");
            ctx->cell_capacity = at.need_cells + at.need_cells / 8 + 1024;
        }
    }
};

}  // namespace

extern "C" {

int rsasa_calculate_sasa_batch(rsasa_context_t *ctx, const float *x, const float *y,
                               const float *z, const float *radius, const uint64_t *id,
                               const uint32_t *structure_offsets, size_t n_structures,
                               float probe_radius, size_t n_points, float *out_atom_sasa,
                               const uint32_t *residue_offsets, size_t n_residues,
                               float *out_residue_sasa)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (n_structures && !structure_offsets)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "structure_offsets is NULL");
    const size_t N = n_structures ? structure_offsets[n_structures] : 0;
    const bool want_res = residue_offsets && n_residues;
    if (N && (!x || !y || !z || !radius))
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "coordinate / radius arrays are NULL");
    if (want_res && !out_residue_sasa)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "out_residue_sasa is NULL");
    if (N && !out_atom_sasa && !want_res)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "no output requested");
    if (N == 0 && !want_res) return RSASA_OK;  // empty input -> empty output (tests/sanity.rs:149-157)
    if (want_res) {
        if (residue_offsets[n_residues] > N)
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "residue_offsets exceed n_atoms");
        uint32_t decreasing = 0;  // (branch-free: vectorised; a million and a half offsets per proteome batch)
        for (size_t k = 0; k < n_residues; k++) decreasing |= (uint32_t)(residue_offsets[k] > residue_offsets[k + 1]);
        if (decreasing) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "residue_offsets must be non-decreasing");
    }

    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    RS_DEVICE(ctx);
    if (ctx->n_pending && (rc = wait_pending(ctx))) return rc;
    if (ctx->small_path) {
        SmallSource in;
        in.x = x; in.y = y; in.z = z; in.radius = radius; in.id = id;
        rc = run_small_host_batch(ctx, in, structure_offsets, n_structures, probe_radius, n_points,
                                  out_atom_sasa, want_res ? residue_offsets : nullptr, want_res ? n_residues : 0,
                                  out_residue_sasa);
        if (rc != kNotSmall) return rc;
    }
    if ((rc = ensure_copy_streams(ctx))) return rc;
    HostBatch call(ctx, x, y, z, radius, id, structure_offsets, n_structures, probe_radius, n_points, out_atom_sasa, residue_offsets,
                   n_residues, out_residue_sasa);
    call.plan();
    if ((rc = call.stage())) return rc;
    return call.piped ? call.run_piped() : call.run_one();
}

// ---- a stream of host batches (ABI 3) ----

static void host_stream_worker(HostStream *hs, int w)
{
    (void)rsasa_context_bind_thread(hs->sub[w], nullptr);
    for (;;) {
        std::shared_ptr<HostStream::Job> job;
        {
            std::unique_lock<std::mutex> lk(hs->mu);
            hs->cv_work.wait(lk, [&] {
                if (hs->quit) return true;
                for (auto &j : hs->jobs) if (!j->taken) return true;
                return false;
            });
            for (auto &j : hs->jobs)
                if (!j->taken) { job = j; break; }  // oldest first
            if (!job) return;                        // quit and nothing left to take
            job->taken = true;
        }
        {
            std::lock_guard<std::recursive_mutex> lk(hs->sub[w]->mu);
            hs->sub[w]->simd_width = job->simd_width;
            hs->sub[w]->small_path = job->small_path;
            hs->sub[w]->overlap_tail = job->overlap_tail;
            const uint32_t hint = hs->sub[w]->tuning.deferred_hint;  // (what this context has learnt stays its own)
            hs->sub[w]->tuning = job->tuning;
            hs->sub[w]->tuning.deferred_hint = hint;
            hs->sub[w]->stream_sub_batches = hs->sub_batches;
            hs->sub[w]->link_gate = &hs->gate;
            hs->sub[w]->link_ticket = job->ticket;
        }
        const int rc = rsasa_calculate_sasa_batch(hs->sub[w], job->x, job->y, job->z, job->radius, job->id, job->structure_offsets,
                                                  job->n_structures, job->probe, job->n_points, job->out_atom,
                                                  job->residue_offsets, job->n_residues, job->out_res);
        hs->gate.advance(job->ticket);  // (a call that never took the link: a small batch, an error)
        std::string msg = rc ? rsasa_context_last_error(hs->sub[w]) : "";
        {
            std::lock_guard<std::mutex> lk(hs->mu);
            job->rc = rc;
            job->error = std::move(msg);
            job->done = true;
        }
        hs->cv_done.notify_all();
    }
}

int rsasa_host_batch_enqueue(rsasa_context_t *ctx, const float *x, const float *y, const float *z,
                             const float *radius, const uint64_t *id, const uint32_t *structure_offsets,
                             size_t n_structures, float probe_radius, size_t n_points, float *out_atom_sasa,
                             const uint32_t *residue_offsets, size_t n_residues, float *out_residue_sasa)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    HostStream *hs = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        if (!ctx->host_stream) {
            hs = new (std::nothrow) HostStream();
            if (!hs) return fail(ctx, RSASA_ERR_OUT_OF_MEMORY, "host stream");
            if (const char *v = tuning_env("RSASA_STREAM_WORKERS")) hs->n_workers = std::min((int)HostStream::kMaxWorkers, std::max(1, std::atoi(v)));
            for (int w = 0; w < hs->n_workers; w++) {
                int own = 2;
                if (const char *v = tuning_env("RSASA_OWN_QUEUES")) own = std::atoi(v);
                rc = context_create(ctx->device, own, &hs->sub[w]);
                if (rc) {
                    for (rsasa_context *sc : hs->sub) rsasa_context_destroy(sc);
                    delete hs;
                    return fail(ctx, rc, "rsasa_context_create (host stream worker)");
                }
            }
            {
                hs->sub_batches = 2;
            }
            for (int w = 0; w < hs->n_workers; w++) hs->th[w] = std::thread(host_stream_worker, hs, w);
            ctx->host_stream = hs;
        }
        hs = ctx->host_stream;
    }
    auto job = std::make_shared<HostStream::Job>();
    {
        // the workers compute with the caller's settings as they are now (lane count, kernel choice); the worker that
        // takes the job applies them (its context is locked for the length of the call it is in)
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        job->simd_width = ctx->simd_width; job->small_path = ctx->small_path; job->overlap_tail = ctx->overlap_tail;
        job->tuning = ctx->tuning;
    }
    job->x = x; job->y = y; job->z = z; job->radius = radius; job->id = id;
    job->structure_offsets = structure_offsets; job->n_structures = n_structures;
    job->probe = probe_radius; job->n_points = n_points; job->out_atom = out_atom_sasa;
    job->residue_offsets = residue_offsets; job->n_residues = n_residues; job->out_res = out_residue_sasa;
    {
        std::unique_lock<std::mutex> lk(hs->mu);
        // a full queue fails at once: completed batches stay queued until rsasa_host_batch_wait returns them, so waiting
        // here for the oldest to be computed could never make room
        if (hs->jobs.size() >= HostStream::kMaxQueued) {
            lk.unlock();
            return fail(ctx, RSASA_ERR_QUEUE_FULL, "eight host batches are enqueued and not waited for: call rsasa_host_batch_wait first");
        }
        job->ticket = hs->next_ticket++;
        hs->jobs.push_back(job);
    }
    hs->cv_work.notify_all();
    return RSASA_OK;
}

// rsasa_host_batch_wait; *took: a batch was there for this thread to wait for
static int host_batch_wait(rsasa_context_t *ctx, bool *took)
{
    *took = false;
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    HostStream *hs = nullptr;
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        hs = ctx->host_stream;
    }
    if (!hs) return RSASA_OK;  // nothing was ever enqueued
    std::shared_ptr<HostStream::Job> job;
    {
        std::unique_lock<std::mutex> lk(hs->mu);
        // the oldest batch that no other thread is waiting for already (two threads of the caller may wait at once: each
        // takes a batch of its own - both taking the front one and both removing "it" removed a batch nobody had waited
        // for, or one from an empty queue: found by tests/test_gpu_concurrency.py test_host_stream_under_stress)
        for (auto &j : hs->jobs)
            if (!j->waited_for) { job = j; break; }
        if (!job) return RSASA_OK;
        job->waited_for = true;
        *took = true;
        hs->cv_done.wait(lk, [&] { return job->done; });
        for (auto it = hs->jobs.begin(); it != hs->jobs.end(); ++it)
            if (*it == job) { hs->jobs.erase(it); break; }
    }
    hs->cv_done.notify_all();  // (an enqueue may be waiting for room)
    if (job->rc) return fail(ctx, job->rc, job->error.c_str());
    return RSASA_OK;
}

int rsasa_host_batch_wait(rsasa_context_t *ctx)
{
    bool took;
    return host_batch_wait(ctx, &took);
}

int rsasa_host_batch_wait_all(rsasa_context_t *ctx)
{
    int first = RSASA_OK;
    for (;;) {
        bool took;
        const int rc = host_batch_wait(ctx, &took);
        if (rc && !first) first = rc;
        if (!took) return first;  // (what is still queued, another thread of the caller is waiting for)
    }
}

// The per-structure calls.  With call combining on (rsasa_context_set_call_combining) concurrent calls of several host
// threads are merged into one batch launch (combine.cpp); otherwise - and for whatever the combiner does not take - a call
// runs by itself: the small path (one structure: no upload, four launches), else the general path.
static int per_structure_call(rsasa_context_t *ctx, const SmallSource &in, size_t n_atoms, float probe_radius, size_t n_points,
                              float *out_sasa)
{
    if (n_atoms >= 0xFFFFFFF0ull) return RSASA_ERR_INVALID_ARGUMENT;
    if (n_atoms && !out_sasa) return RSASA_ERR_INVALID_ARGUMENT;
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (n_atoms && !in.aos && (!in.x || !in.y || !in.z || !in.radius)) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "coordinate / radius arrays are NULL");
    if (n_atoms == 0) return RSASA_OK;  // empty input -> empty output (tests/sanity.rs:149-157)
    if (ctx->combine_wait_us.load(std::memory_order_relaxed) >= 0) {
        rc = combine_call(ctx, in, n_atoms, probe_radius, n_points, out_sasa);
        if (rc != kNotCombined) return rc;
    }
    const uint32_t offsets[2] = {0u, (uint32_t)n_atoms};
    if (in.aos) {
        {
            std::lock_guard<std::recursive_mutex> lk(ctx->mu);
            RS_DEVICE(ctx);
            if (ctx->n_pending && (rc = wait_pending(ctx))) return rc;
            if (ctx->small_path) {
                rc = run_small_host_batch(ctx, in, offsets, 1, probe_radius, n_points, out_sasa, nullptr, 0, nullptr);
                if (rc != kNotSmall) return rc;
            }
        }
        // (a structure the small path does not take - tens of thousands of atoms, or anything unusual: columns for the general path)
        std::vector<float> soa;
        std::vector<uint64_t> ids;
        try {
            soa.resize(4 * n_atoms);
            ids.resize(n_atoms);
        } catch (const std::bad_alloc &) {
            return fail(ctx, RSASA_ERR_OUT_OF_MEMORY, "columns of a per-structure call");
        }
        float *x = soa.data(), *y = x + n_atoms, *z = y + n_atoms, *r = z + n_atoms;
        for (size_t i = 0; i < n_atoms; i++) {
            x[i] = in.aos[i].position[0];
            y[i] = in.aos[i].position[1];
            z[i] = in.aos[i].position[2];
            r[i] = in.aos[i].radius;
            ids[i] = in.aos[i].id;
        }
        return rsasa_calculate_sasa_batch(ctx, x, y, z, r, ids.data(), offsets, 1, probe_radius, n_points, out_sasa, nullptr, 0, nullptr);
    }
    return rsasa_calculate_sasa_batch(ctx, in.x, in.y, in.z, in.radius, in.id, offsets, 1, probe_radius, n_points,
                                      out_sasa, nullptr, 0, nullptr);
}

int rsasa_calculate_sasa_soa(rsasa_context_t *ctx, const float *x, const float *y,
                             const float *z, const float *radius, const uint64_t *id,
                             size_t n_atoms, float probe_radius, size_t n_points,
                             float *out_sasa)
{
    SmallSource in;
    in.x = x; in.y = y; in.z = z; in.radius = radius; in.id = id;
    return per_structure_call(ctx, in, n_atoms, probe_radius, n_points, out_sasa);
}

int rsasa_calculate_sasa_internal(rsasa_context_t *ctx, const rsasa_atom_t *atoms,
                                  size_t n_atoms, float probe_radius, size_t n_points,
                                  ptrdiff_t threads, float *out_sasa)
{
    (void)threads;  // sequential-vs-rayon switch in the reference (src/lib.rs:278); no meaning here
    if (n_atoms && (!atoms || !out_sasa)) return RSASA_ERR_INVALID_ARGUMENT;
    SmallSource in;
    in.aos = atoms;
    return per_structure_call(ctx, in, n_atoms, probe_radius, n_points, out_sasa);
}

int rsasa_calculate_sasa_trajectory(rsasa_context_t *ctx, const float *xyz, size_t n_frames,
                                    size_t n_atoms, const float *radius, const uint64_t *id,
                                    float probe_radius, size_t n_points, float *out_atom_sasa,
                                    const uint32_t *residue_offsets, size_t n_residues,
                                    float *out_residue_sasa)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    const bool want_res = residue_offsets && n_residues;
    if (n_frames == 0 || n_atoms == 0) return RSASA_OK;
    if (!xyz || !radius) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "xyz / radius are NULL");
    if (want_res && !out_residue_sasa) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "out_residue_sasa is NULL");
    if (!out_atom_sasa && !want_res) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "no output requested");
    if (n_atoms >= 0xFFFFFFF0ull || n_frames >= 0xFFFFFFF0ull || n_residues >= 0xFFFFFFF0ull)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "trajectory too large for 32-bit indices");
    if (want_res) {
        if (residue_offsets[n_residues] > n_atoms)
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "residue_offsets exceed n_atoms");
        for (size_t k = 0; k < n_residues; k++)
            if (residue_offsets[k] > residue_offsets[k + 1])
                return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "residue_offsets must be non-decreasing");
    }
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    RS_DEVICE(ctx);
    if (ctx->n_pending && (rc = wait_pending(ctx))) return rc;
    hipStream_t st = ctx->stream;
    // topology columns once
    if ((rc = reserve(ctx, ctx->tr_r, n_atoms * 4))) return rc;
    if (id && (rc = reserve(ctx, ctx->tr_id, n_atoms * 8))) return rc;
    if (want_res && (rc = reserve(ctx, ctx->tr_res, (n_residues + 1) * 4))) return rc;
    RS_HIP(ctx, hipMemcpyAsync(ctx->tr_r.p, radius, n_atoms * 4, hipMemcpyHostToDevice, st));
    if (id) RS_HIP(ctx, hipMemcpyAsync(ctx->tr_id.p, id, n_atoms * 8, hipMemcpyHostToDevice, st));
    if (want_res)
        RS_HIP(ctx, hipMemcpyAsync(ctx->tr_res.p, residue_offsets, (n_residues + 1) * 4,
                                   hipMemcpyHostToDevice, st));
    // Offsets that cover the atoms exactly tile over the frames as they are; otherwise every frame
    // gets one more entry (k_expand_frames) and one gap "residue" whose sum is not copied out.
    const bool res_exact = want_res && residue_offsets[0] == 0 && residue_offsets[n_residues] == n_atoms;
    const size_t res_stride = want_res ? (res_exact ? n_residues : n_residues + 1) : 0;
    // frames in chunks of at most ~32 M atoms (32-bit indices, bounded workspace)
    const size_t chunk_frames = std::max<size_t>(1, std::min<size_t>(n_frames, (32u << 20) / n_atoms));
    std::vector<uint32_t> s_off(chunk_frames + 1);
    for (size_t f0 = 0; f0 < n_frames; f0 += chunk_frames) {
        const size_t nf = std::min(chunk_frames, n_frames - f0);
        const size_t N = nf * n_atoms, R = nf * res_stride;
        if (R >= 0xFFFFFFF0ull) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "trajectory too large for 32-bit indices");
        if ((rc = reserve(ctx, ctx->tr_xyz, N * 12))) return rc;
        if ((rc = reserve(ctx, ctx->in_x, N * 4))) return rc;
        if ((rc = reserve(ctx, ctx->in_y, N * 4))) return rc;
        if ((rc = reserve(ctx, ctx->in_z, N * 4))) return rc;
        if ((rc = reserve(ctx, ctx->in_r, N * 4))) return rc;
        if (id && (rc = reserve(ctx, ctx->in_id, N * 8))) return rc;
        if ((rc = reserve(ctx, ctx->atom_sasa, N * 4))) return rc;
        if (want_res) {
            if ((rc = reserve(ctx, ctx->in_res, (R + 1) * 4))) return rc;
            if ((rc = reserve(ctx, ctx->out_res, R * 4))) return rc;
        }
        RS_HIP(ctx, hipMemcpyAsync(ctx->tr_xyz.p, xyz + f0 * n_atoms * 3, N * 12, hipMemcpyHostToDevice, st));
        launch_expand_frames((const float *)ctx->tr_xyz.p, (const float *)ctx->tr_r.p,
                             id ? (const uint64_t *)ctx->tr_id.p : nullptr,
                             want_res ? (const uint32_t *)ctx->tr_res.p : nullptr, (uint32_t)n_atoms,
                             (uint32_t)nf, (uint32_t)res_stride, (float *)ctx->in_x.p, (float *)ctx->in_y.p,
                             (float *)ctx->in_z.p, (float *)ctx->in_r.p, (uint64_t *)ctx->in_id.p,
                             (uint32_t *)ctx->in_res.p, st);
        for (size_t f = 0; f <= nf; f++) s_off[f] = (uint32_t)(f * n_atoms);
        rsasa_device_batch_t bt{};
        bt.x = (const float *)ctx->in_x.p;
        bt.y = (const float *)ctx->in_y.p;
        bt.z = (const float *)ctx->in_z.p;
        bt.radius = (const float *)ctx->in_r.p;
        bt.id = id ? (const uint64_t *)ctx->in_id.p : nullptr;
        bt.structure_offsets_host = s_off.data();
        bt.n_structures = nf;
        bt.n_atoms = N;
        bt.residue_offsets = want_res ? (const uint32_t *)ctx->in_res.p : nullptr;
        bt.n_residues = R;
        bt.out_atom_sasa = (float *)ctx->atom_sasa.p;
        bt.out_residue_sasa = want_res ? (float *)ctx->out_res.p : nullptr;
        if ((rc = rsasa_batch_enqueue(ctx, &bt, probe_radius, n_points, nullptr))) return rc;
        if ((rc = rsasa_batch_wait(ctx))) return rc;
        if (out_atom_sasa)
            RS_HIP(ctx, hipMemcpy(out_atom_sasa + f0 * n_atoms, ctx->atom_sasa.p, N * 4, hipMemcpyDeviceToHost));
        if (want_res && res_exact)
            RS_HIP(ctx, hipMemcpy(out_residue_sasa + f0 * n_residues, ctx->out_res.p, R * 4,
                                  hipMemcpyDeviceToHost));
        else if (want_res)  // n_residues of every res_stride sums: the gap entries stay behind
            RS_HIP(ctx, hipMemcpy2D(out_residue_sasa + f0 * n_residues, n_residues * 4, ctx->out_res.p,
                                    res_stride * 4, n_residues * 4, nf, hipMemcpyDeviceToHost));
    }
    return RSASA_OK;
}

}  // extern "C"
