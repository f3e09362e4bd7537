// Call combining (rsasa_context_set_call_combining, ABI 4): concurrent per-structure calls become ONE batch launch.
//
// The reference calls its hot path once per file from every rayon worker (src/main.rs:375 `files.par_iter()`, :439 inner
// threads = 1, src/lib.rs:249-254).  Dropped in behind that call pattern unchanged, every call is an upload, four launches
// and a download of its own: 41 us for 2 k atoms, 114 k structures/s from sixteen threads - a tenth of what the engine
// does with a batch.  A batch IS independent structures (own bounding box, grid and largest radius each: the values do not
// depend on a structure's batch-mates, tests/test_gpu_parity.py test_batch_neighbours_do_not_interact), so calls that
// arrive together can share one:
//
//   * a caller scans its atoms (bounding box, largest radius -> its grid: small_structure_grid, what the small path does
//     anyway) and goes to its DEVICE's combiner - contexts of their own or one shared context, either works;
//   * there it joins the OPEN BLOCK of its settings (probe, point count, lane count W, kernel choice, ids or none - calls
//     with different settings never share a block), or opens one: a block is a pinned staging buffer laid out for the
//     largest batch (columns x, y, z, r, id and the values, each at a fixed offset), and joining is taking the next
//     range of atoms in it.  The caller copies ITS atoms into that range at once (AoS records are de-interleaved there:
//     no temporary columns), beside everybody else who is doing the same - nobody waits for a leader to be told where;
//   * the caller that OPENED a block is its leader: it waits for one of the device's LANES (a private context each,
//     created on first use), closes the block - whoever arrives now opens the next one -, waits for the last copy-in to
//     finish, writes the header (status, grids, work list) and runs the small path's launch sequence on the lane's
//     context: the header is the only upload, the binning kernel reads the columns from the pinned block, the occlusion
//     kernels (the matrix-core kernel from 32 768 atoms) write the values into it;
//   * the lane is handed on the moment the kernels have run; every member copies its own slice of the values out of the
//     block, and the last one returns the block to the pool.
//
// No timer decides the batch size: while every lane (three) is busy the arriving calls fill the open block, and its leader
// launches it when a lane comes free - the batches grow to what keeps the lanes busy (group commit).  A lone caller opens
// a block, finds a lane free and runs at once.  max_wait_us > 0 additionally lets a leader that found a lane free hold
// its block open until as many calls have joined as the last batch had, or the time is up.
//
// Anything unusual - non-finite input, a radius that is no cell size, more atoms than a call's share, timing or a
// measurement switch on the context - is not the combiner's: the call runs by itself on the caller's context, where the
// general path validates and reports it.  So a call's error never reaches its batch-mates; what can fail a whole batch
// is the device (a HIP error, out of memory), and then every member gets that status on its own context.
#include "engine_internal.h"

namespace rsasa {

namespace {

constexpr size_t kCombineCallAtoms = 32768;   // a call with more atoms runs by itself (it fills the GPU on its own)
constexpr size_t kBlockAtoms = 196608;        // atoms of one merged batch (a block's records: 4.7 MB of pinned memory)
constexpr size_t kBlockCalls = 256;           // calls of one merged batch
constexpr size_t kBlockWindows = 4096;        // entries of its binning work list
constexpr int kMaxLanes = 8, kMaxBlocks = 2 * kMaxLanes + 2;

struct Block;

struct Request {
    SmallSource in;
    uint32_t n = 0;
    float probe = 0.f;
    uint32_t n_points = 0;
    rsasa_context *ctx = nullptr;
    // the settings that must agree within a batch
    int simd_width = 8;
    OcclusionTuning tuning;
    bool has_id = false;
    // the caller's own preparation
    StructGrid grid{};
    uint32_t atom_off = 0;    // its range of the block's columns

    bool same_settings(const Request &o) const
    {
        return std::memcmp(&probe, &o.probe, 4) == 0 && n_points == o.n_points && simd_width == o.simd_width && has_id == o.has_id &&
               tuning.kernel_version == o.tuning.kernel_version && tuning.atoms_per_wave == o.tuning.atoms_per_wave;
    }
};

struct Block {
    char *h = nullptr, *hout = nullptr;  // pinned: header (status, grids, work list) and the values
    rsasa_atom_t *recs = nullptr;        // pinned: the calls' atoms as 24-byte records, in the order the calls joined
    size_t header_bytes = 0;             // room of `h`: the header of kBlockCalls calls with kBlockWindows work-list entries
    // under Combiner::mu
    enum State { kFree, kOpen, kClosed } state = kFree;
    Request key;                         // the settings of its calls
    std::vector<StructGrid> grids;       // one per call, in the order they joined
    std::vector<uint32_t> sizes;
    size_t atoms = 0, windows = 0;
    uint32_t writers = 0;                // calls still copying their atoms in
    // under mu
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    std::atomic<bool> done_flag{false};  // (the same, for members that spin a little before they sleep)
    int rc = RSASA_OK;
    std::string error;
    uint32_t consumed = 0;
};

struct Combiner {
    std::mutex mu;
    std::condition_variable cv_block;    // "a block is free again"
    std::condition_variable cv_lane;     // leaders: "a lane is free" / "the last writer of your block is done" / "someone joined"
    Block block[kMaxBlocks];
    struct Lane {
        rsasa_context *ctx = nullptr;
        bool busy = false;
    } lane[kMaxLanes];
    int n_lanes = 0, n_blocks = 0;  // (0: not configured yet)
    size_t last_batch = 1;
    std::atomic<uint64_t> batches{0}, calls{0};
    // RSASA_COMBINE_TRACE=1 (under RSASA_TUNING=1): where the time goes, summed over batches / calls, printed at exit
    std::atomic<uint64_t> ns_lane{0}, ns_close{0}, ns_run{0}, ns_call{0}, ns_copy_in{0};
    ~Combiner()
    {
        const uint64_t b = batches.load(), c = std::max<uint64_t>(calls.load(), 1);
        if (b && tuning_env("RSASA_COMBINE_TRACE"))
            std::fprintf(stderr, "call combining: %llu batches, %.2f calls each; per batch us: leader waited %.1f for a lane, %.1f for the last copy-in, "
                         "header + kernels %.1f (of it: header written and launches queued %.1f, waited for the stream %.1f); per call us: copy-in %.1f, whole call %.1f\n", (unsigned long long)b, (double)c / (double)b,
                         ns_lane.load() / 1e3 / b, ns_close.load() / 1e3 / b, ns_run.load() / 1e3 / b, g_small_trace_ns[0].load() / 1e3 / b, g_small_trace_ns[1].load() / 1e3 / b,
                         ns_copy_in.load() / 1e3 / c, ns_call.load() / 1e3 / c);
    }
};
Combiner g_combine[64];
inline uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The pinned buffers of a block, allocated when it is first opened (under Combiner::mu: a handful of times per process).
int block_allocate(rsasa_context *ctx, Block &b)
{
    if (b.h) return RSASA_OK;
    RS_DEVICE(ctx);
    b.header_bytes = small_layout(kBlockCalls, 0, kBlockWindows, 0, true, 0).o_x;
    // one allocation: header | records | values
    const size_t rec_at = (b.header_bytes + 255) & ~size_t(255), out_at = rec_at + kBlockAtoms * sizeof(rsasa_atom_t);
    void *p = nullptr;
    RS_HIP(ctx, hipHostMalloc(&p, out_at + kBlockAtoms * 4, hipHostMallocDefault));
    b.h = (char *)p;
    b.recs = (rsasa_atom_t *)(b.h + rec_at);
    b.hout = b.h + out_at;
    b.grids.reserve(kBlockCalls);
    b.sizes.reserve(kBlockCalls);
    return RSASA_OK;
}

// The closed block's header and kernels on the lane's context; returns the batch's status (message in *error).
int run_block(Combiner::Lane &lane, Block &b, int device, std::string *error)
{
    const size_t S = b.grids.size();
    std::vector<uint4> windows;
    windows.reserve(b.windows);
    size_t N = 0;
    unsigned long long cells16 = 0;
    for (size_t i = 0; i < S; i++) {
        StructGrid &g = b.grids[i];
        g.atom_begin = (uint32_t)N;
        g.sorted_base = (uint32_t)N;
        g.cell_base = (uint32_t)cells16;
        cells16 += lds_cell_slots(g.n_cells);
        for (uint32_t w = 0; w < grid_windows(g.n_cells); w++) windows.push_back(make_uint4((uint32_t)i, w, (uint32_t)N, b.sizes[i]));
        N += b.sizes[i];
    }
    if (!lane.ctx) {
        // (RSASA_COMBINE_OWN_QUEUES=2: a hardware queue of its own for the lane's stream, context.cpp new_stream - measured no
        // better with three lanes and worse with more: 148 k / 218 k against 153 k / 227 k on the runtime's pooled queues)
        static const int own = [] { const char *v = tuning_env("RSASA_COMBINE_OWN_QUEUES"); return v ? std::atoi(v) : 0; }();
        int rc = context_create(device, own, &lane.ctx);
        if (rc) { *error = "call combining: no context for the batch's lane"; return rc; }
    }
    rsasa_context *lc = lane.ctx;
    auto run = [&]() -> int {
        // the lane computes with the members' settings (they agree: same_settings)
        lc->simd_width = b.key.simd_width;
        const uint32_t hint = lc->tuning.deferred_hint;
        lc->tuning = b.key.tuning;
        lc->tuning.deferred_hint = hint;
        RS_DEVICE(lc);
        Lattice lat{};
        int rc = get_lattice(lc, b.key.n_points, &lat);
        if (rc) return rc;
        SmallLayout lay = small_layout(S, N, windows.size(), 0, b.key.has_id, cells16);  // (of the device block; its header part is h's)
        lay.probe = b.key.probe;
        lc->tuning.deferred_hint = 0;  // (nobody reads a lane's deferred count back: the smallest launch over the list - any size is correct)
        if ((rc = small_reserve(lc, lay, lat, false))) return rc;
        return small_run(lc, lay, lat, b.grids.data(), windows.data(), b.h, b.hout, b.recs);
    };
    const int rc = run();
    if (rc) *error = lc->last_error;
    return rc;
}

}  // namespace

int combine_call(rsasa_context *ctx, const SmallSource &in, size_t n_atoms, float probe, size_t n_points, float *out)
{
    if (ctx->device < 0 || ctx->device >= 64 || n_atoms == 0 || n_atoms > kCombineCallAtoms || n_points == 0 || n_points > (1u << 24) ||
        !(probe >= 0.0f) || !std::isfinite(probe))
        return kNotCombined;
    const uint64_t t_in = now_ns();
    Request rq;
    rq.in = in;
    rq.n = (uint32_t)n_atoms;
    rq.probe = probe;
    rq.n_points = (uint32_t)n_points;
    rq.ctx = ctx;
    rq.has_id = in.has_id();
    {
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        if (ctx->timing || ctx->tuning.debug_stop || !ctx->small_path) return kNotCombined;
        rq.simd_width = ctx->simd_width;
        rq.tuning = ctx->tuning;
    }
    // the caller's share of the grid build: bounding box, largest radius, cell grid (anything unusual: not the combiner's)
    if (!small_structure_grid(in, 0, rq.n, probe, &rq.grid)) return kNotCombined;
    const uint32_t n_windows = grid_windows(rq.grid.n_cells);
    if (n_windows > kBlockWindows / 4) return kNotCombined;
    const int wait_us = ctx->combine_wait_us.load(std::memory_order_relaxed);

    Combiner &cb = g_combine[ctx->device];
    std::unique_lock<std::mutex> lk(cb.mu);
    if (cb.n_lanes == 0) {
        cb.n_lanes = 3;  // (measured, 16 / 64 caller threads: 2 lanes 112 k / 180 k structures/s, 3 lanes 153 k / 227 k, 4 146 k / 222 k, 8 138 k / 165 k)
        if (const char *v = tuning_env("RSASA_COMBINE_LANES")) cb.n_lanes = std::min(kMaxLanes, std::max(1, std::atoi(v)));
        cb.n_blocks = 2 * cb.n_lanes + 2;
    }
    // ---- join the open block of these settings, or open one ----
    Block *b = nullptr;
    bool leader = false;
    for (;;) {
        Block *free_block = nullptr;
        for (int k = 0; k < cb.n_blocks && !b; k++) {
            Block &c = cb.block[k];
            if (c.state == Block::kOpen && c.key.same_settings(rq) && c.sizes.size() < kBlockCalls && c.atoms + rq.n <= kBlockAtoms &&
                c.windows + n_windows <= kBlockWindows)
                b = &c;
            else if (c.state == Block::kFree && !free_block)
                free_block = &c;
        }
        if (b) break;
        if (free_block) {
            const int rc = block_allocate(ctx, *free_block);
            if (rc) return rc;
            b = free_block;
            b->state = Block::kOpen;
            b->key = rq;
            b->grids.clear();
            b->sizes.clear();
            b->atoms = b->windows = 0;
            b->writers = 0;
            b->done = false;
            b->done_flag.store(false, std::memory_order_relaxed);
            b->rc = RSASA_OK;
            b->consumed = 0;
            leader = true;
            break;
        }
        cb.cv_block.wait(lk);  // (every block is running or being read out: one comes back within a batch's time)
    }
    rq.atom_off = (uint32_t)b->atoms;
    b->grids.push_back(rq.grid);
    b->sizes.push_back(rq.n);
    b->atoms += rq.n;
    b->windows += n_windows;
    b->writers++;
    if (!leader && wait_us >= 0) cb.cv_lane.notify_all();  // (a leader holding its block open counts the calls that joined)
    lk.unlock();

    // ---- my atoms into my range of the block, beside everybody else's ----
    const uint64_t t_c0 = now_ns();
    small_fill_records(rq.in, 0, rq.n, b->recs, rq.atom_off);
    cb.ns_copy_in.fetch_add(now_ns() - t_c0, std::memory_order_relaxed);

    int rc = RSASA_OK;
    std::string error;
    if (leader) {
        const uint64_t t0 = now_ns();
        lk.lock();
        b->writers--;
        int my_lane = -1;
        auto free_lane = [&] {
            for (int k = 0; k < cb.n_lanes; k++)
                if (!cb.lane[k].busy) return k;
            return -1;
        };
        cb.cv_lane.wait(lk, [&] { return (my_lane = free_lane()) >= 0; });
        cb.lane[my_lane].busy = true;
        if (wait_us > 0 && b->sizes.size() < cb.last_batch) {  // (a lane was free: hold the block open for company, for a while)
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
            cb.cv_lane.wait_until(lk, deadline, [&] { return b->sizes.size() >= cb.last_batch; });
        }
        b->state = Block::kClosed;  // (whoever arrives now opens the next block)
        const uint64_t t1 = now_ns();
        cb.cv_lane.wait(lk, [&] { return b->writers == 0; });
        cb.last_batch = b->sizes.size();
        lk.unlock();
        const uint64_t t2 = now_ns();
        rc = run_block(cb.lane[my_lane], *b, ctx->device, &error);
        const uint64_t t3 = now_ns();
        lk.lock();
        cb.lane[my_lane].busy = false;
        lk.unlock();
        cb.cv_lane.notify_all();
        {
            std::lock_guard<std::mutex> bl(b->mu);
            b->rc = rc;
            b->error = error;
            b->done = true;
            b->done_flag.store(true, std::memory_order_release);
        }
        b->cv.notify_all();
        cb.batches.fetch_add(1, std::memory_order_relaxed);
        cb.calls.fetch_add(b->sizes.size(), std::memory_order_relaxed);
        cb.ns_lane.fetch_add(t1 - t0, std::memory_order_relaxed);
        cb.ns_close.fetch_add(t2 - t1, std::memory_order_relaxed);
        cb.ns_run.fetch_add(t3 - t2, std::memory_order_relaxed);
    } else {
        lk.lock();
        const bool last_writer = --b->writers == 0 && b->state == Block::kClosed;
        lk.unlock();
        if (last_writer) cb.cv_lane.notify_all();
        static const int spin_us = [] { const char *v = tuning_env("RSASA_COMBINE_SPIN_US"); return v ? std::atoi(v) : 0; }();
        if (spin_us > 0) {
            const uint64_t until = now_ns() + (uint64_t)spin_us * 1000u;
            while (!b->done_flag.load(std::memory_order_acquire) && now_ns() < until) __builtin_ia32_pause();
        }
        std::unique_lock<std::mutex> bl(b->mu);
        b->cv.wait(bl, [&] { return b->done; });
        rc = b->rc;
        if (rc) error = b->error;
    }
    if (!rc) std::memcpy(out, b->hout + 4u * (size_t)rq.atom_off, 4u * (size_t)rq.n);
    // the last reader returns the block
    bool last_reader;
    {
        std::lock_guard<std::mutex> bl(b->mu);
        last_reader = ++b->consumed == b->sizes.size();
    }
    if (last_reader) {
        lk.lock();
        b->state = Block::kFree;
        lk.unlock();
        cb.cv_block.notify_all();
    }
    cb.ns_call.fetch_add(now_ns() - t_in, std::memory_order_relaxed);
    return rc ? fail(ctx, rc, error.c_str()) : RSASA_OK;
}

}  // namespace rsasa

extern "C" {

int rsasa_context_set_call_combining(rsasa_context_t *ctx, int max_wait_us)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    ctx->combine_wait_us.store(max_wait_us < 0 ? -1 : std::min(max_wait_us, 100000), std::memory_order_relaxed);
    return RSASA_OK;
}

int rsasa_call_combining_stats(int device, uint64_t *out_batches, uint64_t *out_calls)
{
    if (device < 0 || device >= 64) return RSASA_ERR_INVALID_ARGUMENT;
    if (out_batches) *out_batches = g_combine[device].batches.load(std::memory_order_relaxed);
    if (out_calls) *out_calls = g_combine[device].calls.load(std::memory_order_relaxed);
    return RSASA_OK;
}

}  // extern "C"
