// C ABI of the engine (include/rustsasa_amd.h): contexts, HBM workspace,
// sphere lattice cache, batch enqueue / wait.  Host code only; the kernels
// live in kernels.hip.  There is no CPU compute path in this library.
#include "../../include/rustsasa_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "device_types.h"

namespace rsasa {

// The library's measurement switches (kernel choice, sub-batch sizes, copy experiments: INTEGRATION.md 7) are read only
// when the process sets RSASA_TUNING=1: a production program's numerics path does not depend on stray environment
// variables.  (RSASA_NUMA=0, which only says where threads may run, is read regardless.)
const char *tuning_env(const char *name)
{
    static const bool on = [] { const char *v = std::getenv("RSASA_TUNING"); return v && std::atoi(v) != 0; }();
    return on ? std::getenv(name) : nullptr;
}

// Golden-section spiral (reference src/lib.rs:43-66, constants
// src/utils/consts.rs:18-19), f32 at every step, libm transcendentals.
// Computed on the host and uploaded: the occlusion decision is integer valued,
// so the table must be bit-identical to what the reference's host would use.
void generate_sphere_points(size_t n, float *x, float *y, float *z)
{
    const float pi = 3.14159274101257324219f;
    const float angle_increment = (2.0f * pi) * 1.618034f;
    const float inv_n = 1.0f / (float)n;
    for (size_t i = 0; i < n; i++) {
        const float fi = (float)i;
        const float t = fi * inv_n;
        const float inclination = acosf(1.0f - 2.0f * t);
        const float azimuth = angle_increment * fi;
        const float si = sinf(inclination);
        x[i] = si * cosf(azimuth);
        y[i] = si * sinf(azimuth);
        z[i] = cosf(inclination);
    }
}

// ---- sphere points in patches (point counts above 128 only) ----
// Every kernel only COUNTS points, so the order of the points that share a rule is free.  With many points the
// matrix-core kernel first tests whole patches of 16 points against the nearest candidates (one candidate whose
// cap holds the patch kills all 16: occlusion_mx.inc); for that the fused-rule points [0, n_fused) are reordered so
// that every aligned run of 16 is a compact patch of the sphere (recursive bisection along the widest axis, left
// halves in multiples of 16).  The remainder points keep their places behind them.
static void bisect_points(std::vector<uint32_t> &idx, size_t lo, size_t hi, const float *x, const float *y, const float *z)
{
    const size_t n = hi - lo;
    if (n <= 16) return;
    float mn[3] = {2.f, 2.f, 2.f}, mx[3] = {-2.f, -2.f, -2.f};
    for (size_t i = lo; i < hi; i++) {
        const float c[3] = {x[idx[i]], y[idx[i]], z[idx[i]]};
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], c[k]); mx[k] = std::max(mx[k], c[k]); }
    }
    int ax = 0;
    for (int k = 1; k < 3; k++)
        if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
    const float *c = ax == 0 ? x : ax == 1 ? y : z;
    std::sort(idx.begin() + (long)lo, idx.begin() + (long)hi,
              [c](uint32_t a, uint32_t b) { return c[a] != c[b] ? c[a] < c[b] : a < b; });
    const size_t left = ((n + 15) / 16 / 2) * 16;
    bisect_points(idx, lo, lo + left, x, y, z);
    bisect_points(idx, lo + left, hi, x, y, z);
}

static uint16_t f16_bits(_Float16 h)
{
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
// smallest f16 >= v (v finite, non-negative, below the f16 range's end)
static uint16_t f16_round_up(float v)
{
    _Float16 h = (_Float16)v;
    uint16_t u = f16_bits(h);
    if ((float)h < v) u++;  // next representable value (positive numbers: the bit pattern is monotone)
    return u;
}

struct DeviceBuffer {
    void *p = nullptr;
    size_t cap = 0;
};

// CPUs of the NUMA node a GPU hangs off, from sysfs (numa_node / local_cpulist of its PCI address).  `valid` only on
// machines that have more than one node and say so; RSASA_NUMA=0 switches the whole thing off.  Used to keep the
// context's own threads (and, through rsasa_context_bind_thread, the caller's per-GPU worker threads) next to the
// link their pinned buffers cross - with 8 GPUs on two sockets half of them would otherwise work across the socket
// interconnect (reference: one rayon pool per process, src/main.rs:375; here one context per GPU).
struct NodeCpus {
    bool valid = false;
    int node = -1;
    cpu_set_t set;
};

static NodeCpus device_node_cpus(int device)
{
    NodeCpus nc;
    CPU_ZERO(&nc.set);
    if (const char *v = std::getenv("RSASA_NUMA"))
        if (std::atoi(v) == 0) return nc;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf) - 1, device) != hipSuccess) {
        (void)hipGetLastError();
        return nc;
    }
    for (char *c = bdf; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    const std::string base = std::string("/sys/bus/pci/devices/") + bdf;
    FILE *f = std::fopen((base + "/numa_node").c_str(), "r");
    if (!f) return nc;
    int node = -1;
    const int got = std::fscanf(f, "%d", &node);
    std::fclose(f);
    if (got != 1 || node < 0) return nc;
    f = std::fopen((base + "/local_cpulist").c_str(), "r");
    if (!f) return nc;
    char buf[4096] = {0};
    const bool ok = std::fgets(buf, sizeof(buf), f) != nullptr;
    std::fclose(f);
    if (!ok) return nc;
    int n_set = 0;
    for (char *q = buf; *q;) {  // "0-31,64-95"
        char *end = nullptr;
        const long lo = std::strtol(q, &end, 10);
        if (end == q) break;
        long hi = lo;
        q = end;
        if (*q == '-') {
            hi = std::strtol(q + 1, &end, 10);
            q = end;
        }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &nc.set); n_set++; }
        while (*q == ',' || *q == ' ' || *q == '\n') q++;
    }
    nc.node = node;
    nc.valid = n_set > 0;
    return nc;
}

// Binds a thread to `nc` (intersected with what the thread may run on); false when there is nothing to do.
static bool bind_thread_to(pthread_t th, const NodeCpus &nc)
{
    if (!nc.valid) return false;
    cpu_set_t cur, want;
    if (pthread_getaffinity_np(th, sizeof(cur), &cur) != 0) return false;
    CPU_AND(&want, &cur, &nc.set);
    if (CPU_COUNT(&want) == 0) return false;
    return pthread_setaffinity_np(th, sizeof(want), &want) == 0;
}

struct LatticeEntry {
    float *d = nullptr;  // x | y | z, each `padded` floats, | (x, y, z, 0) records | patch table (16 bytes per patch) | mx_tab
    uint32_t padded = 0;
    uint32_t mx_tab_at = 0;  // float offset of Lattice::mx_tab (0: none, more than 128 points)
    uint32_t n_patches = 0;  // 0: the points are in the reference's order and have no patch table
};

struct Pending {
    bool active = false;
    rsasa_device_batch_t batch{};
    float probe = 0.f;
    size_t n_points = 0;
    hipStream_t stream = nullptr;
    int attempts = 0;
    const uint32_t *id32 = nullptr;  // nullable (pipelined host path): the ids folded by the host; batch.id is then a
                                     // device-accessible pointer the general kernel alone reads (BatchView::id32)
    const uint8_t *radius8 = nullptr;     // nullable (pipelined host path): one-byte radius codes + their table
    const float *radius_table = nullptr;  // (BatchView::radius8); batch.radius is then not read
    int ws = 0;                      // the workspace (and host slot) the batch runs in
    bool ids_needed_known = false;   // the host has checked the ids itself and found that they matter (BatchView::ids_check off)
};

// The distinct radii of a host batch, collected while worker threads turn the radii into one-byte codes: a
// structure file has a dozen distinct radii, so 1 byte per atom crosses the link instead of 4.  More than 256
// distinct values: `failed`, and the f32 radii are uploaded as before.
struct RadiusCodec {
    float table[256];
    std::atomic<int> n{0};
    std::atomic<bool> failed{false};
    std::mutex mu;
    void reset() { n.store(0); failed.store(false); }
    static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
    int code(float value)  // the value's code (compared by bit pattern: -0.0, NaN payloads survive), or -1
    {
        const uint32_t b = bits(value);
        int cnt = n.load(std::memory_order_acquire);
        for (int k = 0; k < cnt; k++)
            if (bits(table[k]) == b) return k;
        std::lock_guard<std::mutex> lk(mu);
        const int now = n.load(std::memory_order_relaxed);
        for (int k = cnt; k < now; k++)
            if (bits(table[k]) == b) return k;
        if (now == 256) { failed.store(true); return -1; }
        table[now] = value;
        n.store(now + 1, std::memory_order_release);
        return now;
    }
};

// What tells whether a batch's ids matter: they do not if the ids of every structure increase strictly (atom serials,
// indices) - then they are all different, and "a neighbour with the atom's own id" (lib.rs:127) is the atom itself.
// starts[0 .. n_starts] are the structures' first atoms, in the numbering of src's entries (src[0] is atom `first`).
struct IdOrder {
    const uint32_t *starts = nullptr;
    size_t n_starts = 0;
    uint32_t first = 0;
    std::atomic<int> *ids_matter = nullptr;  // set to 1 by a worker that finds an id not above its predecessor's
};

// A few worker threads that fold 64-bit ids to 32 bits (device_utils.h fold_id) ahead of the uploads: the
// pipelined host path then moves 4 bytes per id over the link instead of 8.  Jobs (one per sub-batch) are
// worked off in the order they were submitted, every worker taking blocks of the current job.
class FoldPool {
public:
    FoldPool(unsigned n_threads, const NodeCpus &node)
    {
        for (unsigned t = 0; t < n_threads; t++) {
            workers.emplace_back([this] { run(); });
            (void)bind_thread_to(workers.back().native_handle(), node);  // next to the GPU's link (see NodeCpus)
        }
    }
    ~FoldPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        for (auto &w : workers) w.join();
    }
    // queues folding src[0 .. n) into dst (either may be null), coding rad[0 .. n) into rad8 (if codec is set) and
    // checking the order of src (if order.ids_matter is set); returns the job's number for wait()
    unsigned long long submit(const uint64_t *src, uint32_t *dst, size_t n, const float *rad = nullptr, uint8_t *rad8 = nullptr,
                              RadiusCodec *codec = nullptr, IdOrder order = IdOrder())
    {
        std::lock_guard<std::mutex> lk(mu);
        jobs.push_back(Job{src, dst, n, 0, 0, rad, rad8, codec, order});
        cv.notify_all();
        return first_job + jobs.size() - 1;
    }
    void wait(unsigned long long job)  // returns once that job (and every earlier one) is done
    {
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return first_job > job; });
    }

private:
    static constexpr size_t kBlock = 1u << 16;
    struct Job {
        const uint64_t *src;
        uint32_t *dst;
        size_t n, next, finished;  // next block to hand out, blocks finished
        const float *rad;
        uint8_t *rad8;
        RadiusCodec *codec;
        IdOrder order;
    };
    void run()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return quit || (!jobs.empty() && jobs.front().next * kBlock < std::max<size_t>(jobs.front().n, 1)); });
            if (quit) return;
            Job &j = jobs.front();
            const size_t blk = j.next++, n_blocks = (std::max<size_t>(j.n, 1) + kBlock - 1) / kBlock;
            const uint64_t *s = j.src;
            uint32_t *d = j.dst;
            const float *rad = j.rad;
            uint8_t *rad8 = j.rad8;
            RadiusCodec *codec = j.codec;
            const IdOrder order = j.order;
            const size_t b = blk * kBlock, e = std::min(j.n, b + kBlock);
            lk.unlock();
            if (s && order.ids_matter && !order.ids_matter->load(std::memory_order_relaxed)) {
                for (size_t i = std::max<size_t>(b, 1); i < e; i++) {
                    if (s[i] > s[i - 1]) continue;
                    // (rare: a structure's first atom - serials start over - or ids that do matter)
                    const uint32_t atom = order.first + (uint32_t)i;
                    const uint32_t *hit = std::lower_bound(order.starts, order.starts + order.n_starts, atom);
                    if (hit == order.starts + order.n_starts || *hit != atom) {
                        order.ids_matter->store(1, std::memory_order_relaxed);
                        break;
                    }
                }
            }
            if (s && d)
                for (size_t i = b; i < e; i++) d[i] = (uint32_t)s[i] ^ ((uint32_t)(s[i] >> 32) * 0x9E3779B1u);  // fold_id
            if (codec && !codec->failed.load(std::memory_order_relaxed)) {
                uint32_t last_bits = 0;
                int last_code = -1;  // (runs of equal radii are common: backbone N, CA, C, O repeat)
                for (size_t i = b; i < e; i++) {
                    const uint32_t bt = RadiusCodec::bits(rad[i]);
                    if (last_code < 0 || bt != last_bits) {
                        last_code = codec->code(rad[i]);
                        last_bits = bt;
                        if (last_code < 0) break;
                    }
                    rad8[i] = (uint8_t)last_code;
                }
            }
            lk.lock();
            // (the job is still the front one: it leaves the queue only when all its blocks are finished)
            if (++jobs.front().finished == n_blocks) {
                jobs.pop_front();
                first_job++;
                done_cv.notify_all();
                cv.notify_all();
            }
        }
    }
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::deque<Job> jobs;
    unsigned long long first_job = 0;  // number of the job at the front of the queue
    bool quit = false;
};

// rsasa_host_batch_enqueue / _wait: a stream of host batches on one context handle.  Two worker threads, each with a
// private context on the caller's GPU, run rsasa_calculate_sasa_batch on the queued batches in order; the link turn
// (LinkTurn, below) lets the second call's uploads follow the first one's.  Results are handed back oldest first.
// The order in which the calls of one stream take their turns on the link is the order of the batches: the caller waits
// for the OLDEST batch, and a younger one that slipped ahead on the link delays exactly that one (two workers woken
// together: the second batch uploaded first, the first one's results came after both, and the caller - who enqueues the
// next batch when the oldest returns - kept one batch in flight where it meant two).
struct LinkGate {
    std::mutex mu;
    std::condition_variable cv;
    uint64_t next = 1;  // the ticket whose turn it is
    void wait_for(uint64_t ticket)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return next >= ticket; });
    }
    void advance(uint64_t ticket)  // `ticket` has queued its uploads (or will not queue any): idempotent
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (next > ticket) return;
            next = ticket + 1;
        }
        cv.notify_all();
    }
};

struct HostStream {
    struct Job {
        uint64_t ticket = 0;
        const float *x, *y, *z, *radius;
        const uint64_t *id;
        const uint32_t *structure_offsets;
        size_t n_structures;
        float probe;
        size_t n_points;
        float *out_atom;
        const uint32_t *residue_offsets;
        size_t n_residues;
        float *out_res;
        // the caller's settings at the enqueue (the worker's context takes them before it computes)
        int simd_width = 8;
        bool small_path = true, overlap_tail = false;
        OcclusionTuning tuning;
        int rc = 0;
        std::string error;
        bool taken = false, done = false;
    };
    static constexpr int kMaxWorkers = 4;
    int n_workers = 2;
    // The workers' contexts create their streams on hardware queues of their own (new_stream).  The next call's uploads
    // hide a call's fill and drain, so each call is cut into two sub-batches only (measured, ms per proteome batch:
    // 2 sub-batches 4.30, 3 5.15, 8 5.29; on the pooled queues 6.40 / 6.53 / 5.34).
    size_t sub_batches = 0;
    static constexpr size_t kMaxQueued = 8;  // enqueued and not yet waited for (a further enqueue waits for the oldest to complete)
    rsasa_context *sub[kMaxWorkers] = {};
    std::thread th[kMaxWorkers];
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::shared_ptr<Job>> jobs;  // oldest first; entries leave in rsasa_host_batch_wait
    uint64_t next_ticket = 1;
    LinkGate gate;
    bool quit = false;
};

}  // namespace rsasa

using namespace rsasa;

struct rsasa_context {
    int device = 0;
    NodeCpus node;                                // CPUs of the GPU's NUMA node (valid on multi-node hosts only)
    hipStream_t stream = nullptr;
    std::recursive_mutex mu;
    std::string last_error;
    int simd_width = 8;
    bool timing = false;
    bool small_path = true;                       // RSASA_SMALL_PATH=0: small host batches take the general path too
    bool overlap_tail = false;                    // RSASA_OVERLAP_TAIL=1: bin the tail on the side stream, next to the first
                                                  // occlusion launch (only batches with a structure of 65 536 atoms or more have a tail now)
    hipStream_t side_stream = nullptr;            // runs the tail's binning next to the launch stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    rsasa_timings_t timings{};
    bool timings_valid = false;

    // Workspace (device) of one batch in flight.  Two of them: rsasa_batch_enqueue starts batch k + 1 in the other
    // one (on the context's second stream) while batch k still runs - the small kernels at the start and the thin
    // tail of the occlusion kernel at the end of a batch then overlap with its neighbour's.  Host slot w serves
    // workspace w.  Everything else (host-pointer entry points, sub-batches of the pipelined host path) runs in
    // workspace 0.
    struct Workspace {
        DeviceBuffer segments, acc, grids, grid_sums, sid_sorted, deferred_list, cell_of, rank_of, cells, windows, scan_sums,
            sorted_xyzr, sorted_orig, sorted_id, sorted_id32, status, atom_sasa, claim;
        hipEvent_t ev[5] = {};  // timing (rsasa_context_enable_timing): start, grid built, occlusion starts / has run, sums done
        hipEvent_t ev_occ = nullptr;  // the batch's occlusion kernels have run (the other workspace's batch starts its own
        bool occ_recorded = false;    // behind it: two occlusion kernels sharing the CUs only slow each other down)
    } ws[2];
    static constexpr int kInFlight = 2;
    hipStream_t stream2 = nullptr;                // launch stream of workspace 1 (created by the first overlapped enqueue)
    // Experiment (RSASA_GRID_CUS=N): N compute units are set aside for the grid builds - a stream masked to
    // them - and the two launch streams are masked to the others, so batch k + 1's grid build runs BESIDE batch k's
    // occlusion kernel instead of waiting for its workgroups to retire.
    uint32_t grid_cus = 0, cu_mask_words = 0;
    uint32_t cu_reserved[16] = {}, cu_rest[16] = {};
    hipStream_t grid_stream = nullptr;
    hipEvent_t ev_grid[2] = {nullptr, nullptr}, ev_grid_in[2] = {nullptr, nullptr};
    DeviceBuffer &segments = ws[0].segments, &acc = ws[0].acc, &grids = ws[0].grids, &grid_sums = ws[0].grid_sums,
                 &sid_sorted = ws[0].sid_sorted, &deferred_list = ws[0].deferred_list, &cell_of = ws[0].cell_of,
                 &rank_of = ws[0].rank_of, &cells = ws[0].cells, &windows = ws[0].windows, &scan_sums = ws[0].scan_sums,
                 &sorted_xyzr = ws[0].sorted_xyzr, &sorted_orig = ws[0].sorted_orig, &sorted_id = ws[0].sorted_id,
                 &sorted_id32 = ws[0].sorted_id32, &status = ws[0].status, &atom_sasa = ws[0].atom_sasa, &claim = ws[0].claim;
    // staging for the host-pointer entry points (device)
    DeviceBuffer in_x, in_y, in_z, in_r, in_id, in_res, out_res, out_k;
    // Further input / output slots of the pipelined host-buffer path: a slot per sub-batch of a call (kSlots >= the
    // most sub-batches a call is cut into), so the uploads never wait for a slot - they follow each other at the
    // link's rate however far the kernels are behind, and in a stream of host batches (rsasa_host_batch_enqueue) the
    // next call's first upload follows this call's last one while this call's kernels are still running.  With three
    // slots the link idled at every call boundary until the new call's first sub-batch had been computed (5.3 ms per
    // proteome batch in a stream, no better than one call after the other).
    static constexpr int kSlots = 8;
    struct MoreSlot { DeviceBuffer x, y, z, r, id, res, atom_sasa, out_res; } more[kSlots - 1];
    // Pipelined host path: everything of a sub-batch that the host prepares - radius table, rebased residue offsets,
    // folded ids, radius codes - sits in ONE pinned block per sub-batch and crosses the link in ONE copy (every
    // copy costs the link about 12 us of idle time).
    DeviceBuffer in_pack[kSlots];                 // that block of the sub-batch in slot k, on the device
    char *h_pack = nullptr;                       // pinned: the blocks of a whole host batch
    size_t h_pack_cap = 0;
    FoldPool *fold_pool = nullptr;                // the device's shared coding pool (first large host call; never owned)
    RadiusCodec radius_codec;
    hipStream_t copy_stream = nullptr;            // H2D of the next sub-batch while the current one computes
    hipStream_t d2h_stream = nullptr;             // D2H of the previous sub-batch's results meanwhile
    hipEvent_t ev_copy[kSlots] = {};
    hipEvent_t ev_d2h[kSlots] = {};               // output slot k has been copied out
    void *h_out[kSlots] = {};                     // pinned staging for results whose destination is pageable
    size_t h_out_cap[kSlots] = {};
    DeviceBuffer small_in, small_out;          // small host batches: one upload / one download buffer
    void *h_small = nullptr;                   // pinned staging of the same layout
    size_t h_small_cap = 0;
    DeviceBuffer tr_xyz, tr_r, tr_id, tr_res;  // trajectory staging (frame-major xyz, per-topology columns)
    // pinned host
    // Host side of one enqueued batch (pinned): its bounds segments (source of an async upload) and
    // the status block the device writes back.  Slot 0 serves the batch entry points; the pipelined
    // host-buffer path keeps two sub-batches in flight and alternates between slots 0 and 1.
    struct HostSlot {
        Segment *h_segments = nullptr;
        size_t h_segments_cap = 0;
        BatchStatus *h_status = nullptr;
        uint32_t *h_res = nullptr;      // rebased residue offsets of a sub-batch (a pageable source would
        size_t h_res_cap = 0;           // make the "asynchronous" upload wait for the copy stream)
        bool ids_check = false;         // the batch that last used the slot ran with BatchView::ids_check
    } slot[kSlots];
    std::atomic<uint64_t> ids_dropped{0};         // batches / sub-batches that ran without their ids (rsasa_context_ids_dropped)
    bool ids_drop_hint = true;                    // what the last checked batch did (OcclusionChain::expect_ids_dropped)
    bool ids_unordered_hint = false;              // its ids were in no order: the next batch brings the id tables (BatchView::ids_tables)
    hipEvent_t ev_done[kSlots] = {};              // all work of the sub-batch in slot k has been executed
    uint64_t cell_capacity = 0;

    std::map<std::pair<size_t, int>, LatticeEntry> lattices;
    Pending pending[2];   // device batches in flight, oldest first: pending[head], pending[head ^ 1]
    int head = 0, n_pending = 0;
    OcclusionTuning tuning;
    hipEvent_t ev_link = nullptr;  // recorded behind the last upload of a pipelined host call (LinkTurn)
    hipEvent_t tr_ev[8][4] = {};   // RSASA_H2H_TRACE: a sub-batch's uploads and kernels, start and end
    LinkGate *link_gate = nullptr; // a worker context of a stream of host batches: the calls take the link in ticket order
    uint64_t link_ticket = 0;
    int own_queues = 0;            // 1: the copy streams, 2: every stream on a hardware queue of its own (new_stream)
    size_t stream_sub_batches = 0; // a worker context of a stream of host batches: most sub-batches of a call (0: the default)
    struct HostStream *host_stream = nullptr;  // rsasa_host_batch_enqueue / _wait: two workers with a context each
};

namespace {

int fail(rsasa_context *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        // several threads may share a context (host_api.cpp runs two workers on one): the message
        // is written and read under the context's (recursive) mutex
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        ctx->last_error = what;
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return code;
}

// Entry points run on the context's device and leave the calling thread's current device as
// they found it (a host program with several GPUs - or torch - keeps its own current device).
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        err = prev == device ? hipSuccess : hipSetDevice(device);
        if (prev == device) prev = -1;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define RS_DEVICE(ctx)                                                        \
    DeviceGuard device_guard_((ctx)->device);                                 \
    if (device_guard_.err != hipSuccess)                                      \
        return fail((ctx), RSASA_ERR_HIP, "hipSetDevice", device_guard_.err)

#define RS_HIP(ctx, expr)                                                           \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess)                                                       \
            return fail((ctx), e_ == hipErrorOutOfMemory ? RSASA_ERR_OUT_OF_MEMORY \
                                                         : RSASA_ERR_HIP,           \
                        #expr, e_);                                                 \
    } while (0)

// ---- one upload phase at a time per device ----------------------------------------------------------------------
// A pipelined host call is bound by the link and by the kernels at once; its first sub-batch's upload hides behind
// nothing and nothing hides its last sub-batches' kernels.  A STREAM of host batches - two contexts on one GPU, each
// with a call in flight (rsasa_host_batch_enqueue does exactly that) - hides both, provided the calls take turns on the
// link: two calls uploading at the same time share its 52 GB/s and each other's copy gaps, and both finish later than
// one after the other would (two contexts without turns: 5.9 ms per proteome batch against 5.2 for one).  The turn is
// taken before a call queues its first upload and passed on behind its last one: the next call's copy stream waits for
// that upload's event, its first sub-batch then crosses the link while the previous call's last ones compute.
struct LinkTurn {
    std::mutex mu;
    std::condition_variable cv;
    bool busy = false;
    hipEvent_t last = nullptr;            // behind the previous holder's last upload, on its copy stream
    const rsasa_context *owner = nullptr;  // the context `last` belongs to (cleared when it is destroyed)
};
LinkTurn g_link[64];

struct LinkHold {
    LinkTurn *lt = nullptr;
    rsasa_context *ctx = nullptr;
    bool held = false;
    // waits for the turn; the caller's copy stream then waits for the previous holder's last upload
    hipError_t take(rsasa_context *c, hipStream_t cp)
    {
        if (c->device < 0 || c->device >= 64) return hipSuccess;
        lt = &g_link[c->device];
        ctx = c;
        if (c->link_gate) c->link_gate->wait_for(c->link_ticket);
        hipEvent_t prev = nullptr;
        {
            std::unique_lock<std::mutex> lk(lt->mu);
            lt->cv.wait(lk, [&] { return !lt->busy; });
            lt->busy = true;
            held = true;
            if (lt->owner != c) prev = lt->last;
        }
        return prev ? hipStreamWaitEvent(cp, prev, 0) : hipSuccess;
    }
    // every upload of the call has been queued on `cp`
    void pass(hipStream_t cp)
    {
        if (!held) return;
        const bool ok = ctx->ev_link && hipEventRecord(ctx->ev_link, cp) == hipSuccess;
        std::lock_guard<std::mutex> lk(lt->mu);
        if (ok) { lt->last = ctx->ev_link; lt->owner = ctx; }
        lt->busy = false;
        held = false;
        lt->cv.notify_all();
        if (ctx->link_gate) ctx->link_gate->advance(ctx->link_ticket);
    }
    ~LinkHold()
    {
        if (!held) return;  // (an error return: nothing to order behind)
        std::lock_guard<std::mutex> lk(lt->mu);
        lt->busy = false;
        lt->cv.notify_all();
    }
};

// Grows `b` to at least `bytes` (contents are NOT preserved).  The caller has
// already drained the stream if the buffer may be in use.
int reserve(rsasa_context *ctx, DeviceBuffer &b, size_t bytes)
{
    if (bytes <= b.cap) return RSASA_OK;
    if (b.p) {
        // the buffer may be read by work on any of the context's streams (two launch streams, a caller's stream of a
        // batch in flight, the copy streams of the pipelined host path): drain them all, not only the first
        RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->stream2) RS_HIP(ctx, hipStreamSynchronize(ctx->stream2));
        if (ctx->side_stream) RS_HIP(ctx, hipStreamSynchronize(ctx->side_stream));
        if (ctx->copy_stream) RS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        if (ctx->d2h_stream) RS_HIP(ctx, hipStreamSynchronize(ctx->d2h_stream));
        if (ctx->grid_stream) RS_HIP(ctx, hipStreamSynchronize(ctx->grid_stream));
        for (const Pending &pd : ctx->pending)
            if (pd.active && pd.stream && pd.stream != ctx->stream && pd.stream != ctx->stream2) RS_HIP(ctx, hipStreamSynchronize(pd.stream));
        RS_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        want = bytes;
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, RSASA_ERR_OUT_OF_MEMORY, "hipMalloc(workspace)", e);
    }
    b.cap = want;
    return RSASA_OK;
}

void release(DeviceBuffer &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int get_lattice(rsasa_context *ctx, size_t n_points, Lattice *out)
{
    const auto key = std::make_pair(n_points, ctx->simd_width);
    auto it = ctx->lattices.find(key);
    if (it == ctx->lattices.end()) {
        // the cache holds the few point counts a program uses; a sweep over many counts must not
        // pin 28 bytes per point per count forever
        size_t cached_bytes = 0;
        for (const auto &kv : ctx->lattices) cached_bytes += 7 * sizeof(float) * (size_t)kv.second.padded + 16 * (size_t)kv.second.n_patches + 1536;
        if (ctx->lattices.size() >= 16 || cached_bytes > (64u << 20)) {
            RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (const Pending &pd : ctx->pending)
                if (pd.active && pd.stream != ctx->stream) RS_HIP(ctx, hipStreamSynchronize(pd.stream));
            for (auto &kv : ctx->lattices)
                if (kv.second.d) (void)hipFree(kv.second.d);
            ctx->lattices.clear();
        }
        const uint32_t padded = (uint32_t)((n_points + 63) / 64 * 64);
        const uint32_t n_patches = n_points > 128 ? (uint32_t)((n_points + 15) / 16) : 0u;
        const uint32_t patches_padded = (n_patches + 63u) / 64u * 64u;
        const uint32_t tab_at = 7 * padded + 4 * patches_padded;
        std::vector<float> h((size_t)tab_at + mx_tab_floats((uint32_t)n_points), 0.0f);  // x | y | z | (x, y, z, 0) records | patches | mx_tab
        generate_sphere_points(n_points, h.data(), h.data() + padded, h.data() + 2 * (size_t)padded);
        if (n_patches) {
            // compact patches: permute the fused-rule points (see bisect_points), then one table entry per patch:
            // (cz, cy | cx, -1 | eps, 0 | 0, 0) as f16 - centre c (unit, rounded to nearest) and eps >= the largest
            // distance from c to a point of the patch (+ the centre's rounding), rounded up
            const size_t n_fused = n_points - n_points % (size_t)ctx->simd_width;
            std::vector<uint32_t> idx(n_fused);
            for (size_t i = 0; i < n_fused; i++) idx[i] = (uint32_t)i;
            float *px = h.data(), *py = h.data() + padded, *pz = h.data() + 2 * (size_t)padded;
            bisect_points(idx, 0, n_fused, px, py, pz);
            std::vector<float> t(3 * n_fused);
            for (size_t i = 0; i < n_fused; i++) { t[3 * i] = px[idx[i]]; t[3 * i + 1] = py[idx[i]]; t[3 * i + 2] = pz[idx[i]]; }
            for (size_t i = 0; i < n_fused; i++) { px[i] = t[3 * i]; py[i] = t[3 * i + 1]; pz[i] = t[3 * i + 2]; }
            uint16_t *pt = reinterpret_cast<uint16_t *>(h.data() + 7 * (size_t)padded);
            for (uint32_t k = 0; k < n_patches; k++) {
                const size_t b = 16 * (size_t)k, e = std::min(b + 16, n_points);
                double c[3] = {0, 0, 0};
                for (size_t i = b; i < e; i++) { c[0] += px[i]; c[1] += py[i]; c[2] += pz[i]; }
                const double len = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
                _Float16 c16[3];
                for (int q = 0; q < 3; q++) c16[q] = (_Float16)(float)(len > 0 ? c[q] / len : (q == 2 ? 1.0 : 0.0));
                double eps = 0;  // against the centre the kernel will actually use (the f16 one)
                for (size_t i = b; i < e; i++) {
                    const double dx = px[i] - (double)(float)c16[0], dy = py[i] - (double)(float)c16[1], dz = pz[i] - (double)(float)c16[2];
                    eps = std::max(eps, std::sqrt(dx * dx + dy * dy + dz * dz));
                }
                uint16_t *en = pt + 8 * (size_t)k;
                en[0] = f16_bits(c16[2]); en[1] = f16_bits(c16[1]); en[2] = f16_bits(c16[0]); en[3] = f16_bits((_Float16)-1.0f);
                en[4] = f16_round_up((float)(eps * 1.001 + 1e-4)); en[5] = en[6] = en[7] = 0;
            }
            // (entries of patches that do not exist: centre 0, -1 -> 0, eps 0: all zero, never looked at)
        }
        for (size_t i = 0; i < n_points; i++) {
            float *r4 = h.data() + 3 * (size_t)padded + 4 * i;
            r4[0] = h[i];
            r4[1] = h[padded + i];
            r4[2] = h[2 * (size_t)padded + i];
        }
        if (n_points <= 128) {
            // the matrix-core kernel's operand tables (device_types.h mx_tab_floats)
            const uint32_t np = 16u * mx_tab_tiles((uint32_t)n_points), nps = np + 16u;
            float *t = h.data() + tab_at;
            uint16_t *t16 = reinterpret_cast<uint16_t *>(t + 4 * (size_t)nps);
            for (size_t i = 0; i < n_points; i++) {
                const float px = h[i], py = h[padded + i], pz = h[2 * (size_t)padded + i];
                t[i] = pz; t[nps + i] = py; t[2 * (size_t)nps + i] = px; t[3 * (size_t)nps + i] = -1.0f;
                t16[4 * i] = f16_bits((_Float16)pz); t16[4 * i + 1] = f16_bits((_Float16)py);
                t16[4 * i + 2] = f16_bits((_Float16)px); t16[4 * i + 3] = f16_bits((_Float16)-1.0f);
            }
        }
        LatticeEntry e;
        e.padded = padded;
        e.n_patches = n_patches;
        e.mx_tab_at = n_points <= 128 ? tab_at : 0u;
        RS_HIP(ctx, hipMalloc((void **)&e.d, h.size() * sizeof(float)));
        hipError_t err = hipMemcpy(e.d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
        if (err != hipSuccess) {
            (void)hipFree(e.d);
            return fail(ctx, RSASA_ERR_HIP, "hipMemcpy(lattice)", err);
        }
        it = ctx->lattices.emplace(key, e).first;
    }
    out->x = it->second.d;
    out->y = it->second.d + it->second.padded;
    out->z = it->second.d + 2 * (size_t)it->second.padded;
    out->xyz4 = (const float4 *)(it->second.d + 3 * (size_t)it->second.padded);
    out->patches = it->second.n_patches ? (const uint4 *)(it->second.d + 7 * (size_t)it->second.padded) : nullptr;
    out->n_patches = it->second.n_patches;
    out->mx_tab = it->second.mx_tab_at ? it->second.d + it->second.mx_tab_at : nullptr;
    out->n_points = (uint32_t)n_points;
    out->n_fused = (uint32_t)(n_points - n_points % (size_t)ctx->simd_width);
    return RSASA_OK;
}

// Enqueues the whole pipeline for the batch `pd` on its stream, using host slot `hs`.
// The streams beside the launch stream are created on first use.  The runtime multiplexes a
// process's streams onto a few hardware queues in creation order: a context that only serves
// per-structure calls should hold ONE stream, or the launch streams of several contexts (one per host
// thread) all land on the same queue and their kernels run one after the other.
//
// own_queues (the worker contexts of a stream of host batches): a stream created with a CU mask gets a hardware queue
// of its own instead of one from the shared pool, and a mask of every CU restricts nothing.  Streams that share a
// queue run their packets in order - the barrier that makes one worker's kernels wait for its upload would hold the
// other worker's kernels behind it, and the stream of batches would not overlap (measured: 6.6 ms per proteome batch
// on pooled queues that collide, 4.4 on queues of their own).
hipError_t new_stream(rsasa_context *ctx, hipStream_t *out, int level)
{
    if (ctx->own_queues >= level) {
        int n_cu = 0;
        uint32_t mask[16] = {};
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && n_cu > 0 && n_cu <= 512) {
            for (int c = 0; c < n_cu; c++) mask[c / 32] |= 1u << (c % 32);
            // (such a stream is a BLOCKING one: work on the process's legacy default stream and work on it wait for each
            // other - ordering a caller may not expect, never a wrong result)
            if (hipExtStreamCreateWithCUMask(out, (uint32_t)(n_cu + 31) / 32, mask) == hipSuccess) return hipSuccess;
            (void)hipGetLastError();  // (a device that does not take the mask: a pooled queue then)
        }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

int ensure_side_stream(rsasa_context *ctx)
{
    if (!ctx->side_stream) RS_HIP(ctx, new_stream(ctx, &ctx->side_stream, 2));
    return RSASA_OK;
}

int ensure_copy_streams(rsasa_context *ctx)
{
    if (!ctx->copy_stream) RS_HIP(ctx, new_stream(ctx, &ctx->copy_stream, 1));
    if (!ctx->d2h_stream) RS_HIP(ctx, new_stream(ctx, &ctx->d2h_stream, 1));
    return RSASA_OK;
}

int enqueue_batch(rsasa_context *ctx, const Pending &pd, rsasa_context::HostSlot &hs)
{
    rsasa_context::Workspace &W = ctx->ws[pd.ws];
    const rsasa_device_batch_t &bt = pd.batch;
    const size_t N = bt.n_atoms, S = bt.n_structures, R = bt.n_residues;
    hipStream_t st = pd.stream;

    Lattice lat;
    int rc = get_lattice(ctx, pd.n_points, &lat);
    if (rc) return rc;

    // bounds segments: <= kSegmentAtoms atoms of one structure each
    size_t n_seg = 0;
    // (behind the segments, in the same upload: the structures whose ids k_ids_distinct's large table takes)
    size_t n_large = 0;
    bool ids_too_big = false;
    for (size_t s = 0; s < S; s++) {
        const uint32_t b = bt.structure_offsets_host[s], e = bt.structure_offsets_host[s + 1];
        n_seg += (e - b + kSegmentAtoms - 1) / kSegmentAtoms;
        n_large += e - b > kIdAtomsSmall && e - b <= kIdAtomsLarge;
        ids_too_big |= e - b > kIdAtomsLarge;
    }
    const size_t n_seg_all = n_seg + (n_large + 3) / 4;  // (four structure numbers per segment-sized entry)
    if (n_seg_all > hs.h_segments_cap) {
        if (hs.h_segments) {
            RS_HIP(ctx, hipStreamSynchronize(pd.stream));
            RS_HIP(ctx, hipHostFree(hs.h_segments));
            hs.h_segments = nullptr;
            hs.h_segments_cap = 0;
        }
        const size_t cap = n_seg_all + n_seg_all / 2 + 64;
        RS_HIP(ctx, hipHostMalloc((void **)&hs.h_segments, cap * sizeof(Segment), hipHostMallocDefault));
        hs.h_segments_cap = cap;
    }
    bool has_tail = false;  // some structure is too large for the LDS binning: the batch-wide kernels run too
    {
        size_t k = 0, kl = 0;
        uint32_t *large = reinterpret_cast<uint32_t *>(hs.h_segments + n_seg);
        for (size_t s = 0; s < S; s++) {
            const uint32_t b = bt.structure_offsets_host[s], e = bt.structure_offsets_host[s + 1];
            has_tail |= e - b >= kLdsMaxAtoms;
            if (e - b > kIdAtomsSmall && e - b <= kIdAtomsLarge) large[kl++] = (uint32_t)s;
            for (uint32_t a = b; a < e; a += kSegmentAtoms)
                hs.h_segments[k++] = Segment{(uint32_t)s, a, std::min(e, a + kSegmentAtoms), a != b ? 1u : 0u};
        }
    }

    if (ctx->cell_capacity == 0)
        ctx->cell_capacity = std::max<uint64_t>(1u << 16, 20ull * N + 512ull * S);
    ctx->cell_capacity = std::min<uint64_t>(ctx->cell_capacity, 0xFFFFFFF0ull);

    const bool has_id = bt.id != nullptr;  // (with pd.id32 set, bt.id is the general kernel's device-accessible copy)
    if ((rc = reserve(ctx, W.segments, std::max<size_t>(n_seg_all, 1) * sizeof(Segment)))) return rc;
    if ((rc = reserve(ctx, W.acc, std::max<size_t>(S, 1) * sizeof(StructAcc)))) return rc;
    if ((rc = reserve(ctx, W.grids, std::max<size_t>(S, 1) * sizeof(StructGrid)))) return rc;
    if ((rc = reserve(ctx, W.grid_sums, (std::max<size_t>(S, 1) + 255) / 256 * 32))) return rc;
    if ((rc = reserve(ctx, W.sid_sorted, std::max<size_t>(N, 1) * 4))) return rc;
    if ((rc = reserve(ctx, W.deferred_list, std::max<size_t>(N, 1) * 4))) return rc;
    if ((rc = reserve(ctx, W.claim, kClaimBytes))) return rc;
    if (has_tail && (rc = reserve(ctx, W.cell_of, std::max<size_t>(N, 1) * 4))) return rc;  // (batch-wide binning only)
    if ((rc = reserve(ctx, W.rank_of, std::max<size_t>(N, 1) * 4))) return rc;
    // + 1 end marker, + 3: k_zero_cells / k_scan_* access whole 16-byte vectors up to the end marker
    if ((rc = reserve(ctx, W.cells, (size_t)(ctx->cell_capacity + 1 + 3) * 4))) return rc;
    // one k_sort_window workgroup per window of kWindowCells 16-bit cell entries (two per entry of the cell
    // array), at most one partly filled window per structure: whatever fits the cell array fits this list
    const uint64_t window_capacity = std::min<uint64_t>(2 * ctx->cell_capacity / kWindowCells + S + 1, 0x7FFFFFFFull);
    if ((rc = reserve(ctx, W.windows, (size_t)window_capacity * sizeof(uint4)))) return rc;
    if ((rc = reserve(ctx, W.scan_sums, kScanBlocks * 4))) return rc;
    if ((rc = reserve(ctx, W.sorted_xyzr, std::max<size_t>(N, 1) * 16))) return rc;
    if ((rc = reserve(ctx, W.sorted_orig, std::max<size_t>(N, 1) * 4))) return rc;
    // (the matrix-core kernel works on the id folds: no sorted copy of the 64-bit ids then)
    const bool keep_ids = has_id && !occlusion_uses_mx(ctx->tuning, lat, (uint32_t)N);
    if (keep_ids && pd.id32) return fail(ctx, RSASA_ERR_INTERNAL, "folded ids on a batch the per-atom kernels take");
    if (keep_ids && (rc = reserve(ctx, W.sorted_id, std::max<size_t>(N, 1) * 8))) return rc;
    if (has_id && (rc = reserve(ctx, W.sorted_id32, std::max<size_t>(N, 1) * 4))) return rc;
    if ((rc = reserve(ctx, W.status, sizeof(BatchStatus)))) return rc;
    if (!bt.out_atom_sasa && (rc = reserve(ctx, W.atom_sasa, std::max<size_t>(N, 1) * 4))) return rc;

    if (n_seg_all)
        RS_HIP(ctx, hipMemcpyAsync(W.segments.p, hs.h_segments, n_seg_all * sizeof(Segment),
                                   hipMemcpyHostToDevice, st));

    BatchView v{};
    v.x = bt.x; v.y = bt.y; v.z = bt.z; v.radius = bt.radius; v.id = bt.id;
    v.id32 = pd.id32;
    // ids that are all different within their structure change nothing: checked on the device (BatchView::ids_check)
    // unless the host has looked already (pd.ids_needed_known: the host paths check before they upload)
    v.ids_check = (has_id && !pd.id32 && !pd.ids_needed_known && !keep_ids && !tuning_env("RSASA_NO_ID_CHECK")) ? 1u : 0u;
    hs.ids_check = v.ids_check != 0u;
    v.large_sids = reinterpret_cast<const uint32_t *>((const Segment *)W.segments.p + n_seg);
    v.n_large = (uint32_t)n_large;
    v.ids_too_big = ids_too_big ? 1u : 0u;
    v.ids_tables = ctx->ids_unordered_hint ? 1u : 0u;
    v.radius8 = pd.radius8;
    v.radius_table = pd.radius_table;
    v.residue_offsets = bt.residue_offsets;
    v.n_atoms = (uint32_t)N; v.n_structures = (uint32_t)S; v.n_residues = (uint32_t)R;
    v.n_segments = (uint32_t)n_seg;
    v.probe = pd.probe;
    v.segments = (const Segment *)W.segments.p;
    v.acc = (StructAcc *)W.acc.p;
    v.grids = (StructGrid *)W.grids.p;
    v.grid_sums = (GridSums *)W.grid_sums.p;
    v.sid_sorted = (uint32_t *)W.sid_sorted.p;
    v.deferred_list = (uint32_t *)W.deferred_list.p;
    v.claim = (uint32_t *)W.claim.p;
    v.cell_of = (uint32_t *)W.cell_of.p;
    v.rank_of = (uint32_t *)W.rank_of.p;
    v.cells = (uint32_t *)W.cells.p;
    v.cell_capacity = ctx->cell_capacity;
    v.windows = (uint4 *)W.windows.p;
    v.window_capacity = (uint32_t)window_capacity;
    v.scan_block_sums = (uint32_t *)W.scan_sums.p;
    v.sorted_xyzr = (float4 *)W.sorted_xyzr.p;
    v.sorted_orig = (uint32_t *)W.sorted_orig.p;
    v.sorted_id = keep_ids ? (uint64_t *)W.sorted_id.p : nullptr;
    v.sorted_id32 = has_id ? (uint32_t *)W.sorted_id32.p : nullptr;
    v.status = (BatchStatus *)W.status.p;
    v.atom_sasa = bt.out_atom_sasa ? bt.out_atom_sasa : (float *)W.atom_sasa.p;
    v.residue_sasa = (R && bt.residue_offsets) ? bt.out_residue_sasa : nullptr;
    v.neighbor_counts = bt.out_neighbor_counts;

    // Launch stream: grids -> LDS binning -> occlusion of the LDS-binned structures -> (join) ->
    // occlusion of the tail -> sums.  Side stream (forked after the LDS binning): the tail's
    // batch-wide binning, which is bandwidth bound and runs next to the compute-bound occlusion kernel.
    // (RSASA_GRID_CUS experiment: the grid build on the stream of the reserved CUs, the rest behind an event)
    const bool masked = ctx->grid_stream && (pd.stream == ctx->stream || pd.stream == ctx->stream2) && !(ctx->overlap_tail && has_tail);
    hipStream_t gst = masked ? ctx->grid_stream : st;
    if (masked) {
        // the grid stream starts behind everything this batch has queued on its launch stream so far: the upload of the
        // segments above and, on the pipelined host path, the wait for the sub-batch's input copies
        RS_HIP(ctx, hipEventRecord(ctx->ev_grid_in[pd.ws], st));
        RS_HIP(ctx, hipStreamWaitEvent(gst, ctx->ev_grid_in[pd.ws], 0));
    }
    if (ctx->timing) RS_HIP(ctx, hipEventRecord(W.ev[0], gst));
    launch_grid_prepare(v, gst);
    // Two batches in flight: this one's grid build is enqueued beside the other one's occlusion kernel (it gets the CUs
    // when that kernel's workgroups retire: the kernel leaves a CU no room), its occlusion kernel behind it.
    rsasa_context::Workspace &other = ctx->ws[pd.ws ^ 1];
    const bool chain = other.occ_recorded && !tuning_env("RSASA_FREE_OVERLAP");
    const bool overlap = ctx->overlap_tail && has_tail;
    if (overlap && (rc = ensure_side_stream(ctx))) return rc;
    if (overlap) {
        launch_sort_lds(v, st);
        // fork here, not before the LDS binning: two bandwidth-bound phases gain nothing from
        // running side by side, the occlusion kernel (compute bound) hides the tail's binning
        RS_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
        RS_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        launch_sort_tail(v, ctx->side_stream);
        RS_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
        if (ctx->timing) RS_HIP(ctx, hipEventRecord(W.ev[1], st));
        if (chain) RS_HIP(ctx, hipStreamWaitEvent(st, other.ev_occ, 0));
        if (ctx->timing) RS_HIP(ctx, hipEventRecord(W.ev[2], st));
        launch_occlusion(v, lat, ctx->tuning, kOccHead, st);
        RS_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
        launch_occlusion(v, lat, ctx->tuning, kOccRest, st);
    } else {
        launch_sort_lds(v, gst);
        if (has_tail) launch_sort_tail(v, gst);
        if (ctx->timing) RS_HIP(ctx, hipEventRecord(W.ev[1], gst));
        if (masked) {
            RS_HIP(ctx, hipEventRecord(ctx->ev_grid[pd.ws], gst));
            RS_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_grid[pd.ws], 0));
        }
        // (the wait for the other batch's occlusion kernel, the timing event and this batch's "occlusion kernel has ended"
        // event sit around the launch that does the work: OcclusionChain)
        OcclusionChain oc;
        oc.wait = chain ? other.ev_occ : nullptr;
        oc.start = ctx->timing ? W.ev[2] : nullptr;
        oc.done = W.ev_occ;
        oc.expect_ids_dropped = ctx->ids_drop_hint;
        launch_occlusion(v, lat, ctx->tuning, kOccAll, st, &oc);
    }
    if (overlap) RS_HIP(ctx, hipEventRecord(W.ev_occ, st));
    W.occ_recorded = true;
    if (ctx->timing) RS_HIP(ctx, hipEventRecord(W.ev[3], st));
    launch_residue_sums(v, st);
    if (ctx->timing) RS_HIP(ctx, hipEventRecord(W.ev[4], st));
    RS_HIP(ctx, hipMemcpyAsync(hs.h_status, W.status.p, sizeof(BatchStatus),
                               hipMemcpyDeviceToHost, st));
    RS_HIP(ctx, hipGetLastError());
    return RSASA_OK;
}

// Waits for the batch in `pd` (re-running it if the cell array had to grow) and reports its deferred errors.
int wait_one(rsasa_context *ctx, Pending &pd)
{
    if (!pd.active) return RSASA_OK;
    rsasa_context::Workspace &W = ctx->ws[pd.ws];
    for (;;) {
        hipError_t e = hipStreamSynchronize(pd.stream);
        if (e != hipSuccess) {
            pd.active = false;
            return fail(ctx, RSASA_ERR_HIP, "hipStreamSynchronize", e);
        }
        const BatchStatus stt = *ctx->slot[pd.ws].h_status;
        if (stt.grid_too_large) {
            pd.active = false;
            return fail(ctx, RSASA_ERR_GRID_TOO_LARGE,
                        "a structure's cell grid exceeds 2^31 cells (coordinates too sparse)");
        }
        if (stt.bad_input) {
            pd.active = false;
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT,
                        "probe_radius + max radius must be a positive finite number");
        }
        if (!stt.overflow) {
            ctx->tuning.deferred_hint = stt.deferred;  // (sizes the next batch's launch over its deferred list)
            if (ctx->slot[pd.ws].ids_check) {
                ctx->ids_drop_hint = !stt.ids_needed;
                ctx->ids_unordered_hint = stt.ids_unordered != 0;
                if (!stt.ids_needed) ctx->ids_dropped.fetch_add(1, std::memory_order_relaxed);
            }
            if (ctx->timing) {
                float g = 0, o = 0, a = 0, t = 0;
                (void)hipEventElapsedTime(&g, W.ev[0], W.ev[1]);
                (void)hipEventElapsedTime(&o, W.ev[2], W.ev[3]);
                (void)hipEventElapsedTime(&a, W.ev[3], W.ev[4]);
                (void)hipEventElapsedTime(&t, W.ev[0], W.ev[4]);
                ctx->timings = rsasa_timings_t{g, o, a, t, stt.grid_cells, pd.batch.n_atoms, stt.deferred};
                ctx->timings_valid = true;
            }
            pd.active = false;
            return RSASA_OK;
        }
        // the cell array was too small for this batch: grow and run again
        if (stt.total_cells >= 0xFFFFFFF0ull || pd.attempts >= 3) {
            pd.active = false;
            return fail(ctx, RSASA_ERR_GRID_TOO_LARGE, "batch needs more than 2^32 grid cells; split it");
        }
        ctx->cell_capacity = stt.total_cells + stt.total_cells / 8 + 1024;
        pd.attempts++;
        int rc = enqueue_batch(ctx, pd, ctx->slot[pd.ws]);
        if (rc) {
            pd.active = false;
            return rc;
        }
    }
}

// The oldest batch in flight (rsasa_batch_wait), or every one (entry points that need the whole context).
int wait_oldest(rsasa_context *ctx)
{
    if (ctx->n_pending == 0) return RSASA_OK;
    const int rc = wait_one(ctx, ctx->pending[ctx->head]);
    ctx->head ^= 1;
    ctx->n_pending--;
    return rc;
}

int wait_pending(rsasa_context *ctx)
{
    int first = RSASA_OK;
    while (ctx->n_pending) {
        const int rc = wait_one(ctx, ctx->pending[ctx->head]);
        if (rc && !first) first = rc;
        ctx->head ^= 1;
        ctx->n_pending--;
    }
    return first;
}

rsasa_context *g_default_ctx = nullptr;
std::mutex g_default_mu;

int resolve_ctx(rsasa_context *&ctx)
{
    if (ctx) return RSASA_OK;
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default_ctx) {
        int rc = rsasa_context_create(0, &g_default_ctx);
        if (rc) return rc;
    }
    ctx = g_default_ctx;
    return RSASA_OK;
}

}  // namespace

extern "C" {

int rsasa_abi_version(void) { return RSASA_ABI_VERSION; }

const char *rsasa_status_string(int status)
{
    switch (status) {
    case RSASA_OK: return "ok";
    case RSASA_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RSASA_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case RSASA_ERR_HIP: return "HIP runtime error";
    case RSASA_ERR_OUT_OF_MEMORY: return "out of device memory";
    case RSASA_ERR_GRID_TOO_LARGE: return "cell grid too large";
    case RSASA_ERR_INTERNAL: return "internal error";
    default: return "unknown status";
    }
}

int rsasa_device_count(int *out_count)
{
    if (!out_count) return RSASA_ERR_INVALID_ARGUMENT;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out_count = n;
    return RSASA_OK;
}

static int context_create(int device, int own_queues, rsasa_context_t **out_ctx);

int rsasa_context_create(int device, rsasa_context_t **out_ctx)
{
    int own = 0;
    if (const char *v = tuning_env("RSASA_CTX_OWN_QUEUES")) own = std::atoi(v);  // (experiment: DESIGN.md 6)
    return context_create(device, own, out_ctx);
}

static int context_create(int device, int own_queues, rsasa_context_t **out_ctx)
{
    if (!out_ctx) return RSASA_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return RSASA_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return RSASA_ERR_NO_DEVICE;
    rsasa_context *ctx = new (std::nothrow) rsasa_context();
    if (!ctx) return RSASA_ERR_OUT_OF_MEMORY;
    ctx->device = device;
    ctx->own_queues = own_queues;
    ctx->node = device_node_cpus(device);
    DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (const char *v = tuning_env("RSASA_GRID_CUS")) {
        int n_cu = 0;
        if (e == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0 && n_cu <= 512) {
            const int want = std::atoi(v);
            int stride = 1;  // RSASA_GRID_CU_STRIDE: reserved CUs are mask bits 0, stride, 2 stride, ...
            if (const char *sv = tuning_env("RSASA_GRID_CU_STRIDE")) stride = std::max(1, std::atoi(sv));
            if (want > 0 && want * stride <= n_cu && want < n_cu) {
                ctx->grid_cus = (uint32_t)want;
                ctx->cu_mask_words = (uint32_t)(n_cu + 31) / 32;
                for (int c = 0; c < n_cu; c++) ctx->cu_rest[c / 32] |= 1u << (c % 32);
                for (int k = 0; k < want; k++) {
                    const int c = k * stride;
                    ctx->cu_reserved[c / 32] |= 1u << (c % 32);
                    ctx->cu_rest[c / 32] &= ~(1u << (c % 32));
                }
            }
        }
    }
    if (e == hipSuccess && ctx->grid_cus) {
        e = hipExtStreamCreateWithCUMask(&ctx->stream, ctx->cu_mask_words, ctx->cu_rest);
        if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ctx->stream2, ctx->cu_mask_words, ctx->cu_rest);
        if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&ctx->grid_stream, ctx->cu_mask_words, ctx->cu_reserved);
        for (int w = 0; w < 2 && e == hipSuccess; w++) e = hipEventCreateWithFlags(&ctx->ev_grid[w], hipEventDisableTiming);
        for (int w = 0; w < 2 && e == hipSuccess; w++) e = hipEventCreateWithFlags(&ctx->ev_grid_in[w], hipEventDisableTiming);
    } else if (e == hipSuccess) {
        e = new_stream(ctx, &ctx->stream, 2);
        // Experiment (RSASA_GRID_PRIO=1): the grid builds on a stream of the highest priority, so that their workgroups -
        // shaped to fit the slot an occlusion workgroup leaves - are dispatched ahead of the other batch's
        if (e == hipSuccess && tuning_env("RSASA_GRID_PRIO")) {
            int least = 0, greatest = 0;
            e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->grid_stream, hipStreamNonBlocking, greatest);
            for (int w = 0; w < 2 && e == hipSuccess; w++) e = hipEventCreateWithFlags(&ctx->ev_grid[w], hipEventDisableTiming);
            for (int w = 0; w < 2 && e == hipSuccess; w++) e = hipEventCreateWithFlags(&ctx->ev_grid_in[w], hipEventDisableTiming);
        }
    }
    for (int w = 0; w < rsasa_context::kInFlight; w++) {
        for (int i = 0; i < 5 && e == hipSuccess; i++) e = hipEventCreate(&ctx->ws[w].ev[i]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ws[w].ev_occ, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_link, hipEventDisableTiming);
    for (int i = 0; i < rsasa_context::kSlots && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->ev_copy[i], hipEventDisableTiming);
    for (int i = 0; i < rsasa_context::kSlots && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->ev_d2h[i], hipEventDisableTiming);
    for (int i = 0; i < rsasa_context::kSlots && e == hipSuccess; i++) {
        e = hipHostMalloc((void **)&ctx->slot[i].h_status, sizeof(BatchStatus), hipHostMallocDefault);
        if (e == hipSuccess) std::memset(ctx->slot[i].h_status, 0, sizeof(BatchStatus));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        rsasa_context_destroy(ctx);
        return RSASA_ERR_HIP;
    }
    if (const char *v = tuning_env("RSASA_OCCLUSION_KERNEL")) ctx->tuning.kernel_version = std::atoi(v);
    if (const char *v = tuning_env("RSASA_ATOMS_PER_WAVE")) ctx->tuning.atoms_per_wave = (uint32_t)std::atoi(v);
#ifdef RSASA_ABLATE  // timing-ablation builds only (make ablate): the shipped library has no wrong-results switch
    if (const char *v = tuning_env("RSASA_DEBUG_STOP")) ctx->tuning.debug_stop = (uint32_t)std::atoi(v);
#endif
    if (const char *v = tuning_env("RSASA_OVERLAP_TAIL")) ctx->overlap_tail = std::atoi(v) != 0;
    if (const char *v = tuning_env("RSASA_SMALL_PATH")) ctx->small_path = std::atoi(v) != 0;
    *out_ctx = ctx;
    return RSASA_OK;
}

int rsasa_context_destroy(rsasa_context_t *ctx)
{
    if (!ctx) return RSASA_OK;
    if (HostStream *hs = ctx->host_stream) {
        // queued host batches finish (their buffers are the caller's: it has been told to wait for them), then the workers go
        {
            std::unique_lock<std::mutex> lk(hs->mu);
            hs->cv_done.wait(lk, [&] { for (auto &j : hs->jobs) if (!j->done) return false; return true; });
            hs->quit = true;
        }
        hs->cv_work.notify_all();
        for (auto &t : hs->th)
            if (t.joinable()) t.join();
        for (rsasa_context *sc : hs->sub) rsasa_context_destroy(sc);
        delete hs;
        ctx->host_stream = nullptr;
    }
    DeviceGuard guard(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->d2h_stream) (void)hipStreamSynchronize(ctx->d2h_stream);
    for (DeviceBuffer *b : {&ctx->segments, &ctx->acc, &ctx->grids, &ctx->grid_sums, &ctx->sid_sorted, &ctx->deferred_list, &ctx->cell_of,
                            &ctx->rank_of, &ctx->cells, &ctx->windows, &ctx->scan_sums, &ctx->sorted_xyzr,
                            &ctx->sorted_orig, &ctx->sorted_id, &ctx->sorted_id32, &ctx->status, &ctx->atom_sasa, &ctx->claim,
                            &ctx->in_x, &ctx->in_y, &ctx->in_z, &ctx->in_r, &ctx->in_id,
                            &ctx->in_res, &ctx->out_res, &ctx->out_k, &ctx->small_in, &ctx->small_out, &ctx->tr_xyz, &ctx->tr_r,
                            &ctx->tr_id, &ctx->tr_res})
        release(*b);
    for (DeviceBuffer &b : ctx->in_pack) release(b);
    for (auto &m : ctx->more)
        for (DeviceBuffer *b : {&m.x, &m.y, &m.z, &m.r, &m.id, &m.res, &m.atom_sasa, &m.out_res}) release(*b);
    // (the coding pool is the device's, shared by its contexts: it stays)
    if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
    for (auto &kv : ctx->lattices)
        if (kv.second.d) (void)hipFree(kv.second.d);
    for (int i = 0; i < rsasa_context::kSlots; i++) {
        if (ctx->slot[i].h_segments) (void)hipHostFree(ctx->slot[i].h_segments);
        if (ctx->slot[i].h_status) (void)hipHostFree(ctx->slot[i].h_status);
        if (ctx->slot[i].h_res) (void)hipHostFree(ctx->slot[i].h_res);
        if (ctx->ev_done[i]) (void)hipEventDestroy(ctx->ev_done[i]);
    }
    for (int w = 0; w < rsasa_context::kInFlight; w++) {
        for (int i = 0; i < 5; i++)
            if (ctx->ws[w].ev[i]) (void)hipEventDestroy(ctx->ws[w].ev[i]);
        if (ctx->ws[w].ev_occ) (void)hipEventDestroy(ctx->ws[w].ev_occ);
    }
    {
        rsasa_context::Workspace &w1 = ctx->ws[1];
        for (DeviceBuffer *b : {&w1.segments, &w1.acc, &w1.grids, &w1.grid_sums, &w1.sid_sorted, &w1.deferred_list, &w1.cell_of, &w1.rank_of,
                                &w1.cells, &w1.windows, &w1.scan_sums, &w1.sorted_xyzr, &w1.sorted_orig, &w1.sorted_id, &w1.sorted_id32,
                                &w1.status, &w1.atom_sasa, &w1.claim})
            release(*b);
    }
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->grid_stream) (void)hipStreamDestroy(ctx->grid_stream);
    for (hipEvent_t ev : ctx->ev_grid)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->ev_grid_in)
        if (ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < rsasa_context::kSlots; i++)
        if (ctx->ev_copy[i]) (void)hipEventDestroy(ctx->ev_copy[i]);
    for (int i = 0; i < rsasa_context::kSlots; i++) {
        if (ctx->ev_d2h[i]) (void)hipEventDestroy(ctx->ev_d2h[i]);
        if (ctx->h_out[i]) (void)hipHostFree(ctx->h_out[i]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->device >= 0 && ctx->device < 64) {
        std::lock_guard<std::mutex> lk(g_link[ctx->device].mu);
        if (g_link[ctx->device].owner == ctx) { g_link[ctx->device].last = nullptr; g_link[ctx->device].owner = nullptr; }
    }
    if (ctx->ev_link) (void)hipEventDestroy(ctx->ev_link);
    for (auto &row : ctx->tr_ev)
        for (hipEvent_t e : row)
            if (e) (void)hipEventDestroy(e);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RSASA_OK;
}

const char *rsasa_context_last_error(const rsasa_context_t *ctx)
{
    if (!ctx) return "";
    // a copy per calling thread, taken under the lock: another thread's failure cannot change
    // (or free) the string while this one reads it; valid until this thread's next call
    static thread_local std::string copy;
    rsasa_context *c = const_cast<rsasa_context *>(ctx);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    copy = c->last_error;
    return copy.c_str();
}

int rsasa_context_get_device(const rsasa_context_t *ctx, int *out_device)
{
    if (!ctx || !out_device) return RSASA_ERR_INVALID_ARGUMENT;
    *out_device = ctx->device;
    return RSASA_OK;
}

int rsasa_context_set_simd_width(rsasa_context_t *ctx, int w)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (w != 1 && w != 4 && w != 8 && w != 16)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "simd_width must be 1, 4, 8 or 16");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->simd_width = w;
    return RSASA_OK;
}

int rsasa_context_get_simd_width(rsasa_context_t *ctx, int *out_simd_width)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (!out_simd_width) return RSASA_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    *out_simd_width = ctx->simd_width;
    return RSASA_OK;
}

int rsasa_context_bind_thread(rsasa_context_t *ctx, int *out_numa_node)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (out_numa_node) *out_numa_node = ctx->node.valid ? ctx->node.node : -1;
    (void)bind_thread_to(pthread_self(), ctx->node);
    return RSASA_OK;
}

int rsasa_context_enable_timing(rsasa_context_t *ctx, int enable)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    // a batch enqueued under the other setting has (or lacks) its events: finish what is in flight first
    if (ctx->n_pending && (ctx->timing != (enable != 0))) {
        RS_DEVICE(ctx);
        if ((rc = wait_pending(ctx))) return rc;
    }
    ctx->timing = enable != 0;
    ctx->timings_valid = false;
    return RSASA_OK;
}

int rsasa_context_get_timings(rsasa_context_t *ctx, rsasa_timings_t *out)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (!out) return RSASA_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ctx->timings_valid)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "no timed batch has completed");
    *out = ctx->timings;
    return RSASA_OK;
}

int rsasa_context_ids_dropped(rsasa_context_t *ctx, uint64_t *out_batches)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (!out_batches) return RSASA_ERR_INVALID_ARGUMENT;
    uint64_t n = ctx->ids_dropped.load(std::memory_order_relaxed);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (ctx->host_stream)  // (the stream's batches run on its workers' contexts)
        for (int w = 0; w < ctx->host_stream->n_workers; w++)
            if (ctx->host_stream->sub[w]) n += ctx->host_stream->sub[w]->ids_dropped.load(std::memory_order_relaxed);
    *out_batches = n;
    return RSASA_OK;
}

int rsasa_batch_enqueue(rsasa_context_t *ctx, const rsasa_device_batch_t *batch,
                        float probe_radius, size_t n_points, void *hip_stream)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!batch) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "batch is NULL");
    if (n_points == 0 || n_points > (1u << 24))
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "n_points must be in [1, 2^24]");
    if (!(probe_radius >= 0.0f) || !std::isfinite(probe_radius))
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "probe_radius must be finite and >= 0");
    if (batch->n_atoms >= 0xFFFFFFF0ull || batch->n_structures >= 0xFFFFFFF0ull ||
        batch->n_residues >= 0xFFFFFFF0ull)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "batch too large for 32-bit indices");
    if (batch->n_atoms && (!batch->x || !batch->y || !batch->z || !batch->radius))
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "coordinate / radius arrays are NULL");
    if (batch->n_structures && !batch->structure_offsets_host)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "structure_offsets_host is NULL");
    if (batch->n_structures) {
        const uint32_t *o = batch->structure_offsets_host;
        if (o[0] != 0 || o[batch->n_structures] != batch->n_atoms)
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "structure_offsets must span [0, n_atoms]");
        for (size_t s = 0; s < batch->n_structures; s++)
            if (o[s] > o[s + 1])
                return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "structure_offsets must be non-decreasing");
    } else if (batch->n_atoms) {
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "atoms without structures");
    }
    if (batch->n_residues && batch->residue_offsets && !batch->out_residue_sasa)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "out_residue_sasa is NULL");

    RS_DEVICE(ctx);
    // Up to two batches in flight, each in its own workspace; a third waits for the oldest one.
    while (ctx->n_pending >= rsasa_context::kInFlight)
        if ((rc = wait_oldest(ctx))) return rc;
    const int w = ctx->n_pending ? ctx->pending[ctx->head].ws ^ 1 : 0;
    if (w == 1 && !hip_stream && !ctx->stream2)
        RS_HIP(ctx, new_stream(ctx, &ctx->stream2, 2));
    Pending &pd = ctx->pending[ctx->head ^ (ctx->n_pending ? 1 : 0)];
    pd = Pending{};
    pd.batch = *batch;
    pd.probe = probe_radius;
    pd.n_points = n_points;
    pd.stream = hip_stream ? (hipStream_t)hip_stream : (w ? ctx->stream2 : ctx->stream);
    pd.ws = w;
    rc = enqueue_batch(ctx, pd, ctx->slot[w]);
    pd.active = (rc == RSASA_OK);
    if (pd.active) ctx->n_pending++;
    return rc;
}

int rsasa_batch_wait(rsasa_context_t *ctx)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    RS_DEVICE(ctx);
    return wait_oldest(ctx);
}

int rsasa_batch_wait_all(rsasa_context_t *ctx)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    RS_DEVICE(ctx);
    return wait_pending(ctx);
}

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
constexpr int kNotSmall = 1;  // (positive: not an error) the batch goes through the general path

// Batches of a few structures handed over in host memory - the literal drop-in use, one
// calculate_sasa_internal call per structure - are latency bound: ~17 kernel launches and half a
// dozen small copies.  Here the host computes the bounding boxes and grids itself (N is small),
// so the device needs ONE upload (inputs + grids + status, through pinned staging), four
// launches (LDS binning, the two occlusion kernels, residue sums) and one download.
int run_small_host_batch(rsasa_context *ctx, const float *x, const float *y, const float *z,
                         const float *radius, const uint64_t *id, const uint32_t *so, size_t S,
                         float probe, size_t n_points, float *out_atom, const uint32_t *ro, size_t R,
                         float *out_res)
{
    // anything unusual is left to the general path, which validates and reports it
    if (S == 0 || S > kSmallStructures || so[0] != 0 || n_points == 0 || n_points > (1u << 24) ||
        !(probe >= 0.0f) || !std::isfinite(probe) || ctx->timing || ctx->tuning.debug_stop)
        return kNotSmall;
    const size_t N = so[S];
    if (N == 0 || N > kSmallAtoms) return kNotSmall;
    std::vector<StructGrid> grids(S);
    std::vector<uint4> windows;  // work list of k_sort_window (the general path builds it on the device)
    unsigned long long total_cells = 0;  // 16-bit entries of the cell array
    for (size_t s = 0; s < S; s++) {
        if (so[s] > so[s + 1]) return kNotSmall;  // (the general path reports it)
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mr = 0.0f;
        bool finite = true, odd_r = false;
        for (uint32_t i = so[s]; i < so[s + 1]; i++) {
            const float p[3] = {x[i], y[i], z[i]};
            for (int k = 0; k < 3; k++) {
                mn[k] = fminf(mn[k], p[k]);
                mx[k] = fmaxf(mx[k], p[k]);
                finite &= std::isfinite(p[k]);
            }
            mr = fmaxf(mr, radius[i]);
            finite &= std::isfinite(radius[i]);
            odd_r |= !(radius[i] >= 0.0f && radius[i] <= 64.0f) || !(fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2])) <= 1e8f);
        }
        if (!finite || !small_grid(mn, mx, mr, probe, so[s + 1] - so[s], &grids[s])) return kNotSmall;
        grids[s].atom_begin = so[s];
        grids[s].odd_radii = (odd_r ? 1u : 0u) | (grid_group_shift(grids[s].n_atoms, grids[s].n_cells) << 8);
        grids[s].sorted_base = so[s];
        grids[s].cell_base = (uint32_t)total_cells;
        total_cells += lds_cell_slots(grids[s].n_cells);
        for (uint32_t w = 0; w < grid_windows(grids[s].n_cells); w++) windows.push_back(make_uint4((uint32_t)s, w, grids[s].atom_begin, grids[s].n_atoms));
    }
    const unsigned long long tail_begin = (total_cells / 2ull + 1023ull) & ~1023ull;
    const size_t W = windows.size();

    Lattice lat;
    int rc = get_lattice(ctx, n_points, &lat);
    if (rc) return rc;
    // staging layout (16-byte aligned sections): status | grids | windows | x | y | z | r | id | residue offsets
    auto up = [](size_t v) { return (v + 15) & ~size_t(15); };
    const size_t o_grid = 64, o_win = o_grid + up(S * sizeof(StructGrid)), o_x = o_win + up(W * sizeof(uint4)), o_y = o_x + up(N * 4), o_z = o_y + up(N * 4),
                 o_r = o_z + up(N * 4), o_id = o_r + up(N * 4), o_res = o_id + (id ? up(N * 8) : 0),
                 in_bytes = o_res + (R ? up((R + 1) * 4) : 0);
    const size_t o_oa = 0, o_or = up(N * 4), out_bytes = o_or + up(R * 4);
    const size_t host_bytes = in_bytes + out_bytes;
    if (host_bytes > ctx->h_small_cap) {
        if (ctx->h_small) {
            RS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            RS_HIP(ctx, hipHostFree(ctx->h_small));
            ctx->h_small = nullptr;
            ctx->h_small_cap = 0;
        }
        RS_HIP(ctx, hipHostMalloc(&ctx->h_small, host_bytes * 2, hipHostMallocDefault));
        ctx->h_small_cap = host_bytes * 2;
    }
    if ((rc = reserve(ctx, ctx->small_in, in_bytes))) return rc;
    if ((rc = reserve(ctx, ctx->small_out, out_bytes))) return rc;
    if ((rc = reserve(ctx, ctx->sid_sorted, N * 4))) return rc;
    if ((rc = reserve(ctx, ctx->deferred_list, N * 4))) return rc;
    if ((rc = reserve(ctx, ctx->claim, kClaimBytes))) return rc;
    if ((rc = reserve(ctx, ctx->rank_of, N * 4))) return rc;
    if ((rc = reserve(ctx, ctx->cells, (size_t)(tail_begin + 8) * 4))) return rc;
    if ((rc = reserve(ctx, ctx->sorted_xyzr, N * 16))) return rc;
    if ((rc = reserve(ctx, ctx->sorted_orig, N * 4))) return rc;
    const bool keep_ids = id && !occlusion_uses_mx(ctx->tuning, lat, (uint32_t)N);
    if (keep_ids && (rc = reserve(ctx, ctx->sorted_id, N * 8))) return rc;
    if (id && (rc = reserve(ctx, ctx->sorted_id32, N * 4))) return rc;

    char *h = (char *)ctx->h_small;
    BatchStatus stt{};
    stt.total_cells = tail_begin;
    stt.tail_cell_begin = tail_begin;
    stt.tail_atom_base = (uint32_t)N;
    stt.n_windows = (uint32_t)W;
    std::memcpy(h, &stt, sizeof stt);
    std::memcpy(h + o_grid, grids.data(), S * sizeof(StructGrid));
    if (W) std::memcpy(h + o_win, windows.data(), W * sizeof(uint4));
    std::memcpy(h + o_x, x, N * 4);
    std::memcpy(h + o_y, y, N * 4);
    std::memcpy(h + o_z, z, N * 4);
    std::memcpy(h + o_r, radius, N * 4);
    if (id) std::memcpy(h + o_id, id, N * 8);
    if (R) std::memcpy(h + o_res, ro, (R + 1) * 4);
    hipStream_t st = ctx->stream;
    char *d = (char *)ctx->small_in.p, *dout = (char *)ctx->small_out.p;
    // One structure of a few thousand atoms - the per-structure call: no upload at all.  The binning
    // kernel gets grid and status as kernel arguments and reads the atoms from the pinned staging
    // block (they cross the link once or twice; an upload costs 15 us before the first kernel starts).
    const bool single = S == 1 && N <= kSingleAtoms;
    if (!single) RS_HIP(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    const char *src = single ? h : d;

    BatchView v{};
    v.x = (const float *)(src + o_x); v.y = (const float *)(src + o_y); v.z = (const float *)(src + o_z);
    v.radius = (const float *)(src + o_r);
    v.id = id ? (const uint64_t *)(src + o_id) : nullptr;
    v.residue_offsets = R ? (const uint32_t *)(src + o_res) : nullptr;
    v.n_atoms = (uint32_t)N; v.n_structures = (uint32_t)S; v.n_residues = (uint32_t)R;
    v.probe = probe;
    v.grids = (StructGrid *)(d + o_grid);
    v.status = (BatchStatus *)d;
    v.sid_sorted = (uint32_t *)ctx->sid_sorted.p;
    v.deferred_list = (uint32_t *)ctx->deferred_list.p;
    v.claim = (uint32_t *)ctx->claim.p;
    v.cell_of = (uint32_t *)ctx->cell_of.p;
    v.rank_of = (uint32_t *)ctx->rank_of.p;
    v.cells = (uint32_t *)ctx->cells.p;
    v.cell_capacity = tail_begin + 8;
    v.windows = (uint4 *)(d + o_win);
    v.window_capacity = (uint32_t)W;
    v.sorted_xyzr = (float4 *)ctx->sorted_xyzr.p;
    v.sorted_orig = (uint32_t *)ctx->sorted_orig.p;
    v.sorted_id = keep_ids ? (uint64_t *)ctx->sorted_id.p : nullptr;
    v.sorted_id32 = id ? (uint32_t *)ctx->sorted_id32.p : nullptr;
    // (the single-structure call also gets its results written straight into the pinned block)
    char *hout = h + in_bytes;
    char *outp = single ? hout : dout;
    v.atom_sasa = (float *)(outp + o_oa);
    v.residue_sasa = R ? (float *)(outp + o_or) : nullptr;
    // (and the general kernel is only launched if the straight-line one says it left atoms to it:
    // a word of the pinned block's header, looked at after the stream has drained)
    uint32_t *flag = reinterpret_cast<uint32_t *>(h + 56);
    *flag = 0u;
    v.defer_flag = single ? flag : nullptr;
    if (single) launch_sort_single(v, SingleJob{grids[0], stt}, st);
    else launch_sort_lds(v, st);
    launch_occlusion(v, lat, ctx->tuning, kOccAll, st);
    launch_residue_sums(v, st);
    if (!single) RS_HIP(ctx, hipMemcpyAsync(hout, dout, out_bytes, hipMemcpyDeviceToHost, st));
    RS_HIP(ctx, hipGetLastError());
    RS_HIP(ctx, hipStreamSynchronize(st));
    if (single && *flag) {
        launch_occlusion_deferred(v, lat, st);
        launch_residue_sums(v, st);
        RS_HIP(ctx, hipGetLastError());
        RS_HIP(ctx, hipStreamSynchronize(st));
    }
    if (out_atom) std::memcpy(out_atom, hout + o_oa, N * 4);
    if (R) std::memcpy(out_res, hout + o_or, R * 4);
    return RSASA_OK;
}

}  // namespace

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
    static const bool h2h_trace = tuning_env("RSASA_H2H_TRACE") != nullptr;  // host-side phases of a pipelined call, to stderr
    const auto tr_t0 = std::chrono::steady_clock::now();
    // (calls of several contexts on one clock; with the trace on, a reference event ties the device's clock to it)
    static const auto epoch = std::chrono::steady_clock::now();
    static hipEvent_t tr_ref = nullptr;
    static double tr_ref_host_us = 0;
    static std::mutex tr_mu;
    if (h2h_trace) {
        std::lock_guard<std::mutex> lkt(tr_mu);
        if (!tr_ref && hipEventCreate(&tr_ref) == hipSuccess) {
            (void)hipEventRecord(tr_ref, ctx->stream);
            (void)hipEventSynchronize(tr_ref);
            tr_ref_host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - epoch).count();
        }
        for (auto &row : ctx->tr_ev)
            for (hipEvent_t &e : row)
                if (!e) (void)hipEventCreate(&e);
    }
    auto tr_rec = [&](int k, int i, hipStream_t s) {
        if (h2h_trace && ctx->tr_ev[k][i]) (void)hipEventRecord(ctx->tr_ev[k][i], s);
    };
    auto tr = [&](const char *what) {
        if (h2h_trace) {
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "h2h ctx %p at %9.1f us, %8.1f us into the call: %s\n", (void *)ctx,
                         std::chrono::duration<double, std::micro>(now - epoch).count(),
                         std::chrono::duration<double, std::micro>(now - tr_t0).count(), what);
        }
    };
    if (ctx->n_pending && (rc = wait_pending(ctx))) return rc;
    if (ctx->small_path) {
        rc = run_small_host_batch(ctx, x, y, z, radius, id, structure_offsets, n_structures, probe_radius, n_points,
                                  out_atom_sasa, want_res ? residue_offsets : nullptr, want_res ? n_residues : 0,
                                  out_residue_sasa);
        if (rc != kNotSmall) return rc;
    }

    if ((rc = ensure_copy_streams(ctx))) return rc;
    // Large batches are cut into sub-batches of whole structures (and whole residues) whose
    // host-to-device copies run on a second stream into a second set of input buffers while the
    // previous sub-batch computes: the PCIe transfer hides behind the kernels.
    size_t kSubAtoms = 1500000;  // smallest sub-batch worth its own launch sequence
    if (const char *v = tuning_env("RSASA_SUB_ATOMS")) kSubAtoms = (size_t)std::max(100000, std::atoi(v));
    std::vector<size_t> cut{0};        // structure indices where sub-batches begin / end
    if (N >= 2 * kSubAtoms && n_structures > 1) {
        // (a worker of a stream of host batches: the NEXT call's uploads hide a call's fill and drain, so two sub-batches
        // - one upload running beside one half's kernels - are enough, and every sub-batch fewer is a grid build fewer
        // between the occlusion kernels: HostStream)
        size_t max_sub = ctx->stream_sub_batches ? ctx->stream_sub_batches : 8;
        if (const char *v = tuning_env("RSASA_SUB_BATCHES")) max_sub = (size_t)std::max(2, std::atoi(v));
        const size_t n_sub = std::min<size_t>(max_sub, N / kSubAtoms);
        // The link is the longest leg.  The first sub-batch's upload is not hidden behind anything, and nothing hides
        // the last one's kernels: each gets half a share (RSASA_H2H_TAIL=0: only the first).
        static const bool half_tail = !(tuning_env("RSASA_H2H_TAIL") && std::atoi(tuning_env("RSASA_H2H_TAIL")) == 0);
        const size_t first = half_tail ? N / (2 * n_sub - 2) : N / (2 * n_sub - 1);
        const size_t share = half_tail ? 2 * first : (N - first) / (n_sub - 1);
        auto boundary = [&](size_t k) { return first + (k - 1) * share; };  // first atom of sub-batch k >= 1
        size_t next = 1;
        for (size_t sidx = 1; sidx < n_structures && next < n_sub; sidx++) {
            const size_t a0 = structure_offsets[sidx];
            if (a0 < boundary(next)) continue;
            if (want_res && !std::binary_search(residue_offsets, residue_offsets + n_residues + 1, (uint32_t)a0))
                continue;  // a residue spans this structure boundary: cut later
            cut.push_back(sidx);
            while (next < n_sub && boundary(next) <= a0) next++;
        }
    }
    cut.push_back(n_structures);
    size_t max_atoms = 1, max_res = 1;
    std::vector<size_t> res_cut(cut.size(), 0);
    for (size_t c = 0; c + 1 < cut.size(); c++) {
        max_atoms = std::max<size_t>(max_atoms, structure_offsets[cut[c + 1]] - structure_offsets[cut[c]]);
        if (want_res) {
            res_cut[c + 1] = c + 2 == cut.size()
                                 ? n_residues
                                 : (size_t)(std::lower_bound(residue_offsets, residue_offsets + n_residues + 1,
                                                             structure_offsets[cut[c + 1]]) - residue_offsets);
            max_res = std::max(max_res, res_cut[c + 1] - res_cut[c]);
        }
    }
    const bool piped = cut.size() > 2;
    const size_t n_sub = cut.size() - 1;
    constexpr size_t kTableWords = 256;  // a sub-batch's offsets block on the device: radius table | residue offsets
    constexpr int kSlots = rsasa_context::kSlots;
    const int n_slots = piped ? (int)std::min<size_t>((size_t)kSlots, n_sub) : 1;
    DeviceBuffer *bx[kSlots] = {&ctx->in_x}, *by[kSlots] = {&ctx->in_y}, *bz[kSlots] = {&ctx->in_z}, *br[kSlots] = {&ctx->in_r};
    DeviceBuffer *bi[kSlots] = {&ctx->in_id}, *bo[kSlots] = {&ctx->in_res}, *oa[kSlots] = {&ctx->atom_sasa}, *orr[kSlots] = {&ctx->out_res};
    for (int k = 1; k < kSlots; k++) {
        rsasa_context::MoreSlot &m = ctx->more[k - 1];
        bx[k] = &m.x; by[k] = &m.y; bz[k] = &m.z; br[k] = &m.r; bi[k] = &m.id; bo[k] = &m.res; oa[k] = &m.atom_sasa; orr[k] = &m.out_res;
    }
    const float *dev_x[kSlots] = {}, *dev_y[kSlots] = {}, *dev_z[kSlots] = {};
    for (int k = 0; k < n_slots; k++) {
        if ((rc = reserve(ctx, *bx[k], max_atoms * 4))) return rc;
        if ((rc = reserve(ctx, *by[k], max_atoms * 4))) return rc;
        if ((rc = reserve(ctx, *bz[k], max_atoms * 4))) return rc;
        dev_x[k] = (const float *)bx[k]->p; dev_y[k] = (const float *)by[k]->p; dev_z[k] = (const float *)bz[k]->p;
        if ((rc = reserve(ctx, *br[k], max_atoms * 4))) return rc;
        if (id && !piped && (rc = reserve(ctx, *bi[k], max_atoms * 8))) return rc;  // (pipelined: below, unless the ids are folded)
        if (want_res && !piped && (rc = reserve(ctx, *bo[k], (max_res + 1) * 4))) return rc;
        if ((rc = reserve(ctx, *oa[k], max_atoms * 4))) return rc;
        if (want_res && (rc = reserve(ctx, *orr[k], max_res * 4))) return rc;
    }

    // Ids on the pipelined path: the link is the longest leg, and the matrix-core kernel only looks at 32-bit folds
    // of the ids.  With the caller's ids in pinned memory the host folds them (a few worker threads, one sub-batch
    // ahead of the uploads) and 4 bytes per atom cross the link instead of 8; the general kernel reads the few full
    // ids it needs (atoms whose folds collide) straight from the caller's array, mapped into the device's
    // address space.  Pageable ids, or a sub-batch the per-atom kernels take: the 64-bit ids are uploaded.
    const uint64_t *id_mapped = nullptr;
    bool fold_ids = false;
    if (piped && id && !tuning_env("RSASA_NO_ID_FOLD")) {
        Lattice lat_probe;
        void *dp = nullptr;
        if (n_points >= 1 && n_points <= (1u << 24) && get_lattice(ctx, n_points, &lat_probe) == RSASA_OK &&
            hipHostGetDevicePointer(&dp, const_cast<uint64_t *>(id), 0) == hipSuccess && dp) {
            fold_ids = true;
            for (size_t c = 0; c + 1 < cut.size(); c++)
                fold_ids &= occlusion_uses_mx(ctx->tuning, lat_probe,
                                              (uint32_t)(structure_offsets[cut[c + 1]] - structure_offsets[cut[c]]));
            id_mapped = (const uint64_t *)dp;
        } else {
            (void)hipGetLastError();
        }
    }
    // Radii on the pipelined path: one-byte codes into the table of the batch's distinct radii (RadiusCodec), coded by
    // the same worker threads.
    const bool code_radii = piped && !tuning_env("RSASA_NO_RADIUS_CODES");
    // the sub-batches' pinned blocks: radius table | residue offsets | folded ids | radius codes, 16-byte aligned parts
    struct Pack { size_t base = 0, o_res = 0, o_id = 0, o_r8 = 0, bytes = 0; };
    std::vector<Pack> pack(cut.size());
    if (piped) {
        auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
        size_t total = 0, largest = 0;
        for (size_t c = 0; c + 1 < cut.size(); c++) {
            const size_t na = structure_offsets[cut[c + 1]] - structure_offsets[cut[c]];
            const size_t nr = want_res ? res_cut[c + 1] - res_cut[c] : 0;
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
    const bool check_ids = id && !tuning_env("RSASA_NO_ID_CHECK");
    bool host_check = false;
    if (!piped && check_ids && cut.size() == 2 && structure_offsets[n_structures] >= 262144u && n_points >= 1 && n_points <= (1u << 24)) {
        Lattice lat_probe;
        host_check = get_lattice(ctx, n_points, &lat_probe) == RSASA_OK &&
                     occlusion_uses_mx(ctx->tuning, lat_probe, structure_offsets[n_structures]);
    }
    if ((fold_ids || code_radii || host_check) && !ctx->fold_pool) {
        unsigned nt = std::thread::hardware_concurrency() / 4;
        if (const char *v = tuning_env("RSASA_FOLD_THREADS")) nt = (unsigned)std::atoi(v);
        // ONE pool per device for all its contexts: two contexts with a stream of host batches between them (or
        // process_files' pair) would otherwise run two pools of sixteen threads against each other - under a CPU quota
        // (the measurement boxes: 16 CPUs) both are throttled, a sub-batch's coding takes 8 ms instead of 1 and its upload
        // waits for it.  Jobs are worked off in the order they were submitted, whoever submitted them.
        static std::mutex pools_mu;
        static FoldPool *pools[64] = {};
        {
            std::lock_guard<std::mutex> lkp(pools_mu);
            const int d = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
            if (!pools[d]) pools[d] = new (std::nothrow) FoldPool(std::min(16u, std::max(2u, nt)), ctx->node);  // (lives as long as the process)
            ctx->fold_pool = pools[d];
        }
        if (!ctx->fold_pool) return fail(ctx, RSASA_ERR_OUT_OF_MEMORY, "fold pool");
    }
    std::vector<unsigned long long> fold_job(cut.size(), 0);
    // (no fold job may outlive this call: the workers read the caller's id array)
    struct FoldDrain {
        FoldPool *pool = nullptr;
        unsigned long long last = 0;
        ~FoldDrain() { if (pool) pool->wait(last); }
    } fold_drain;

    // Results leave on their own stream while the next sub-batch computes.  A destination in
    // pinned (page-locked) host memory takes the copy directly; a pageable one gets it through
    // pinned staging, moved to its place by this thread once the copy has landed.
    auto is_pinned = [](const void *p) {
        hipPointerAttribute_t at{};
        if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return at.type == hipMemoryTypeHost;
    };
    const bool atoms_direct = !out_atom_sasa || is_pinned(out_atom_sasa);
    const bool res_direct = !want_res || is_pinned(out_residue_sasa);
    const size_t stage_atoms = (out_atom_sasa && !atoms_direct) ? max_atoms * 4 : 0;
    const size_t stage_bytes = stage_atoms + ((want_res && !res_direct) ? max_res * 4 : 0);
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

    // host copies of the rebased offsets stay alive until their sub-batch has been waited for
    std::vector<uint32_t> so[kSlots];
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
    tr("setup done");
    std::vector<char> use_codes(cut.size(), 0);  // sub-batch c's radii travel as codes (decided once its coding job is done)
    // sub-batch c's ids stay on the host: those of each of its structures increase strictly, so they are all different and
    // change nothing (IdOrder; found by the workers that fold them)
    std::vector<char> drop_ids(cut.size(), 0);
    std::unique_ptr<std::atomic<int>[]> ids_matter(new (std::nothrow) std::atomic<int>[cut.size()]);
    if (!ids_matter) return fail(ctx, RSASA_ERR_OUT_OF_MEMORY, "id flags");
    // the coordinates of a sub-batch do not wait for its coding job (ids, radius codes): they are queued first, and the
    // first sub-batch's start crossing the link while its block is still being written
    auto upload_xyz = [&](size_t c, hipStream_t st) -> int {
        const int k = (int)(c % kSlots);
        const size_t s0 = cut[c], s1 = cut[c + 1], a0 = structure_offsets[s0], na = structure_offsets[s1] - a0;
        if (na) {
            // (one hipMemcpy2DAsync of three rows for x, y, z a fixed distance apart runs at the link's rate by itself -
            // tools/microbench_copy2d.hip - but is a kernel: behind the occlusion kernels it waits for CUs, and the call
            // took 6.5 instead of 5.1 ms)
            RS_HIP(ctx, hipMemcpyAsync(bx[k]->p, x + a0, na * 4, hipMemcpyHostToDevice, st));
            RS_HIP(ctx, hipMemcpyAsync(by[k]->p, y + a0, na * 4, hipMemcpyHostToDevice, st));
            RS_HIP(ctx, hipMemcpyAsync(bz[k]->p, z + a0, na * 4, hipMemcpyHostToDevice, st));
        }
        return RSASA_OK;
    };
    auto upload = [&](size_t c, hipStream_t st) -> int {
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
    };
    auto enqueue = [&](size_t c) -> int {
        const int k = (int)(c % kSlots);
        const size_t s0 = cut[c], s1 = cut[c + 1], na = structure_offsets[s1] - structure_offsets[s0];
        const size_t nr = want_res ? res_cut[c + 1] - res_cut[c] : 0;
        rsasa_device_batch_t bt{};
        bt.x = dev_x[k];
        bt.y = dev_y[k];
        bt.z = dev_z[k];
        bt.radius = (const float *)br[k]->p;
        bt.id = id && !drop_ids[c] ? (const uint64_t *)bi[k]->p : nullptr;
        bt.structure_offsets_host = so[k].data();
        bt.n_structures = s1 - s0;
        bt.n_atoms = na;
        bt.residue_offsets = nr ? (const uint32_t *)bo[k]->p : nullptr;
        bt.n_residues = nr;
        bt.out_atom_sasa = (float *)oa[k]->p;
        bt.out_residue_sasa = nr ? (float *)orr[k]->p : nullptr;
        bt.out_neighbor_counts = nullptr;
        if (!(na || nr)) return RSASA_OK;
        return rsasa_batch_enqueue(ctx, &bt, probe_radius, n_points, nullptr);
    };
    // staged results of output slot k that still have to be moved to the caller's arrays
    struct Staged { bool active = false; size_t a0 = 0, na = 0, r0 = 0, nr = 0; } staged[kSlots];
    auto drain = [&](int k) -> int {
        if (!staged[k].active) return RSASA_OK;
        RS_HIP(ctx, hipEventSynchronize(ctx->ev_d2h[k]));
        const char *h = (const char *)ctx->h_out[k];
        if (out_atom_sasa && !atoms_direct && staged[k].na)
            std::memcpy(out_atom_sasa + staged[k].a0, h, staged[k].na * 4);
        if (want_res && !res_direct && staged[k].nr)
            std::memcpy(out_residue_sasa + staged[k].r0, h + stage_atoms, staged[k].nr * 4);
        staged[k].active = false;
        return RSASA_OK;
    };

    hipStream_t st = ctx->stream, dn = ctx->d2h_stream;
    auto copy_out = [&](size_t c) -> int {  // sub-batch c's results (all its kernels have been waited for or ordered before)
        const int k = (int)(c % kSlots);
        const size_t a0 = structure_offsets[cut[c]], na = structure_offsets[cut[c + 1]] - a0;
        const size_t r0 = res_cut[c], nr = want_res ? res_cut[c + 1] - r0 : 0;
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
    };
    if (!piped) {
        // one sub-batch: upload, kernels, wait (re-runs with a larger cell array if needed), copy out
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
        if ((rc = enqueue(0))) return rc;
        const size_t na = structure_offsets[cut[1]], nr = want_res ? res_cut[1] : 0;
        if ((na || nr) && (rc = rsasa_batch_wait(ctx))) return rc;
        if ((rc = copy_out(0))) return rc;
        if ((rc = drain(0))) return rc;
        RS_HIP(ctx, hipStreamSynchronize(dn));
        return RSASA_OK;
    }

    // Several sub-batches on three streams: copy-in (sub-batch c + 1), compute (c), copy-out (c - 1).
    // kSlots sub-batches are in flight: the host queues the next one (upload, then kernels behind the
    // upload's event) while earlier ones compute and never waits in between - the uploads, which are the
    // longest leg (PCIe), follow each other without a gap.  A
    // sub-batch's status block (host slot c % kSlots) is only read when its slot is needed again or at
    // the end; if one of them reports that the cell array was too small, everything is drained, the
    // array grows to the largest size reported and the call starts over (outputs are simply
    // written again) - that happens on a context's first large call at most.
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
        uint64_t need_cells = 0;
        int err = RSASA_OK;
        LinkHold turn;  // (released without an event on an error return)
        auto check = [&](int k) {  // status of the sub-batch that used host slot k (its event has been waited for)
            const BatchStatus stt = *ctx->slot[k].h_status;
            if (stt.grid_too_large && !err)
                err = fail(ctx, RSASA_ERR_GRID_TOO_LARGE, "a structure's cell grid exceeds 2^31 cells (coordinates too sparse)");
            if (stt.bad_input && !err)
                err = fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "probe_radius + max radius must be a positive finite number");
            if (stt.overflow) need_cells = std::max<uint64_t>(need_cells, stt.total_cells);
            else ctx->tuning.deferred_hint = stt.deferred;
            if (!stt.overflow && ctx->slot[k].ids_check) {
                ctx->ids_drop_hint = !stt.ids_needed;
                ctx->ids_unordered_hint = stt.ids_unordered != 0;
                if (!stt.ids_needed) ctx->ids_dropped.fetch_add(1, std::memory_order_relaxed);
            }
        };
        bool used[kSlots] = {};
        if (fold_ids || code_radii) {
            // all sub-batches' folds and radius codes, in order, while the uploads follow behind (the previous
            // attempt's copies out of the pinned blocks have all been waited for)
            if (code_radii) ctx->radius_codec.reset();
            for (size_t c = 0; c < n_sub; c++) {
                const size_t a0 = structure_offsets[cut[c]], na = structure_offsets[cut[c + 1]] - a0;
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
        // (the coding jobs above run while this call waits for its turn on the link)
        RS_HIP(ctx, turn.take(ctx, cp));
        tr("turn on the link taken");
        for (size_t c = 0; c < n_sub; c++) {
            const int k = (int)(c % kSlots);
            if (used[k]) {
                // slot k (host segments / status, input and output buffers) was sub-batch c - kSlots's
                RS_HIP(ctx, hipEventSynchronize(ctx->ev_done[k]));
                check(k);
                if ((rc = drain(k))) return rc;  // its staged results, if the destination is pageable
            }
            tr_rec(k, 0, cp);
            if ((rc = upload_xyz(c, cp))) return rc;
            if (fold_ids || code_radii) ctx->fold_pool->wait(fold_job[c]);
            use_codes[c] = code_radii && !ctx->radius_codec.failed.load();
            drop_ids[c] = fold_ids && check_ids && !ids_matter[c].load();
            if (drop_ids[c]) ctx->ids_dropped.fetch_add(1, std::memory_order_relaxed);
            if (c == 0) tr("first sub-batch coded");
            if ((rc = upload(c, cp))) return rc;
            tr_rec(k, 1, cp);
            RS_HIP(ctx, hipEventRecord(ctx->ev_copy[k], cp));
            // consecutive sub-batches alternate between the context's two workspaces and launch streams: a
            // sub-batch's grid build is then queued beside its predecessor's occlusion kernel and starts in its tail
            // (enqueue_batch chains the occlusion kernels themselves)
            const int w = (int)(c & 1);
            hipStream_t st = w ? ctx->stream2 : ctx->stream;
            RS_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_copy[k], 0));
            if (used[k]) RS_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_d2h[k], 0));  // output slot k has left the device
            const size_t s0 = cut[c], s1 = cut[c + 1], na = structure_offsets[s1] - structure_offsets[s0];
            const size_t nr = want_res ? res_cut[c + 1] - res_cut[c] : 0;
            Pending pd;
            pd.batch.x = dev_x[k];
            pd.batch.y = dev_y[k];
            pd.batch.z = dev_z[k];
            pd.batch.radius = (const float *)br[k]->p;
            pd.batch.id = drop_ids[c] ? nullptr : fold_ids ? id_mapped + structure_offsets[s0] : id ? (const uint64_t *)bi[k]->p : nullptr;
            const char *dblk = (const char *)ctx->in_pack[k].p;
            pd.id32 = fold_ids && !drop_ids[c] ? (const uint32_t *)(dblk + pack[c].o_id) : nullptr;
            pd.ids_needed_known = fold_ids && check_ids && !drop_ids[c];
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
            tr_rec(k, 2, st);
            if ((rc = enqueue_batch(ctx, pd, ctx->slot[k]))) return rc;
            tr_rec(k, 3, st);
            RS_HIP(ctx, hipEventRecord(ctx->ev_done[k], st));
            RS_HIP(ctx, hipStreamWaitEvent(dn, ctx->ev_done[k], 0));
            if ((rc = copy_out(c))) return rc;
            used[k] = true;
            if (h2h_trace) tr(c + 1 == n_sub ? "last sub-batch enqueued" : "sub-batch enqueued");
        }
        turn.pass(cp);  // the next call's uploads follow this one's last
        tr("turn passed on");
        for (int k = 0; k < kSlots; k++) {
            if (!used[k]) continue;
            RS_HIP(ctx, hipEventSynchronize(ctx->ev_done[k]));
            check(k);
            if ((rc = drain(k))) return rc;
        }
        RS_HIP(ctx, hipStreamSynchronize(dn));
        tr("all done");
        if (h2h_trace && tr_ref && n_sub <= (size_t)kSlots) {
            // the device's side of the same call: each sub-batch's uploads and kernels on the host trace's clock
            for (size_t c = 0; c < n_sub; c++) {
                float t[4] = {};
                for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&t[i], tr_ref, ctx->tr_ev[c][i]);
                std::fprintf(stderr, "h2h ctx %p device: sub-batch %zu uploads %9.1f .. %9.1f us, kernels %9.1f .. %9.1f us\n", (void *)ctx, c,
                             tr_ref_host_us + t[0] * 1e3, tr_ref_host_us + t[1] * 1e3, tr_ref_host_us + t[2] * 1e3, tr_ref_host_us + t[3] * 1e3);
            }
        }
        if (err) return err;
        if (!need_cells) return RSASA_OK;
        if (need_cells >= 0xFFFFFFF0ull || attempt >= 3)
            return fail(ctx, RSASA_ERR_GRID_TOO_LARGE, "batch needs more than 2^32 grid cells; split it");
        ctx->cell_capacity = need_cells + need_cells / 8 + 1024;
    }
}

// ---- a stream of host batches (ABI 3) ----

int rsasa_context_clone_settings(rsasa_context_t *dst, rsasa_context_t *src)
{
    int rc = resolve_ctx(src);
    if (rc) return rc;
    if (!dst || dst == src) return dst ? RSASA_OK : RSASA_ERR_INVALID_ARGUMENT;
    int simd = 8;
    bool small = true, overlap = false;
    OcclusionTuning tune;
    {
        std::lock_guard<std::recursive_mutex> lk(src->mu);
        simd = src->simd_width; small = src->small_path; overlap = src->overlap_tail; tune = src->tuning;
    }
    std::lock_guard<std::recursive_mutex> lk(dst->mu);
    tune.deferred_hint = dst->tuning.deferred_hint;  // (a measurement of dst's own batches, not a setting)
    dst->simd_width = simd; dst->small_path = small; dst->overlap_tail = overlap; dst->tuning = tune;
    return RSASA_OK;
}

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
        // (a full queue: wait until its oldest batch has been computed - its status stays queued for rsasa_host_batch_wait)
        hs->cv_done.wait(lk, [&] { return hs->jobs.size() < HostStream::kMaxQueued || hs->jobs.front()->done; });
        if (hs->jobs.size() >= HostStream::kMaxQueued)
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "too many host batches enqueued and not waited for (rsasa_host_batch_wait)");
        job->ticket = hs->next_ticket++;
        hs->jobs.push_back(job);
    }
    hs->cv_work.notify_all();
    return RSASA_OK;
}

int rsasa_host_batch_wait(rsasa_context_t *ctx)
{
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
        if (hs->jobs.empty()) return RSASA_OK;
        job = hs->jobs.front();
        hs->cv_done.wait(lk, [&] { return job->done; });
        hs->jobs.pop_front();
    }
    hs->cv_done.notify_all();  // (an enqueue may be waiting for room)
    if (job->rc) return fail(ctx, job->rc, job->error.c_str());
    return RSASA_OK;
}

int rsasa_host_batch_wait_all(rsasa_context_t *ctx)
{
    int first = RSASA_OK;
    for (;;) {
        {
            int rc = resolve_ctx(ctx);
            if (rc) return rc;
            std::lock_guard<std::recursive_mutex> lk(ctx->mu);
            if (!ctx->host_stream) return first;
            std::lock_guard<std::mutex> lk2(ctx->host_stream->mu);
            if (ctx->host_stream->jobs.empty()) return first;
        }
        const int rc = rsasa_host_batch_wait(ctx);
        if (rc && !first) first = rc;
    }
}

int rsasa_calculate_sasa_soa(rsasa_context_t *ctx, const float *x, const float *y,
                             const float *z, const float *radius, const uint64_t *id,
                             size_t n_atoms, float probe_radius, size_t n_points,
                             float *out_sasa)
{
    if (n_atoms >= 0xFFFFFFF0ull) return RSASA_ERR_INVALID_ARGUMENT;
    if (n_atoms && !out_sasa) return RSASA_ERR_INVALID_ARGUMENT;
    const uint32_t offsets[2] = {0u, (uint32_t)n_atoms};
    return rsasa_calculate_sasa_batch(ctx, x, y, z, radius, id, offsets, 1, probe_radius, n_points,
                                      out_sasa, nullptr, 0, nullptr);
}

int rsasa_calculate_sasa_internal(rsasa_context_t *ctx, const rsasa_atom_t *atoms,
                                  size_t n_atoms, float probe_radius, size_t n_points,
                                  ptrdiff_t threads, float *out_sasa)
{
    (void)threads;  // sequential-vs-rayon switch in the reference (src/lib.rs:278); no meaning here
    if (n_atoms && (!atoms || !out_sasa)) return RSASA_ERR_INVALID_ARGUMENT;
    std::vector<float> soa;
    std::vector<uint64_t> ids;
    try {
        soa.resize(4 * n_atoms);
        ids.resize(n_atoms);
    } catch (const std::bad_alloc &) {
        return RSASA_ERR_OUT_OF_MEMORY;
    }
    float *x = soa.data(), *y = x + n_atoms, *z = y + n_atoms, *r = z + n_atoms;
    for (size_t i = 0; i < n_atoms; i++) {
        x[i] = atoms[i].position[0];
        y[i] = atoms[i].position[1];
        z[i] = atoms[i].position[2];
        r[i] = atoms[i].radius;
        ids[i] = atoms[i].id;
    }
    return rsasa_calculate_sasa_soa(ctx, x, y, z, r, ids.data(), n_atoms, probe_radius, n_points,
                                    out_sasa);
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

int rsasa_segment_sums(rsasa_context_t *ctx, const float *values, size_t n_values,
                       const uint32_t *offsets, size_t n_segments, float *out)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (n_segments == 0) return RSASA_OK;
    if (!offsets || !out || (n_values && !values))
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_values >= 0xFFFFFFF0ull || n_segments >= 0xFFFFFFF0ull || offsets[n_segments] > n_values)
        return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "offsets exceed n_values");
    for (size_t k = 0; k < n_segments; k++)
        if (offsets[k] > offsets[k + 1])
            return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    RS_DEVICE(ctx);
    if (ctx->n_pending && (rc = wait_pending(ctx))) return rc;
    if ((rc = reserve(ctx, ctx->atom_sasa, std::max<size_t>(n_values, 1) * 4))) return rc;
    if ((rc = reserve(ctx, ctx->in_res, (n_segments + 1) * 4))) return rc;
    if ((rc = reserve(ctx, ctx->out_res, n_segments * 4))) return rc;
    if ((rc = reserve(ctx, ctx->status, sizeof(BatchStatus)))) return rc;
    hipStream_t st = ctx->stream;
    RS_HIP(ctx, hipMemsetAsync(ctx->status.p, 0, sizeof(BatchStatus), st));
    if (n_values)
        RS_HIP(ctx, hipMemcpyAsync(ctx->atom_sasa.p, values, n_values * 4, hipMemcpyHostToDevice, st));
    RS_HIP(ctx, hipMemcpyAsync(ctx->in_res.p, offsets, (n_segments + 1) * 4, hipMemcpyHostToDevice, st));
    BatchView v{};
    v.residue_offsets = (const uint32_t *)ctx->in_res.p;
    v.n_residues = (uint32_t)n_segments;
    v.status = (BatchStatus *)ctx->status.p;
    v.atom_sasa = (float *)ctx->atom_sasa.p;
    v.residue_sasa = (float *)ctx->out_res.p;
    launch_residue_sums(v, st);
    RS_HIP(ctx, hipMemcpyAsync(out, ctx->out_res.p, n_segments * 4, hipMemcpyDeviceToHost, st));
    RS_HIP(ctx, hipStreamSynchronize(st));
    return RSASA_OK;
}

int rsasa_sphere_points(size_t n_points, float *out_x, float *out_y, float *out_z)
{
    if (!n_points || !out_x || !out_y || !out_z) return RSASA_ERR_INVALID_ARGUMENT;
    generate_sphere_points(n_points, out_x, out_y, out_z);
    return RSASA_OK;
}

}  // extern "C"
