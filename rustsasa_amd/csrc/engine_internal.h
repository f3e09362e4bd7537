// Internals shared by the engine's host sources (context.cpp, host_batch.cpp, combine.cpp): the context object, its
// helper types and the functions that cross file boundaries.  Nothing here is part of the C ABI (include/rustsasa_amd.h).
#pragma once
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

const char *tuning_env(const char *name);  // an RSASA_* measurement switch, read only under RSASA_TUNING=1 (context.cpp)
void generate_sphere_points(size_t n, float *x, float *y, float *z);

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
NodeCpus device_node_cpus(int device);
bool bind_thread_to(pthread_t th, const NodeCpus &nc);  // binds a thread to `nc` (intersected with what it may run on)

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
    bool solo_ok = false;            // the caller runs the batch again if asked to (rsasa_batch_wait: OcclusionChain::solo)
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
        bool waited_for = false;  // a caller's thread is in rsasa_host_batch_wait for this batch
    };
    static constexpr int kMaxWorkers = 4;
    int n_workers = 2;
    // The workers' contexts create their streams on hardware queues of their own (new_stream).  The next call's uploads
    // hide a call's fill and drain, so each call is cut into two sub-batches only (measured, ms per proteome batch:
    // 2 sub-batches 4.30, 3 5.15, 8 5.29; on the pooled queues 6.40 / 6.53 / 5.34).
    size_t sub_batches = 0;
    static constexpr size_t kMaxQueued = 8;  // enqueued and not yet waited for (a further enqueue fails at once: RSASA_ERR_QUEUE_FULL)
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
            sorted_xyzr, sorted_orig, sorted_id, sorted_id32, status, atom_sasa, claim, ids_seg;
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
    std::atomic<uint64_t> ids_kept_structures{0}; // structures of the last checked (sub-)batch that kept their ids (rsasa_context_ids_kept)
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
    std::atomic<int> combine_wait_us{-1};      // rsasa_context_set_call_combining: -1 off, else how long a leader may hold a batch back for company
};

namespace rsasa {

int fail(rsasa_context *ctx, int code, const char *what, hipError_t e = hipSuccess);

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
extern LinkTurn g_link[64];

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
        std::unique_lock<std::mutex> lk(lt->mu);
        lt->cv.wait(lk, [&] { return !lt->busy; });
        lt->busy = true;
        held = true;
        // (the event belongs to the previous holder's context: rsasa_context_destroy clears lt->last under this lock before it
        // destroys the event, so the wait is queued while the lock is held)
        return lt->owner != c && lt->last ? hipStreamWaitEvent(cp, lt->last, 0) : hipSuccess;
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

int reserve(rsasa_context *ctx, DeviceBuffer &b, size_t bytes);  // grows `b` (contents are NOT preserved)
void release(DeviceBuffer &b);
int get_lattice(rsasa_context *ctx, size_t n_points, Lattice *out);
hipError_t new_stream(rsasa_context *ctx, hipStream_t *out, int level);
int ensure_side_stream(rsasa_context *ctx);
int ensure_copy_streams(rsasa_context *ctx);
int enqueue_batch(rsasa_context *ctx, const Pending &pd, rsasa_context::HostSlot &hs);
int wait_one(rsasa_context *ctx, Pending &pd);
int wait_oldest(rsasa_context *ctx);
int wait_pending(rsasa_context *ctx);
int resolve_ctx(rsasa_context *&ctx);
int context_create(int device, int own_queues, rsasa_context_t **out_ctx);  // own_queues: rsasa_context::own_queues
int batch_enqueue(rsasa_context *ctx, const rsasa_device_batch_t *batch, float probe_radius, size_t n_points, void *hip_stream,
                  bool ids_needed_known);  // rsasa_batch_enqueue with the host's verdict on the ids

// ---- the small-batch path (host_batch.cpp), shared with the call combiner (combine.cpp) ----
// Where a small batch's atoms come from: columns (the SoA entry points) or rsasa_atom_t records (rsasa_calculate_sasa_internal:
// they are de-interleaved straight into the pinned staging block, no temporary columns).
struct SmallSource {
    const float *x = nullptr, *y = nullptr, *z = nullptr, *radius = nullptr;
    const uint64_t *id = nullptr;
    const rsasa_atom_t *aos = nullptr;
    bool has_id() const { return aos != nullptr || id != nullptr; }
};
struct SmallLayout {  // the pinned staging block of a small batch (see small_layout)
    size_t S = 0, N = 0, W = 0, R = 0;
    bool has_id = false;
    float probe = 0.f;
    unsigned long long tail_begin = 0;
    size_t o_grid = 0, o_win = 0, o_x = 0, o_y = 0, o_z = 0, o_r = 0, o_id = 0, o_res = 0, in_bytes = 0;
    size_t o_oa = 0, o_or = 0, out_bytes = 0;
};
constexpr int kNotSmall = 1;  // (positive: not an error) the batch goes through the general path
bool small_structure_grid(const SmallSource &src, uint32_t begin, uint32_t end, float probe, StructGrid *out);
void small_fill(const SmallSource &src, uint32_t begin, uint32_t end, const SmallLayout &lay, char *h, size_t at);
SmallLayout small_layout(size_t S, size_t N, size_t W, size_t R, bool has_id, unsigned long long total_cells16);
int small_reserve(rsasa_context *ctx, const SmallLayout &l, const Lattice &lat, bool own_staging);
int small_run(rsasa_context *ctx, const SmallLayout &l, const Lattice &lat, const StructGrid *grids, const uint4 *windows, char *h, char *hout,
              const void *records);
extern std::atomic<uint64_t> g_small_trace_ns[2];
void small_fill_records(const SmallSource &src, uint32_t begin, uint32_t end, rsasa_atom_t *recs, size_t at);
int run_small_host_batch(rsasa_context *ctx, const SmallSource &in, const uint32_t *so, size_t S, float probe, size_t n_points,
                         float *out_atom, const uint32_t *ro, size_t R, float *out_res);

// ---- call combining (combine.cpp) ----
// rsasa_context_set_call_combining: per-structure calls of several host threads are merged into one batch launch.
// kNotCombined (positive: not an error): the call is not one the combiner takes - it runs by itself.
constexpr int kNotCombined = 2;
int combine_call(rsasa_context *ctx, const SmallSource &in, size_t n_atoms, float probe, size_t n_points, float *out);

}  // namespace rsasa
