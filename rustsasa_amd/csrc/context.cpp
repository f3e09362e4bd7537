// C ABI of the engine (include/rustsasa_amd.h): contexts, HBM workspace,
// sphere lattice cache, batch enqueue / wait.  Host code only; the kernels
// live in kernels.hip.  There is no CPU compute path in this library.
#include "engine_internal.h"

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

NodeCpus device_node_cpus(int device)
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
bool bind_thread_to(pthread_t th, const NodeCpus &nc)
{
    if (!nc.valid) return false;
    cpu_set_t cur, want;
    if (pthread_getaffinity_np(th, sizeof(cur), &cur) != 0) return false;
    CPU_AND(&want, &cur, &nc.set);
    if (CPU_COUNT(&want) == 0) return false;
    return pthread_setaffinity_np(th, sizeof(want), &want) == 0;
}

}  // namespace rsasa

namespace rsasa {

int fail(rsasa_context *ctx, int code, const char *what, hipError_t e)
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

LinkTurn g_link[64];

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
    for (size_t s = 0; s < S; s++) {
        const uint32_t b = bt.structure_offsets_host[s], e = bt.structure_offsets_host[s + 1];
        n_seg += (e - b + kSegmentAtoms - 1) / kSegmentAtoms;
        n_large += e - b > kIdAtomsSmall && e - b <= kIdAtomsLarge;
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
    const size_t ids_seg_words = (N + 2047) / 2048;  // (BatchView::ids_seg: a bit per 64 atoms, two bitmaps)
    if (has_id && (rc = reserve(ctx, W.ids_seg, 2 * ids_seg_words * 4 + 16))) return rc;
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
    // (host-folded ids are checked as well: folds that differ are ids that differ, and a structure with two equal folds
    // simply keeps its ids)
    v.ids_check = (has_id && !pd.ids_needed_known && !keep_ids && !tuning_env("RSASA_NO_ID_CHECK")) ? 1u : 0u;
    hs.ids_check = v.ids_check != 0u;
    v.large_sids = reinterpret_cast<const uint32_t *>((const Segment *)W.segments.p + n_seg);
    v.n_large = (uint32_t)n_large;
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
    v.ids_seg = has_id ? (uint32_t *)W.ids_seg.p : nullptr;
    v.ids_seg_words = (uint32_t)ids_seg_words;
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
        oc.solo = pd.solo_ok && v.ids_check != 0u;
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
        if (!stt.overflow && ctx->slot[pd.ws].ids_check && (stt.ids_unordered & 2u) != 0u && pd.attempts < 3) {
            // the batch's one occlusion launch was the id-less instantiation and the ids do matter (OcclusionChain::solo):
            // nothing was computed - the batch runs again, with its ids and without the check
            ctx->ids_drop_hint = false;
            ctx->ids_unordered_hint = (stt.ids_unordered & 1u) != 0u;
            pd.solo_ok = false;  // (as a pair this time: every structure in the instantiation that is its own)
            pd.attempts++;
            int rc = enqueue_batch(ctx, pd, ctx->slot[pd.ws]);
            if (rc) {
                pd.active = false;
                return rc;
            }
            continue;
        }
        if (!stt.overflow) {
            ctx->tuning.deferred_hint = stt.deferred;  // (sizes the next batch's launch over its deferred list)
            if (ctx->slot[pd.ws].ids_check) {
                ctx->ids_drop_hint = !stt.ids_needed;
                ctx->ids_unordered_hint = (stt.ids_unordered & 1u) != 0u;
                ctx->ids_kept_structures.store(stt.ids_needed, std::memory_order_relaxed);
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

}  // namespace rsasa

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
    case RSASA_ERR_QUEUE_FULL: return "host batch queue full";
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


int rsasa_context_create(int device, rsasa_context_t **out_ctx)
{
    int own = 0;
    if (const char *v = tuning_env("RSASA_CTX_OWN_QUEUES")) own = std::atoi(v);  // (experiment: DESIGN.md 6)
    return context_create(device, own, out_ctx);
}

}  // extern "C"

int rsasa::context_create(int device, int own_queues, rsasa_context_t **out_ctx)
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

extern "C" {

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
                            &ctx->sorted_orig, &ctx->sorted_id, &ctx->sorted_id32, &ctx->status, &ctx->atom_sasa, &ctx->claim, &ctx->ws[0].ids_seg,
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
                                &w1.status, &w1.atom_sasa, &w1.claim, &w1.ids_seg})
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

int rsasa_context_ids_kept(rsasa_context_t *ctx, uint64_t *out_structures)
{
    int rc = resolve_ctx(ctx);
    if (rc) return rc;
    if (!out_structures) return fail(ctx, RSASA_ERR_INVALID_ARGUMENT, "out_structures is NULL");
    *out_structures = ctx->ids_kept_structures.load(std::memory_order_relaxed);
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
    return rsasa::batch_enqueue(ctx, batch, probe_radius, n_points, hip_stream, false);
}

}  // extern "C"

// rsasa_batch_enqueue; ids_needed_known: the caller has looked at the ids itself and found that they matter (the host
// paths check while the coordinates cross the link): the device does not check again
int rsasa::batch_enqueue(rsasa_context *ctx, const rsasa_device_batch_t *batch, float probe_radius, size_t n_points, void *hip_stream,
                         bool ids_needed_known)
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
    pd.solo_ok = true;  // (rsasa_batch_wait runs a batch again when the id-less launch was the wrong guess: wait_one)
    pd.ids_needed_known = ids_needed_known;
    rc = enqueue_batch(ctx, pd, ctx->slot[w]);
    pd.active = (rc == RSASA_OK);
    if (pd.active) ctx->n_pending++;
    return rc;
}

extern "C" {

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
