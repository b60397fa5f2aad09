// chip_api.hip -- C-ABI entry points of libcerebro_hip.so (see include/cerebro_hip.h for the contract and
// the reference lines each entry point replaces).  Host-side C++ over the HIP runtime; no torch, no CPU
// fallback: every compute entry point runs HIP kernels or fails with a CHIP_ERR_* status.
#include "chip_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace chip;

extern "C" {

const char *chip_strerror(int status)
{
    switch (status) {
        case CHIP_OK: return "ok";
        case CHIP_ERR_INVALID_ARG: return "invalid argument";
        case CHIP_ERR_NO_DEVICE: return "no usable HIP device";
        case CHIP_ERR_HIP: return "HIP runtime error (see chip_last_hip_error)";
        case CHIP_ERR_OOM: return "out of device memory";
        case CHIP_ERR_NOT_F32: return "descriptor value not exactly representable as float32";
        case CHIP_ERR_NONFINITE: return "NaN/Inf in descriptor";
        case CHIP_ERR_RANGE: return "index / length out of range";
        case CHIP_ERR_UNSUPPORTED: return "unsupported configuration";
        case CHIP_ERR_TOO_FEW_POINTS: return "fewer than 20 correspondences";
        case CHIP_ERR_BUSY: return "async slot busy or empty";
    }
    return "unknown status";
}

int chip_abi_version(void) { return CHIP_ABI_VERSION; }

int chip_last_hip_error(const chip_ctx *ctx, const char **text)
{
    if (!ctx) return (int)hipErrorInvalidValue;
    if (text) *text = hipGetErrorString(ctx->last_hip);
    return (int)ctx->last_hip;
}

void chip_dot_params_default(chip_dot_params *p)
{
    if (!p) return;
    p->locality = 12;                 // Cerebro.cpp:912
    p->thresh = (double)(float)0.85;  // Cerebro.cpp:913 (float) compared with double at :1056
    p->lag = 50;                      // Cerebro.cpp:914
    p->min_new = 3;                   // Cerebro.cpp:962
    p->min_k = 5;                     // Cerebro.cpp:1022
}

void chip_ransac_params_default(chip_ransac_params *p)
{
    if (!p) return;
    p->error_thresh = 0.03;        // DlsPnpWithRansac.cpp:208
    p->min_inlier_ratio = 0.7;     // :209
    p->max_iterations = 50;        // :210
    p->min_iterations = 5;         // :211
    p->use_mle = 1;                // :212
    p->sample_size = 15;           // DlsPnpWithRansac.h:45
    p->failure_probability = 0.01; // theia::RansacParameters default
    p->seed = 0x5EEDCE7EB80ULL;
    p->n_hypotheses = 0;
    p->reserved = 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ internals
static int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

static int ensure_capacity(Ctx *c, int64_t local_rows)
{
    // caller holds append_mu
    const int64_t need = (local_rows + c->seg_rows - 1) >> c->seg_shift;
    if (need > kMaxSegs) return CHIP_ERR_OOM;
    bool grew = false;
    while ((int64_t)c->segs.size() < need) {
        float *p = nullptr;
        CHIP_HIP(c, hipMalloc(&p, (size_t)c->seg_rows * c->D * sizeof(float)));
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->segs.push_back(p);
        }
        grew = true;
    }
    if (grew) {
        CHIP_HIP(c, hipMemcpyAsync(c->seg_table_dev, c->segs.data(), c->segs.size() * sizeof(float *), hipMemcpyHostToDevice, c->s_append));
        CHIP_HIP(c, hipStreamSynchronize(c->s_append));
    }
    return CHIP_OK;
}

static void destroy_ctx(chip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    pnp_destroy(c);
    icp_destroy(c);
    batch_destroy(c);
    for (float *p : c->segs) (void)hipFree(p);
    if (c->seg_table_dev) (void)hipFree(c->seg_table_dev);
    if (c->ring_dev) (void)hipFree(c->ring_dev);
    if (c->stage_dev) (void)hipFree(c->stage_dev);
    if (c->flags_dev) (void)hipFree(c->flags_dev);
    if (c->flags_host) (void)hipHostFree(c->flags_host);
    for (int i = 0; i < Ctx::kRing; i++) {
        if (c->partial_dev[i]) (void)hipFree(c->partial_dev[i]);
        if (c->ev_scan[i]) (void)hipEventDestroy(c->ev_scan[i]);
        if (c->ev_merged[i]) (void)hipEventDestroy(c->ev_merged[i]);
    }
    if (c->s_scan) (void)hipStreamDestroy(c->s_scan);
    if (c->s_scan2) (void)hipStreamDestroy(c->s_scan2);
    if (c->topk_host) (void)hipHostFree(c->topk_host);
    if (c->qvec_dev) (void)hipFree(c->qvec_dev);
    for (Slot &s : c->slots) {
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.host) (void)hipHostFree(s.host);
    }
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    if (c->own_query_stream && c->s_query) (void)hipStreamDestroy(c->s_query);
    if (c->s_append) (void)hipStreamDestroy(c->s_append);
    if (c->s_pnp) (void)hipStreamDestroy(c->s_pnp);
    delete c;
}

static int create_impl(chip_ctx *c, int64_t capacity_hint)
{
    hipDeviceProp_t prop;
    CHIP_HIP(c, hipGetDeviceProperties(&prop, c->device));
    c->n_cus = prop.multiProcessorCount;
    std::strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
    if (std::strncmp(c->arch, "gfx950", 6) != 0 && !std::getenv("CHIP_ALLOW_ANY_ARCH")) return CHIP_ERR_NO_DEVICE;

    // segment geometry: power-of-two rows, ~512 MiB each
    int64_t rows = kSegBytesTarget / ((int64_t)c->D * 4);
    int shift = 0;
    while ((2ll << shift) <= rows) shift++;
    if (shift < 6) shift = 6;
    c->seg_shift = shift;
    c->seg_rows = 1ll << shift;
    // The querier reads segs[] without c->mu (row_ptr_host) while the appender may open a new segment: reserve the whole
    // table (32 KiB) so that push_back never reallocates -- entries below the published length are immutable.
    c->segs.reserve(kMaxSegs);

    CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_query, hipStreamNonBlocking));
    CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_append, hipStreamNonBlocking));
    CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_pnp, hipStreamNonBlocking));
    CHIP_HIP(c, hipMalloc(&c->seg_table_dev, kMaxSegs * sizeof(float *)));
    CHIP_HIP(c, hipMemset(c->seg_table_dev, 0, kMaxSegs * sizeof(float *)));
    if (c->nranks > 1) {
        CHIP_HIP(c, hipMalloc(&c->ring_dev, (size_t)CHIP_RING_ROWS * c->D * sizeof(float)));
        CHIP_HIP(c, hipMemset(c->ring_dev, 0, (size_t)CHIP_RING_ROWS * c->D * sizeof(float)));
    }
    c->stage_bytes = 64ull << 20;
    if (c->stage_bytes < (size_t)c->D * 8 * 64) c->stage_bytes = (size_t)c->D * 8 * 64;
    CHIP_HIP(c, hipMalloc(&c->stage_dev, c->stage_bytes));
    CHIP_HIP(c, hipMalloc(&c->flags_dev, sizeof(uint32_t)));
    CHIP_HIP(c, hipHostMalloc(&c->flags_host, sizeof(uint32_t), hipHostMallocDefault));

    c->scan_block = env_int("CHIP_SCAN_BLOCK", 0);   // 0 = chosen from D per launch (kernels.hip scan_shape)
    if (c->scan_block != 256 && c->scan_block != 512 && c->scan_block != 768 && c->scan_block != 1024) c->scan_block = 0;
    c->scan_blocks_per_cu = env_int("CHIP_SCAN_BPC", 2);
    if (c->scan_blocks_per_cu < 1) c->scan_blocks_per_cu = 1;
    c->scan_variant = env_int("CHIP_SCAN_VARIANT", 0);
    c->scan_reserve = env_int("CHIP_SCAN_RESERVE", 0);
    c->max_grid = 512;  // K2 (one 512-thread workgroup) keeps one partial list per thread
    {
        const int pr = env_int("CHIP_SCAN_STREAM_PRIORITY", 0);   // tuning only: 0 = default class, 1 = highest, -1 = lowest
        if (pr == 0) CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_scan, hipStreamNonBlocking));
        else {
            int lo = 0, hi = 0;
            CHIP_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
            CHIP_HIP(c, hipStreamCreateWithPriority(&c->s_scan, hipStreamNonBlocking, pr > 0 ? hi : lo));
        }
    }
    if (env_int("CHIP_SCAN_STREAMS", 2) >= 2) CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_scan2, hipStreamNonBlocking));
    for (int i = 0; i < Ctx::kRing; i++) {
        CHIP_HIP(c, hipMalloc(&c->partial_dev[i], (size_t)c->max_grid * CHIP_MAX_NQ * CHIP_MAX_TOPK * sizeof(chip_topk_entry)));
        CHIP_HIP(c, hipEventCreateWithFlags(&c->ev_scan[i], hipEventDisableTiming));
        CHIP_HIP(c, hipEventCreateWithFlags(&c->ev_merged[i], hipEventDisableTiming));
    }
    CHIP_HIP(c, hipHostMalloc(&c->topk_host, (size_t)CHIP_MAX_NQ * CHIP_MAX_TOPK * sizeof(chip_topk_entry), hipHostMallocDefault));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&c->topk_dev, c->topk_host, 0));
    CHIP_HIP(c, hipMalloc(&c->qvec_dev, (size_t)CHIP_MAX_NQ * c->D * sizeof(float)));
    for (Slot &s : c->slots) {
        CHIP_HIP(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        // pinned + mapped: the deciding workgroup stores the record here directly, no D2H copy kernel per tick
        CHIP_HIP(c, hipHostMalloc(&s.host, sizeof(chip_tick_result), hipHostMallocDefault));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&s.dev, s.host, 0));
    }
    int rc = pnp_create(c);
    if (rc != CHIP_OK) return rc;
    if (capacity_hint > 0) {
        std::lock_guard<std::mutex> lk(c->append_mu);
        rc = ensure_capacity(c, local_count(c, capacity_hint));
        if (rc != CHIP_OK) return rc;
    }
    return CHIP_OK;
}

// Enqueue K1 (scan + per-workgroup top-k, on s_scan) and K2 (cross-workgroup merge [+ accept decision], on the ctx
// stream behind an event) for nq queries over the global prefix [0,k).  out (device or pinned host, optional) gets
// [nq][K]; res (optional) the decision record of Cerebro.cpp:1056.  Consecutive calls pipeline: scans run back to
// back on s_scan while the previous merge (and whatever the caller enqueues after it on the ctx stream) proceeds.
static int enqueue_scan_merge(Ctx *c, int64_t k, const float *const *q, int nq, int K, int64_t l,
                              const chip_dot_params *p, chip_topk_entry *out, chip_tick_result *res, bool tick = false)
{
    const int b = (int)(c->n_enqueued++ % Ctx::kRing);
    // Ticks (queries already resident) over a short prefix alternate between two scan streams so that the ramp-down of
    // one launch overlaps the ramp-up of the next (measured, 1 -> 2 streams: 10k rows 45 -> 38 us/tick, 60k 172 -> 153,
    // 125k 317 -> 308, 500k 1191 -> 1157; at 1M the gain is < 1 %, and launches that overlap would no longer have a
    // meaningful per-launch duration for the roofline, so long scans and profiled runs stay on one stream).  Anything that
    // uploads its queries on s_scan first stays on s_scan.
    const bool short_scan = (double)local_count(c, k) * c->D * sizeof(float) <= 8.0 * 1024 * 1024 * 1024;
    hipStream_t s_scan = (tick && short_scan && !c->prof_on && c->s_scan2 && (c->n_enqueued & 1)) ? c->s_scan2 : c->s_scan;
    ScanArgs a;
    a.seg_table = c->seg_table_dev;
    a.seg_shift = c->seg_shift;
    a.seg_mask = c->seg_rows - 1;
    a.n_rows = local_count(c, k);
    a.D = c->D;
    a.K = K;
    for (int i = 0; i < CHIP_MAX_NQ; i++) a.q[i] = i < nq ? q[i] : nullptr;
    a.idx_mul = c->nranks;
    a.idx_add = c->nranks == 1 ? 0 : c->rank;
    a.partial = c->partial_dev[b];
    const int grid = scan_grid_for(c, a.n_rows, nq);

    // The merge that last read this buffer ran kRing ticks ago; only when it is not already complete (a stalled ctx
    // stream) does the scan stream need a barrier packet -- in steady state this costs nothing.
    if (hipEventQuery(c->ev_merged[b]) != hipSuccess) CHIP_HIP(c, hipStreamWaitEvent(s_scan, c->ev_merged[b], 0));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->prof_on) {
        if (c->prof_used + 2 > c->prof_ev.size()) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t e;
                CHIP_HIP(c, hipEventCreate(&e));
                c->prof_ev.push_back(e);
            }
        }
        e0 = c->prof_ev[c->prof_used];
        e1 = c->prof_ev[c->prof_used + 1];
        c->prof_used += 2;
        c->prof_bytes_last = (double)a.n_rows * c->D * 4.0;
        CHIP_HIP(c, hipEventRecord(e0, s_scan));
    }
    int rc = launch_scan(c, s_scan, a, nq, grid);
    if (rc != CHIP_OK) return rc;
    if (e1) CHIP_HIP(c, hipEventRecord(e1, s_scan));
    CHIP_HIP(c, hipEventRecord(c->ev_scan[b], s_scan));
    if (c->ring_dev) c->last_scan_ev[s_scan == c->s_scan2 ? 1 : 0] = c->ev_scan[b];   // caller holds ring_mu (RingGuard)
    CHIP_HIP(c, hipStreamWaitEvent(c->s_query, c->ev_scan[b], 0));

    MergeArgs m;
    m.in = c->partial_dev[b];
    m.n_lists = grid;
    m.K = K;
    m.out = out;
    m.result = res;
    m.l = l;
    m.locality = p ? p->locality : 0;
    m.thresh = p ? p->thresh : 0.0;
    rc = launch_merge(c, c->s_query, m, nq);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, hipEventRecord(c->ev_merged[b], c->s_query));
    return CHIP_OK;
}

// Sharded ctx: the query rows of a scan are read from the replicated ring, which the appender overwrites in place.  A
// querier holds ring_mu from the residency check until the scan's completion event is recorded (RingGuard); the appender
// announces the length its call will reach (rows_pending) and picks up the newest scan events under the same lock before it
// touches the ring, and its ring writes wait for those scans.  So a scan either was validated against the post-append
// length, or is ordered before the ring writes.
struct RingGuard {
    std::unique_lock<std::mutex> lk;
    explicit RingGuard(Ctx *c) : lk(c->ring_mu, std::defer_lock) { if (c->ring_dev) lk.lock(); }
};

// Pointers of the query rows (device).  Single GPU: straight into the DB; sharded: the replicated ring.
static int query_row_ptrs(Ctx *c, const int64_t *rows, int nq, int64_t n_global, const float **q)
{
    int64_t total;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        total = c->rows_global;
    }
    for (int i = 0; i < nq; i++) {
        const int64_t g = rows[i];
        if (g < 0 || g >= n_global || g >= total) return CHIP_ERR_RANGE;
        if (c->nranks == 1) {
            q[i] = row_ptr_host(c, g);
        } else {
            // no longer in the replicated ring -- counting the rows of an append that is in flight right now
            const int64_t horizon = c->rows_pending > total ? c->rows_pending : total;
            if (g < horizon - CHIP_RING_ROWS) return CHIP_ERR_RANGE;
            q[i] = c->ring_dev + (g % CHIP_RING_ROWS) * (int64_t)c->D;
        }
    }
    return CHIP_OK;
}

// Host part of the tick (Cerebro.cpp:960-966, :1019-1022, :1098).  Returns CHIP_TICK_* in *status.
// last_l is NOT written here: the caller commits it (c->last_l = l) once the tick has actually been enqueued (or at once
// for TOO_SHORT), so a failed enqueue leaves the state as the reference's loop would (it only reaches :1098 at the end of an
// executed pass).
static int tick_prepare(Ctx *c, int64_t l, const chip_dot_params *p, int32_t *status, int64_t *k_out)
{
    int64_t n;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        n = c->rows_global;
    }
    if (l < 0 || l > n) return CHIP_ERR_RANGE;
    if (l - c->last_l < p->min_new) { *status = CHIP_TICK_SKIPPED; return CHIP_OK; }  // :962-966, last_l untouched
    if (l < 3) return CHIP_ERR_RANGE;  // needs descriptors l-1, l-2, l-3 (:987-989)
    const int64_t k = l - p->lag;      // :1019
    *k_out = k;
    *status = (k > p->min_k) ? CHIP_TICK_SCANNED : CHIP_TICK_TOO_SHORT;  // :1022
    return CHIP_OK;
}

static void fill_immediate(chip_tick_result *r, int32_t status)
{
    std::memset(r, 0, sizeof *r);
    r->status = status;
    r->idx_curr = r->idx_prev = -1;
    for (int q = 0; q < 3; q++) { r->argmax[q] = -1; r->maxv[q] = -INFINITY; }
}

static int tick_enqueue_slot(Ctx *c, int64_t l, const chip_dot_params *p, Slot &s)
{
    if (s.in_flight) return CHIP_ERR_BUSY;
    int32_t status = 0;
    int64_t k = 0;
    int rc = tick_prepare(c, l, p, &status, &k);
    if (rc != CHIP_OK) return rc;
    if (status != CHIP_TICK_SCANNED) {
        if (status == CHIP_TICK_TOO_SHORT) c->last_l = l;   // :1098 (the else-branch of :1022 still ends the pass)
        fill_immediate(s.host, status);
        s.immediate = true;
        s.in_flight = true;
        return CHIP_OK;
    }
    const int64_t rows[3] = {l - 1, l - 2, l - 3};  // v, vm, vmm (:987-989)
    const float *q[3];
    RingGuard rg(c);
    rc = query_row_ptrs(c, rows, 3, l, q);
    if (rc != CHIP_OK) return rc;
    rc = enqueue_scan_merge(c, k, q, 3, CHIP_DEFAULT_TOPK, l, p, nullptr, s.dev, true);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, hipEventRecord(s.done, c->s_query));
    c->last_l = l;                     // :1098
    s.immediate = false;
    s.in_flight = true;
    return CHIP_OK;
}

static int tick_collect_slot(Ctx *c, Slot &s, chip_tick_result *out)
{
    if (!s.in_flight) return CHIP_ERR_BUSY;
    if (!s.immediate) CHIP_HIP(c, hipEventSynchronize(s.done));
    *out = *s.host;
    s.in_flight = false;
    return CHIP_OK;
}

static int sync_topk_out(Ctx *c, int nq, int K, double *scores, int64_t *idx)
{
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));  // the last block stored the list into pinned host memory
    for (int i = 0; i < nq * K; i++) {
        if (scores) scores[i] = c->topk_host[i].score;
        if (idx) idx[i] = c->topk_host[i].idx;
    }
    return CHIP_OK;
}

template <typename T>
static int append_impl(Ctx *c, const T *desc, int64_t n, uint32_t flags, int64_t *first_index, bool is_f64)
{
    if (!c || !desc || n < 0) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> alk(c->append_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    int64_t first;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        first = c->rows_global;
    }
    if (first_index) *first_index = first;
    if (n == 0) return CHIP_OK;
    int rc = ensure_capacity(c, local_count(c, first + n));
    if (rc != CHIP_OK) return rc;

    // Upload rows [from, from+count) of this call in staging-sized chunks and run K3 on them.
    auto pass = [&](int64_t from, int64_t count, bool ring) -> int {
        const int64_t chunk_rows = (int64_t)(c->stage_bytes / ((size_t)c->D * sizeof(T)));
        for (int64_t off = from; off < from + count; off += chunk_rows) {
            const int64_t m = (from + count - off) < chunk_rows ? (from + count - off) : chunk_rows;
            CHIP_HIP(c, hipMemcpyAsync(c->stage_dev, desc + off * c->D, (size_t)m * c->D * sizeof(T), hipMemcpyHostToDevice, c->s_append));
            const int r = is_f64 ? launch_narrow_f64(c, c->s_append, (const double *)c->stage_dev, m, first + off, c->flags_dev, ring)
                                 : launch_copy_f32(c, c->s_append, (const float *)c->stage_dev, m, first + off, c->flags_dev, ring);
            if (r != CHIP_OK) return r;
            // the staging buffer is reused by the next chunk: stream order serialises copy -> kernel -> copy
        }
        return CHIP_OK;
    };
    // pass 1 writes the DB only (rows past the published length are invisible); the ring is updated after validation
    *c->flags_host = 0;
    CHIP_HIP(c, hipMemsetAsync(c->flags_dev, 0, sizeof(uint32_t), c->s_append));
    rc = pass(0, n, false);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, hipMemcpyAsync(c->flags_host, c->flags_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, c->s_append));
    CHIP_HIP(c, hipStreamSynchronize(c->s_append));

    const uint32_t bad = *c->flags_host;
    if (bad & 2u) return CHIP_ERR_NONFINITE;
    if ((bad & 1u) && !(flags & CHIP_APPEND_ALLOW_ROUNDING)) return CHIP_ERR_NOT_F32;

    if (c->ring_dev) {  // sharded: mirror the newest rows into the replicated ring (DB store is idempotent)
        hipEvent_t ev[2];
        {
            std::lock_guard<std::mutex> rl(c->ring_mu);
            c->rows_pending = first + n;        // queriers now validate ring residency against the length after this call
            ev[0] = c->last_scan_ev[0];
            ev[1] = c->last_scan_ev[1];
        }
        for (hipEvent_t e : ev)                 // scans already enqueued read their queries before the ring is overwritten
            if (e) CHIP_HIP(c, hipStreamWaitEvent(c->s_append, e, 0));
        const int64_t m = n < CHIP_RING_ROWS ? n : CHIP_RING_ROWS;
        rc = pass(n - m, m, true);
        if (rc != CHIP_OK) return rc;
        CHIP_HIP(c, hipStreamSynchronize(c->s_append));
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);  // publish the new length only now (rows fully resident)
        c->rows_global = first + n;
        c->rows_local = local_count(c, c->rows_global);
        if (bad & 1u) c->lossy_rows += n;  // upper bound: rows of this call
    }
    return CHIP_OK;
}

extern "C" {

// ------------------------------------------------------------------------------------------------ lifecycle
int chip_create(chip_ctx **out, int32_t D, int64_t capacity_hint, int32_t device, int32_t shard_rank, int32_t shard_count)
{
    if (!out) return CHIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (D <= 0 || shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count || capacity_hint < 0) return CHIP_ERR_INVALID_ARG;
    if (D % 4 != 0 || (size_t)D * 4 * CHIP_MAX_NQ > 160 * 1024) return CHIP_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CHIP_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return CHIP_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CHIP_ERR_NO_DEVICE;
    chip_ctx *c = new (std::nothrow) chip_ctx();
    if (!c) return CHIP_ERR_OOM;
    c->D = D;
    c->device = device;
    c->rank = shard_rank;
    c->nranks = shard_count;
    int rc = create_impl(c, capacity_hint);
    if (rc != CHIP_OK) { destroy_ctx(c); return rc; }
    *out = c;
    return CHIP_OK;
}

void chip_destroy(chip_ctx *ctx) { destroy_ctx(ctx); }

int chip_set_stream(chip_ctx *c, void *hip_stream)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    if (c->own_query_stream) CHIP_HIP(c, hipStreamDestroy(c->s_query));
    c->s_query = (hipStream_t)hip_stream;   // may be 0: HIP's null stream is a valid external stream
    c->own_query_stream = false;
    return CHIP_OK;
}

int chip_reset_stream(chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    if (c->own_query_stream) return CHIP_OK;
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_query, hipStreamNonBlocking));
    c->own_query_stream = true;
    return CHIP_OK;
}

int chip_synchronize(chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(c->s_append));
    CHIP_HIP(c, hipStreamSynchronize(c->s_scan));
    if (c->s_scan2) CHIP_HIP(c, hipStreamSynchronize(c->s_scan2));
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    CHIP_HIP(c, hipStreamSynchronize(c->s_pnp));
    return CHIP_OK;
}

// ------------------------------------------------------------------------------------------------ append
int chip_db_append_f64(chip_ctx *c, const double *desc, int64_t n, uint32_t flags, int64_t *first_index)
{
    return append_impl<double>(c, desc, n, flags, first_index, true);
}

int chip_db_append_f32(chip_ctx *c, const float *desc, int64_t n, int64_t *first_index)
{
    return append_impl<float>(c, desc, n, 0, first_index, false);
}

int64_t chip_db_size(const chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->rows_global;
}

int chip_db_read_rows_f32(chip_ctx *c, const int64_t *rows, int64_t n, float *out)
{
    if (!c || !rows || !out || n < 0) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    int64_t total;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        total = c->rows_global;
    }
    for (int64_t i = 0; i < n; i++) {
        const int64_t g = rows[i];
        if (g < 0 || g >= total) return CHIP_ERR_RANGE;
        const float *src;
        if (owns_row(c, g)) src = row_ptr_host(c, local_of(c, g));
        else if (g >= total - CHIP_RING_ROWS) src = c->ring_dev + (g % CHIP_RING_ROWS) * (int64_t)c->D;
        else return CHIP_ERR_RANGE;
        CHIP_HIP(c, hipMemcpyAsync(out + i * c->D, src, (size_t)c->D * sizeof(float), hipMemcpyDeviceToHost, c->s_query));
    }
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    return CHIP_OK;
}

int chip_db_append_synthetic(chip_ctx *c, int64_t n, uint64_t seed,
                             const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant)
{
    if (!c || n < 0 || n_plant < 0 || (n_plant > 0 && (!plant_dst || !plant_src || !plant_kind))) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> alk(c->append_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    int64_t first;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        first = c->rows_global;
    }
    for (int64_t i = 0; i < n_plant; i++) {
        if (plant_dst[i] < first || plant_dst[i] >= first + n || plant_src[i] < 0) return CHIP_ERR_RANGE;
        if (i > 0 && plant_dst[i] <= plant_dst[i - 1]) return CHIP_ERR_INVALID_ARG;
        if (plant_kind[i] != 1 && plant_kind[i] != 2) return CHIP_ERR_INVALID_ARG;
    }
    if (n == 0) return CHIP_OK;
    int rc = ensure_capacity(c, local_count(c, first + n));
    if (rc != CHIP_OK) return rc;
    int64_t *pd = nullptr, *ps = nullptr;
    int32_t *pk = nullptr;
    if (n_plant > 0) {
        CHIP_HIP(c, hipMalloc(&pd, n_plant * sizeof(int64_t)));
        CHIP_HIP(c, hipMalloc(&ps, n_plant * sizeof(int64_t)));
        CHIP_HIP(c, hipMalloc(&pk, n_plant * sizeof(int32_t)));
        CHIP_HIP(c, hipMemcpyAsync(pd, plant_dst, n_plant * sizeof(int64_t), hipMemcpyHostToDevice, c->s_append));
        CHIP_HIP(c, hipMemcpyAsync(ps, plant_src, n_plant * sizeof(int64_t), hipMemcpyHostToDevice, c->s_append));
        CHIP_HIP(c, hipMemcpyAsync(pk, plant_kind, n_plant * sizeof(int32_t), hipMemcpyHostToDevice, c->s_append));
    }
    if (c->ring_dev) {   // same ordering against in-flight scans as append_impl
        hipEvent_t ev[2];
        {
            std::lock_guard<std::mutex> rl(c->ring_mu);
            c->rows_pending = first + n;
            ev[0] = c->last_scan_ev[0];
            ev[1] = c->last_scan_ev[1];
        }
        for (hipEvent_t e : ev)
            if (e) (void)hipStreamWaitEvent(c->s_append, e, 0);
    }
    rc = launch_synth(c, c->s_append, first, n, seed, pd, ps, pk, n_plant);
    hipError_t e = hipStreamSynchronize(c->s_append);
    if (pd) (void)hipFree(pd);
    if (ps) (void)hipFree(ps);
    if (pk) (void)hipFree(pk);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, e);
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->rows_global = first + n;
        c->rows_local = local_count(c, c->rows_global);
    }
    return CHIP_OK;
}

// ------------------------------------------------------------------------------------------------ queries
static int check_query_args(chip_ctx *c, int64_t k, int32_t nq, int32_t topk, int64_t *n_global)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (nq < 1 || nq > CHIP_MAX_NQ || topk < 1 || topk > CHIP_MAX_TOPK) return CHIP_ERR_UNSUPPORTED;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        *n_global = c->rows_global;
    }
    if (k < 0 || k > *n_global) return CHIP_ERR_RANGE;
    return CHIP_OK;
}

int chip_query_rows(chip_ctx *c, int64_t k, const int64_t *query_rows, int32_t nq, int32_t topk, double *scores, int64_t *idx)
{
    int64_t n = 0;
    int rc = check_query_args(c, k, nq, topk, &n);
    if (rc != CHIP_OK) return rc;
    if (!query_rows) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    const float *q[CHIP_MAX_NQ];
    RingGuard rg(c);
    rc = query_row_ptrs(c, query_rows, nq, n, q);
    if (rc != CHIP_OK) return rc;
    rc = enqueue_scan_merge(c, k, q, nq, topk, 0, nullptr, c->topk_dev, nullptr);
    if (rc != CHIP_OK) return rc;
    return sync_topk_out(c, nq, topk, scores, idx);
}

int chip_query_vectors_f32(chip_ctx *c, int64_t k, const float *queries, int32_t nq, int32_t topk, double *scores, int64_t *idx)
{
    int64_t n = 0;
    int rc = check_query_args(c, k, nq, topk, &n);
    if (rc != CHIP_OK) return rc;
    if (!queries) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    // on the scan stream: K1 reads qvec_dev there (the previous synchronous call has fully drained, so no WAR hazard)
    CHIP_HIP(c, hipMemcpyAsync(c->qvec_dev, queries, (size_t)nq * c->D * sizeof(float), hipMemcpyHostToDevice, c->s_scan));
    const float *q[CHIP_MAX_NQ];
    for (int i = 0; i < nq; i++) q[i] = c->qvec_dev + (size_t)i * c->D;
    rc = enqueue_scan_merge(c, k, q, nq, topk, 0, nullptr, c->topk_dev, nullptr);
    if (rc != CHIP_OK) return rc;
    return sync_topk_out(c, nq, topk, scores, idx);
}

// ------------------------------------------------------------------------------------------------ tick
int chip_loop_tick(chip_ctx *c, int64_t l, const chip_dot_params *p, chip_tick_result *out)
{
    if (!c || !p || !out) return CHIP_ERR_INVALID_ARG;
    if (c->nranks != 1) return CHIP_ERR_UNSUPPORTED;  // sharded ctx: chip_scan_local + chip_merge_decide
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    Slot &s = c->slots[CHIP_MAX_INFLIGHT - 1];
    int rc = tick_enqueue_slot(c, l, p, s);
    if (rc != CHIP_OK) return rc;
    return tick_collect_slot(c, s, out);
}

int chip_loop_tick_enqueue(chip_ctx *c, int64_t l, const chip_dot_params *p, int32_t slot)
{
    if (!c || !p || slot < 0 || slot >= CHIP_MAX_INFLIGHT - 1) return CHIP_ERR_INVALID_ARG;
    if (c->nranks != 1) return CHIP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    return tick_enqueue_slot(c, l, p, c->slots[slot]);
}

int chip_loop_tick_collect(chip_ctx *c, int32_t slot, chip_tick_result *out)
{
    if (!c || !out || slot < 0 || slot >= CHIP_MAX_INFLIGHT - 1) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    return tick_collect_slot(c, c->slots[slot], out);
}

int64_t chip_loop_last_l(const chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    return c->last_l;
}
void chip_loop_reset(chip_ctx *c)
{
    if (!c) return;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    c->last_l = 0;
}

int chip_scan_local(chip_ctx *c, int64_t l, const chip_dot_params *p, int32_t topk, void *dev_out, int32_t *status)
{
    if (!c || !p || !dev_out || !status) return CHIP_ERR_INVALID_ARG;
    if (topk < 1 || topk > CHIP_MAX_TOPK) return CHIP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    int64_t k = 0;
    int rc = tick_prepare(c, l, p, status, &k);
    if (rc == CHIP_OK && *status == CHIP_TICK_TOO_SHORT) c->last_l = l;
    if (rc != CHIP_OK || *status != CHIP_TICK_SCANNED) return rc;
    const int64_t rows[3] = {l - 1, l - 2, l - 3};
    const float *q[3];
    RingGuard rg(c);
    rc = query_row_ptrs(c, rows, 3, l, q);
    if (rc != CHIP_OK) return rc;
    // scan on s_scan, then (behind an event) the local merge on the ctx stream writes this rank's 3 x topk list to
    // dev_out: everything the caller enqueues next on the ctx stream (the all-gather) is ordered after it, while the
    // next tick's scan is free to start as soon as this scan ends.
    rc = enqueue_scan_merge(c, k, q, 3, topk, l, nullptr, (chip_topk_entry *)dev_out, nullptr, true);
    if (rc == CHIP_OK) c->last_l = l;  // :1098
    return rc;
}

static int merge_enqueue_slot(Ctx *c, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists, int32_t topk, Slot &s)
{
    if (s.in_flight) return CHIP_ERR_BUSY;
    MergeArgs m;
    m.in = (const chip_topk_entry *)dev_gathered;
    m.n_lists = n_lists;
    m.K = topk;
    m.out = nullptr;
    m.result = s.dev;   // pinned + mapped: no D2H copy
    m.l = l;
    m.locality = p->locality;
    m.thresh = p->thresh;
    int rc = launch_merge(c, c->s_query, m, 3);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, hipEventRecord(s.done, c->s_query));
    s.immediate = false;
    s.in_flight = true;
    return CHIP_OK;
}

int chip_merge_decide(chip_ctx *c, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists,
                      int32_t topk, chip_tick_result *out)
{
    if (!c || !p || !dev_gathered || !out || n_lists < 1) return CHIP_ERR_INVALID_ARG;
    if (topk < 1 || topk > CHIP_MAX_TOPK) return CHIP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    Slot &s = c->slots[CHIP_MAX_INFLIGHT - 1];
    int rc = merge_enqueue_slot(c, l, p, dev_gathered, n_lists, topk, s);
    if (rc != CHIP_OK) return rc;
    return tick_collect_slot(c, s, out);
}

int chip_merge_decide_enqueue(chip_ctx *c, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists,
                              int32_t topk, int32_t slot)
{
    if (!c || !p || !dev_gathered || n_lists < 1 || slot < 0 || slot >= CHIP_MAX_INFLIGHT - 1) return CHIP_ERR_INVALID_ARG;
    if (topk < 1 || topk > CHIP_MAX_TOPK) return CHIP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    return merge_enqueue_slot(c, l, p, dev_gathered, n_lists, topk, c->slots[slot]);
}

// ------------------------------------------------------------------------------------------------ introspection
int chip_get_info(const chip_ctx *c, chip_info *info)
{
    if (!c || !info) return CHIP_ERR_INVALID_ARG;
    std::memset(info, 0, sizeof *info);
    std::lock_guard<std::mutex> lk(c->mu);
    info->abi_version = CHIP_ABI_VERSION;
    info->D = c->D;
    info->device = c->device;
    info->shard_rank = c->rank;
    info->shard_count = c->nranks;
    info->n_cus = c->n_cus;
    info->rows_global = c->rows_global;
    info->rows_local = c->rows_local;
    info->capacity_local = (int64_t)c->segs.size() * c->seg_rows;
    info->lossy_rows = c->lossy_rows;
    std::strncpy(info->arch, c->arch, sizeof(info->arch) - 1);
    return CHIP_OK;
}

int chip_profile_enable(chip_ctx *c, int32_t on)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    c->prof_on = on != 0;
    return CHIP_OK;
}

int chip_profile_reset(chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    c->prof_used = 0;
    return CHIP_OK;
}

int chip_profile_scan(chip_ctx *c, double *total_ms, int64_t *n_launches, double *bytes_per_launch_last, double *span_ms)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(c->s_scan));
    if (c->s_scan2) CHIP_HIP(c, hipStreamSynchronize(c->s_scan2));
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    double tot = 0.0, span = 0.0;
    for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
        float ms = 0.f;
        CHIP_HIP(c, hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]));
        tot += ms;
        CHIP_HIP(c, hipEventElapsedTime(&ms, c->prof_ev[0], c->prof_ev[i + 1]));  // launches alternate between two streams
        if (ms > span) span = ms;
    }
    if (total_ms) *total_ms = tot;
    if (n_launches) *n_launches = (int64_t)(c->prof_used / 2);
    if (bytes_per_launch_last) *bytes_per_launch_last = c->prof_bytes_last;
    if (span_ms) *span_ms = span;
    return CHIP_OK;
}

}  // extern "C"
