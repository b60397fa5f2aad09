// chip_api.hip -- C-ABI entry points of libcerebro_hip.so (see include/cerebro_hip.h for the contract and
// the reference lines each entry point replaces).  Host-side C++ over the HIP runtime; no torch, no CPU
// fallback: every compute entry point runs HIP kernels or fails with a CHIP_ERR_* status.
//
// This file holds the single-device context.  chip_multi.hip builds on its building blocks (declared in
// chip_internal.h): the in-library exchange of per-shard top-k lists over RCCL, and group contexts that drive the
// GPUs of one node from one process.  Entry points dispatch on ctx->group.
#include "chip_internal.h"
#include <cmath>
#include <cstdlib>
#include <chrono>
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
        case CHIP_ERR_COMM: return "RCCL error (see chip_last_comm_error)";
        case CHIP_ERR_SHARD_FAILED: return "a shard could not take part in this tick / query (all ranks see this status); retry";
        case CHIP_ERR_GROUP_BROKEN: return "multi-GPU ctx is broken by an earlier partial failure; destroy it";
    }
    return "unknown status";
}

int chip_abi_version(void) { return CHIP_ABI_VERSION; }
int chip_build_scan_forms(void) { return scan_forms_built(); }
int chip_build_test_hooks(void)
{
#ifdef CHIP_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

int chip_last_hip_error(const chip_ctx *ctx, const char **text)
{
    if (!ctx) return (int)hipErrorInvalidValue;
    const Ctx *c = ctx->group ? group_root(const_cast<chip_ctx *>(ctx)) : ctx;
    if (text) *text = hipGetErrorString(c->last_hip);
    return (int)c->last_hip;
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
    p->sampler = CHIP_SAMPLER_FRESH;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ internals
namespace chip {

int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

// ------------------------------------------------------------------------------------------------ resident scan instance
static inline void host_store_fence()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_sfence();                        // drains the write-combining buffers of the BAR mapping
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
// (CHIP_TICK_RESIDENT=1; kernels.hip db_scan_resident, chip_internal.h ResidentCmd).  Host side: one command at a time; the line is
// written head first, tail last (x86 stores stay in order; the fences keep the compiler from moving them).
static void resident_write_line(Ctx *c, const ResidentCmd &cmd)
{
    if (c->res_direct) {
        // every workgroup's own line, through the write-combining BAR mapping: eight 8-byte stores per line in address order (head first,
        // tail last; a full 64-byte line leaves the CPU as one PCIe write), one fence for all of them
        ResidentCmd t = cmd;
        t.tail = t.head;
        uint64_t w[8];
        std::memcpy(w, &t, 64);
        volatile uint64_t *d = reinterpret_cast<volatile uint64_t *>(c->res_cmd_host);
        int first = 0;
#ifdef CHIP_TEST_HOOKS   // test build only (make testlibs -> cerebro_amd/lib/hooks/): the product library has no fault injection
        // CHIP_TEST_RESIDENT_SKIP_MASTER=k: the k-th command never reaches workgroup 0's line -- the other workgroups run it, workgroup 0
        // leaves at its lease without it: exactly what a command posted in the moment the lease runs out can look like (tests drive
        // resident_collect's recovery with it)
        if (cmd.n_rows >= 0 && c->res_test_skip_master > 0 && c->res_ticks + 1 == c->res_test_skip_master && !c->res_test_skipped) {
            first = 1;
            c->res_test_skipped = true;
        }
#endif
        for (int j = first; j < c->res_grid; j++)
            for (int i = 0; i < 8; i++) d[j * 8 + i] = w[i];
        host_store_fence();
        return;
    }
    ResidentCmd *d = c->res_cmd_host;
    __atomic_store_n(&d->head, cmd.head, __ATOMIC_RELAXED);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    d->locality = cmd.locality;
    d->n_rows = cmd.n_rows;
    d->tick_l = cmd.tick_l;
    d->thresh = cmd.thresh;
    d->result = cmd.result;
    d->seq_ptr = cmd.seq_ptr;
    d->seq_val = cmd.seq_val;
    d->dyn_claim = cmd.dyn_claim;
    __atomic_store_n(&d->tail, cmd.head, __ATOMIC_RELEASE);
    // a line in device memory is written through the write-combining BAR mapping: push it out (one full 64-byte line = one PCIe write)
    if (c->res_cmd_in_vram) host_store_fence();
}

static uint32_t resident_next_number(Ctx *c)
{
    uint32_t n = c->res_cmd_no + 1;
    if (n == 0u || n == kResidentLeave) n = 1u;
    c->res_cmd_no = n;
    return n;
}

static void resident_free(Ctx *c)
{
    if (c->s_resident) { (void)hipStreamDestroy(c->s_resident); c->s_resident = nullptr; }
    if (c->res_pinned) { (void)hipHostFree(c->res_pinned); c->res_pinned = nullptr; }
    if (c->res_cmd_vram) { (void)hipFree(c->res_cmd_vram); c->res_cmd_vram = nullptr; }
    if (c->res_cmd_dev) { (void)hipFree(c->res_cmd_dev); c->res_cmd_dev = nullptr; }
    if (c->res_partial) { (void)hipFree(c->res_partial); c->res_partial = nullptr; }
    if (c->res_ticket) { (void)hipFree(c->res_ticket); c->res_ticket = nullptr; }
    c->res_cmd_host = nullptr; c->res_cmd_hostdev = nullptr; c->res_exit_host = nullptr; c->res_exit_hostdev = nullptr;
    c->res_cmd_in_vram = false; c->res_direct = false;
    c->res_ready = false;
}

// does a pattern the CPU stores through the BAR mapping of `v` (64 bytes of device memory) read back through a copy?  Leaves zeros.
static bool bar_write_probe(void *v)
{
    if (hipMemset(v, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
    uint32_t back[16] = {0};
    volatile uint32_t *w = static_cast<volatile uint32_t *>(v);
    for (int i = 0; i < 16; i++) w[i] = 0x5eed0000u + (uint32_t)i;
    host_store_fence();
    bool ok = hipMemcpy(back, v, 64, hipMemcpyDeviceToHost) == hipSuccess;
    for (int i = 0; ok && i < 16; i++) ok = back[i] == 0x5eed0000u + (uint32_t)i;
    for (int i = 0; i < 16; i++) w[i] = 0u;
    host_store_fence();
    return ok;
}

static int resident_alloc_body(Ctx *c)
{
    // A stream of the HIGHEST priority: the runtime multiplexes streams onto a few hardware queues per priority level, and a packet
    // behind the resident kernel in its queue would wait for the instance to leave (measured: a tick stream that shared the queue
    // stalled for the whole lease).  The ctx's other streams are of the default priority, so this one gets a queue to itself.
    int pr_lo = 0, pr_hi = 0;
    CHIP_HIP(c, hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
    CHIP_HIP(c, hipStreamCreateWithPriority(&c->s_resident, hipStreamNonBlocking, pr_hi));
    void *h = nullptr;
    CHIP_HIP(c, hipHostMalloc(&h, 128, hipHostMallocDefault));   // the command line + the line of the exit word
    std::memset(h, 0, 128);
    c->res_pinned = h;
    c->res_cmd_host = static_cast<ResidentCmd *>(h);
    c->res_exit_host = reinterpret_cast<unsigned long long *>(static_cast<char *>(h) + 64);
    void *hd = nullptr;
    CHIP_HIP(c, hipHostGetDevicePointer(&hd, h, 0));
    c->res_cmd_hostdev = static_cast<uint32_t *>(hd);
    c->res_exit_hostdev = reinterpret_cast<unsigned long long *>(static_cast<char *>(hd) + 64);
    // Where the command line lives.  With a large PCIe BAR the host can store straight into device memory, and a posted write that the
    // device then finds locally beats a line the device has to fetch over PCIe (ping-pong with one polling wave: 1.89 us against 2.52,
    // scripts/probes/bar_pingpong.hip).  Used when the device says so AND a pattern written that way into THE buffer itself reads back
    // through a copy (probing a throw-away allocation says nothing about this one); CHIP_RESIDENT_BAR=0 keeps the line in
    // pinned host memory.
    // CHIP_RESIDENT_BAR=2 (the default) goes one step further: the host writes EVERY workgroup's line that way (16 KiB per command,
    // 0.6 us of host time) and nobody relays: 10k rows 35.9-36.4 -> 34.7-35.0 us.  1: only workgroup 0's line, relayed on the device.
    c->res_grid = c->n_cus * (c->scan_short_bpc > 0 ? c->scan_short_bpc : 1);
    if (c->res_grid > c->max_grid) c->res_grid = c->max_grid;
    CHIP_HIP(c, hipMalloc((void **)&c->res_cmd_dev, (size_t)c->max_grid * 64));
    CHIP_HIP(c, hipMemset(c->res_cmd_dev, 0, (size_t)c->max_grid * 64));
    const int bar_mode = env_int("CHIP_RESIDENT_BAR", 2);
    int large_bar = 0;
    if (bar_mode != 0 && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, c->device) == hipSuccess && large_bar) {
        if (bar_mode == 2) {
            // the workgroups' lines themselves are the host's target (line 0 is workgroup 0's): probe the first and the last line of them
            if (bar_write_probe(c->res_cmd_dev) && bar_write_probe(reinterpret_cast<char *>(c->res_cmd_dev) + (size_t)(c->max_grid - 1) * 64)) {
                c->res_cmd_in_vram = true;
                c->res_direct = true;
                c->res_cmd_host = reinterpret_cast<ResidentCmd *>(c->res_cmd_dev);
                c->res_cmd_hostdev = c->res_cmd_dev;
            }
        }
        if (!c->res_direct) {                   // mode 1, or mode 2 whose lines are not host-writable: one line of its own, relayed on the device
            void *v = nullptr;
            if (hipMalloc(&v, 64) == hipSuccess) {
                if (bar_write_probe(v)) {
                    c->res_cmd_vram = v;
                    c->res_cmd_in_vram = true;
                    c->res_cmd_host = static_cast<ResidentCmd *>(v);
                    c->res_cmd_hostdev = static_cast<uint32_t *>(v);
                } else {
                    (void)hipFree(v);
                }
            }
        }
    }
    CHIP_HIP(c, hipMalloc((void **)&c->res_partial, (size_t)c->max_grid * CHIP_MAX_NQ * CHIP_MAX_TOPK * sizeof(chip_topk_entry)));
    CHIP_HIP(c, hipMalloc((void **)&c->res_ticket, 64));
    return CHIP_OK;
}

// res_mu held.  Either everything exists afterwards (res_ready, published last) or nothing does and the mode is off for this ctx: the
// caller falls back to the launched tick (a half-built state must not be entered by the next tick).
static int resident_alloc(Ctx *c)
{
    if (c->res_ready) return CHIP_OK;
    const int rc = resident_alloc_body(c);
    if (rc != CHIP_OK) {
        resident_free(c);
        c->tick_resident = false;
        std::fprintf(stderr, "[cerebro_hip] resident scan instance: allocation failed (%s) -- this ctx launches its ticks\n", chip_strerror(rc));
        return CHIP_ERR_UNSUPPORTED;
    }
    c->res_ready = true;
    return CHIP_OK;
}

// A new instance behind whatever the stream still holds (the previous one on its way out).  res_mu held.
static int resident_launch(Ctx *c)
{
    ResidentArgs ra;
    ScanArgs &a = ra.base;
    a.seg_table = c->seg_table_dev;
    a.seg_shift = c->seg_shift;
    a.seg_mask = c->seg_rows - 1;
    a.n_rows = 0;
    a.D = c->D;
    a.K = 1;
    for (int i = 0; i < CHIP_MAX_NQ; i++) a.q[i] = nullptr;
    a.idx_mul = 1;
    a.idx_add = 0;
    a.partial = c->res_partial;
    a.rows_form = 1;
    a.plain_loads = 1;
    a.fused_ticket = c->res_ticket;
    a.stamps = c->stamps_dev;            // tuning only (CHIP_SCAN_STAMPS=1)
    ra.cmd_host = c->res_cmd_hostdev;
    ra.cmd_dev = c->res_cmd_dev;
    ra.exit_host = c->res_exit_hostdev;
    ra.instance = ++c->res_instance;
    ra.lease_ticks = (unsigned long long)c->res_lease_ms * 100000ull;   // s_memrealtime: 100 MHz
    ra.done = c->res_done;
    ra.direct = c->res_direct ? 1u : 0u;
    // the workgroups' lines still hold what the previous instance was told last (its leave mark included), the ticket whatever it left.
    // (direct mode: the lines are the host's to write -- the caller has posted the command the new instance is to find there)
    if (!c->res_direct) CHIP_HIP(c, hipMemsetAsync(c->res_cmd_dev, 0, (size_t)c->max_grid * 64, c->s_resident));
    CHIP_HIP(c, hipMemsetAsync(c->res_ticket, 0, 64, c->s_resident));
    int rc = launch_resident(c, c->s_resident, ra, c->res_grid);
    if (rc != CHIP_OK) return rc;
    c->res_alive = true;
    c->res_launches++;
    return CHIP_OK;
}

static bool resident_has_left(const Ctx *c)
{
    return __atomic_load_n(c->res_exit_host, __ATOMIC_ACQUIRE) == c->res_instance;
}

static void resident_stop_locked(Ctx *c)
{
    if (!c->res_ready || !c->res_alive) return;
    if (c->res_busy && c->res_slot) {   // a command is still running: let it finish (its collector only reads the completion word)
        const volatile unsigned long long *w = c->res_slot->seq_host;
        for (long spin = 0; spin < (1L << 28) && __atomic_load_n(w, __ATOMIC_ACQUIRE) != c->res_slot->seq_want && !resident_has_left(c); spin++) {}
    }
    if (!resident_has_left(c)) {
        ResidentCmd cmd{};
        cmd.head = resident_next_number(c);
        cmd.n_rows = -1;
        resident_write_line(c, cmd);
        c->res_done = cmd.head;          // nobody runs a leave command twice
    }
    (void)hipStreamSynchronize(c->s_resident);
    c->res_alive = false;
}

void resident_stop(Ctx *c)
{
    if (!c->tick_resident) return;
    std::lock_guard<std::mutex> lk(c->res_mu);
    resident_stop_locked(c);
}

void resident_pause(Ctx *c)
{
    std::lock_guard<std::mutex> lk(c->res_mu);
    c->res_inhibit++;                    // from here on resident_tick_enqueue launches nothing (it checks under this mutex)
    resident_stop_locked(c);
}

void resident_resume(Ctx *c)
{
    std::lock_guard<std::mutex> lk(c->res_mu);
    if (c->res_inhibit > 0) c->res_inhibit--;
}

static bool resident_eligible(Ctx *c, int64_t k)
{
    if (!c->tick_resident || c->nranks != 1 || c->xchg || c->parent || c->group || !c->own_query_stream || c->prof_on || !c->tick_fused ||
        !c->tick_poll || c->res_busy)
        return false;
    // One workgroup per CU, so that the ctx's other kernels (append, PnP, ICP) still find registers and LDS next to the instance.  That
    // shape is what a launch gives a cache-sized prefix anyway, and a synchronous tick loses nothing with it up to the reference's own
    // capacity (29k rows: 93.9-94.7 us launched with one workgroup per CU, 93.8-94.1 with two; profiles/r05_resident.md).
    if ((double)k * c->D * c->elem > c->res_max_bytes) return false;
    const int grid = c->n_cus * (c->scan_short_bpc > 0 ? c->scan_short_bpc : 1);
    return scan_rows_form(c, k, 3, grid, false) == 1;
}

constexpr int kResidentPaused = 1;   // (internal, > 0: not a status code) the mode is paused, launch this tick

// The tick as a command to the resident instance (launched here if there is none, or if the last one has left).  query_mu held.
static int resident_tick_enqueue(Ctx *c, int64_t k, int64_t l, const chip_dot_params *p, Slot &s)
{
    std::lock_guard<std::mutex> lk(c->res_mu);
    if (c->res_inhibit > 0) return kResidentPaused;   // a section that frees / allocates / rewrites the table is open: this tick is launched
    int rc = resident_alloc(c);
    if (rc != CHIP_OK) return rc;
    if (c->res_alive && resident_has_left(c)) c->res_alive = false;
    const bool launch = !c->res_alive;
    if (launch && c->res_direct) (void)hipStreamSynchronize(c->s_resident);   // the old instance is gone before its lines are rewritten
    s.seq_want = ++c->tick_seq;
    ResidentCmd cmd{};
    cmd.head = resident_next_number(c);
    cmd.locality = p->locality;
    cmd.n_rows = k;
    cmd.tick_l = l;
    cmd.thresh = p->thresh;
    cmd.result = (uint64_t)(uintptr_t)s.dev;
    cmd.seq_ptr = (uint64_t)(uintptr_t)s.seq_dev;
    cmd.seq_val = s.seq_want;
    // rows claimed within the workgroup: beyond cache-sized prefixes, as for launches (29k rows: 92.3 -> 91.7 us; neutral at 10k)
    cmd.dyn_claim = (c->scan_claim == 1 || (c->scan_claim < 0 && (double)k * c->D * c->elem > c->scan_half_bytes)) ? 1u : 0u;   // (the instance runs the product stream only)
    c->res_pending = cmd;
    resident_write_line(c, cmd);           // (before the launch: an instance must never find the previous one's leave mark in its lines)
    if (launch) {
        rc = resident_launch(c);
        if (rc != CHIP_OK) return rc;
    }
    c->res_busy = true;
    c->res_slot = &s;
    c->res_ticks++;
    s.poll = true;
    s.resident = true;
    s.wait_ev = nullptr;
    return CHIP_OK;
}

// Wait for the command's completion word.  An instance that left before it saw the command (its lease ran out just then) is replaced:
// the new one finds the line and runs it.
static int resident_collect(Ctx *c, Slot &s)
{
    const volatile unsigned long long *w = s.seq_host;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spin = 1;; spin++) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == s.seq_want) break;
        if ((spin & 0x3ff) == 0) {
            std::lock_guard<std::mutex> lk(c->res_mu);
            if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == s.seq_want) break;
            if (!c->res_alive || resident_has_left(c)) {   // left by its lease, or retired by another thread (resident_stop) before it saw the line
                c->res_alive = false;
                if (c->res_direct) {
                    // some workgroups may have found the command in their lines while workgroup 0 was already leaving: send everybody
                    // home, wait until the instance is gone, then post the command again for a new one
                    ResidentCmd bye{};
                    bye.head = bye.tail = kResidentLeave;
                    bye.n_rows = -1;
                    resident_write_line(c, bye);
                    (void)hipStreamSynchronize(c->s_resident);
                    if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == s.seq_want) break;   // (it had run after all)
                }
                // (one line for all: it may hold the leave command of a resident_stop that gave up waiting for this one)
                resident_write_line(c, c->res_pending);
                const int rc = resident_launch(c);
                if (rc != CHIP_OK) { c->res_busy = false; c->res_slot = nullptr; s.resident = false; return rc; }
            }
            if ((spin & 0xfffff) == 0 && std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - t0).count() > 30) {
                c->last_hip = hipErrorLaunchTimeOut;
                c->res_busy = false;          // whatever state the instance is in, its lease ends it; later ticks may try again
                c->res_slot = nullptr;
                s.resident = false;
                return CHIP_ERR_HIP;
            }
        }
    }
    std::lock_guard<std::mutex> lk(c->res_mu);
    c->res_busy = false;
    c->res_slot = nullptr;
    c->res_done = c->res_cmd_no;
    s.resident = false;
    return CHIP_OK;
}

static int ensure_capacity(Ctx *c, int64_t local_rows)
{
    // caller holds append_mu
    const int64_t need = (local_rows + c->seg_rows - 1) >> c->seg_shift;
    if (need > kMaxSegs) return CHIP_ERR_OOM;
    bool grew = false;
    // the segment table is about to change: no instance from here until the table is final (an instance reads the table through
    // caches nothing invalidates while it lives; rows of the new segment are published only after this function returns)
    ResidentPause paused(c, c->tick_resident && (int64_t)c->segs.size() < need);
    while ((int64_t)c->segs.size() < need) {
        void *p = nullptr;
        CHIP_HIP(c, hipMalloc(&p, (size_t)c->seg_rows * c->D * c->elem));
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->segs.push_back(p);
        }
        grew = true;
    }
    if (grew) {
        CHIP_HIP(c, hipMemcpyAsync(c->seg_table_dev, c->segs.data(), c->segs.size() * sizeof(void *), hipMemcpyHostToDevice, c->s_append));
        CHIP_HIP(c, hipStreamSynchronize(c->s_append));
    }
    return CHIP_OK;
}

// (Re)allocate everything whose size depends on the storage type.  Only ever called on an EMPTY DB (create, or the automatic
// switch to double rows on the first append), with the append lock held or before the ctx is published.
static int configure_storage(Ctx *c, int elem)
{
    ResidentPause paused(c, c->tick_resident);
    for (void *p : c->segs) (void)hipFree(p);
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->segs.clear();
    }
    if (c->ring_dev) { (void)hipFree(c->ring_dev); c->ring_dev = nullptr; }
    if (c->qvec_dev) { (void)hipFree(c->qvec_dev); c->qvec_dev = nullptr; }
    c->elem = elem;
    // segment geometry: power-of-two rows, ~512 MiB (float rows) / ~1 GiB (double rows) each
    int64_t rows = kSegBytesTarget / ((int64_t)c->D * 4);
    int shift = 0;
    while ((2ll << shift) <= rows) shift++;
    if (shift < 6) shift = 6;
    c->seg_shift = shift;
    c->seg_rows = 1ll << shift;
    CHIP_HIP(c, hipMemset(c->seg_table_dev, 0, kMaxSegs * sizeof(void *)));
    if (c->nranks > 1) {
        CHIP_HIP(c, hipMalloc(&c->ring_dev, (size_t)CHIP_RING_ROWS * c->D * elem));
        CHIP_HIP(c, hipMemset(c->ring_dev, 0, (size_t)CHIP_RING_ROWS * c->D * elem));
    }
    CHIP_HIP(c, hipMalloc(&c->qvec_dev, (size_t)CHIP_MAX_NQ * c->D * elem));
    return CHIP_OK;
}

void ctx_destroy(chip_ctx *c)
{
    if (!c) return;
    if (c->group) { group_destroy(c); delete c; return; }
    (void)hipSetDevice(c->device);
    resident_stop(c);
    (void)hipDeviceSynchronize();
    resident_free(c);
    exchange_destroy(c);
    pnp_destroy(c);
    icp_destroy(c);
    batch_destroy(c);
    for (void *p : c->segs) (void)hipFree(p);
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
    for (hipStream_t x : c->s_scan_x)
        if (x) (void)hipStreamDestroy(x);
    if (c->topk_host) (void)hipHostFree(c->topk_host);
    if (c->qvec_dev) (void)hipFree(c->qvec_dev);
    if (c->scores_dev) (void)hipFree(c->scores_dev);
    if (c->stamps_dev) (void)hipFree(c->stamps_dev);
    if (c->tickets_dev) (void)hipFree(c->tickets_dev);
    if (c->pair_ctr_dev) (void)hipFree(c->pair_ctr_dev);
    if (c->seq_host_all) (void)hipHostFree(c->seq_host_all);
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

static int create_impl(chip_ctx *c, int64_t capacity_hint, int elem)
{
    hipDeviceProp_t prop;
    CHIP_HIP(c, hipGetDeviceProperties(&prop, c->device));
    c->n_cus = prop.multiProcessorCount;
    std::strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
    if (std::strncmp(c->arch, "gfx950", 6) != 0 && !std::getenv("CHIP_ALLOW_ANY_ARCH")) return CHIP_ERR_NO_DEVICE;

    // The querier reads segs[] without c->mu (row_ptr_host) while the appender may open a new segment: reserve the whole
    // table (32 KiB) so that push_back never reallocates -- entries below the published length are immutable.
    c->segs.reserve(kMaxSegs);

    CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_query, hipStreamNonBlocking));
    CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_append, hipStreamNonBlocking));
    CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_pnp, hipStreamNonBlocking));
    CHIP_HIP(c, hipMalloc(&c->seg_table_dev, kMaxSegs * sizeof(void *)));
    int rc = configure_storage(c, elem);
    if (rc != CHIP_OK) return rc;
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
    c->scan_rows = env_int("CHIP_SCAN_ROWS", 0);
    c->scan_depth = env_int("CHIP_SCAN_DEPTH", 1);      // 1 = the product's claimed stream; 2..4: experimental forms (kernels.hip scan_rows_body)
    if (c->scan_depth < 1 || c->scan_depth > 7) c->scan_depth = 1;
    { const int st = env_int("CHIP_SCAN_STAGGER", 0); if (st > 0 && st < 4096) c->scan_depth |= st << 8; }   // tuning builds only (kernels.hip)
    c->scan_claim = env_int("CHIP_SCAN_CLAIM", -1);   // -1 = auto (full-occupancy launches of the row-batched kernel), 0 = never, 1 = always
    c->tick_same_stream = env_int("CHIP_TICK_SAME_STREAM", 1) != 0;
    c->scan_short_bpc = env_int("CHIP_SCAN_SHORT_BPC", 1);
    c->scan_plain_bytes = (double)env_int("CHIP_SCAN_PLAIN_MIB", 768) * 1024 * 1024;
    c->scan_half_bytes = (double)env_int("CHIP_SCAN_HALF_MIB", 192) * 1024 * 1024;
    c->scan_sync_plain_bytes = (double)env_int("CHIP_SCAN_SYNC_PLAIN_MIB", 4096) * 1024 * 1024;
    c->scan_overlap_bytes = (double)env_int("CHIP_SCAN_OVERLAP_GIB", 8) * 1024 * 1024 * 1024;
    // a sharded ctx gets three small kernels per tick through its ctx stream underneath the scans: keep slots free for them
    c->scan_reserve = env_int("CHIP_SCAN_RESERVE", c->nranks > 1 ? 4 : 0);
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
    {
        // Tick streams.  Short ticks of a plain ctx run scan + merge on ONE stream and rotate over up to four of them: a 10k-row tick
        // is ~37 us of scan, ~15 us of one-workgroup merge and ~12 us of launch / event gap per stream (rocprofv3 timeline,
        // profiles/r03_tick_timeline_10k.md), so two streams leave the memory system idle a third of the time.  Sharded / group
        // contexts (merge + exchange on the ctx stream) use the first two only.
        const int ns = env_int("CHIP_SCAN_STREAMS", c->nranks > 1 ? 2 : 4);
        if (ns >= 2) CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_scan2, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++)
            if (ns >= 3 + i) CHIP_HIP(c, hipStreamCreateWithFlags(&c->s_scan_x[i], hipStreamNonBlocking));
    }
    for (int i = 0; i < Ctx::kRing; i++) {
        CHIP_HIP(c, hipMalloc(&c->partial_dev[i], (size_t)c->max_grid * CHIP_MAX_NQ * CHIP_MAX_TOPK * sizeof(chip_topk_entry)));
        CHIP_HIP(c, hipEventCreateWithFlags(&c->ev_scan[i], hipEventDisableTiming));
        CHIP_HIP(c, hipEventCreateWithFlags(&c->ev_merged[i], hipEventDisableTiming));
    }
    CHIP_HIP(c, hipMalloc(&c->tickets_dev, Ctx::kRing * sizeof(int32_t)));
    CHIP_HIP(c, hipMemset(c->tickets_dev, 0, Ctx::kRing * sizeof(int32_t)));
    if ((c->scan_depth & 255) == 7) {   // tuning builds only (pair-claimed stream, profiles/r06_short_scan.md): the product allocates nothing for it
        CHIP_HIP(c, hipMalloc(&c->pair_ctr_dev, (size_t)Ctx::kRing * Ctx::kPairCtrs * Ctx::kPairStride * sizeof(uint32_t)));
        CHIP_HIP(c, hipMemset(c->pair_ctr_dev, 0, (size_t)Ctx::kRing * Ctx::kPairCtrs * Ctx::kPairStride * sizeof(uint32_t)));
    }
    c->tick_fused = env_int("CHIP_TICK_FUSED", 1) != 0;
    c->tick_poll = env_int("CHIP_TICK_POLL", 1) != 0;
    // opt-in: synchronous ticks over cache-sized prefixes go to a scan instance that stays on the chip (chip_internal.h ResidentCmd)
    c->tick_resident = env_int("CHIP_TICK_RESIDENT", 0) != 0 && (scan_forms_built() & CHIP_SCAN_FORM_ROWS) != 0;
    c->res_max_bytes = (double)env_int("CHIP_RESIDENT_MAX_MIB", 512) * 1024 * 1024;   // 32k rows of 4096 floats: the reference's capacity is 29k
    c->res_lease_ms = env_int("CHIP_RESIDENT_LEASE_MS", 250);
#ifdef CHIP_TEST_HOOKS
    c->res_test_skip_master = env_int("CHIP_TEST_RESIDENT_SKIP_MASTER", 0);
#endif
    if (c->res_lease_ms < 1) c->res_lease_ms = 1;
    CHIP_HIP(c, hipHostMalloc(&c->seq_host_all, sizeof(unsigned long long) * CHIP_MAX_INFLIGHT, hipHostMallocDefault));
    std::memset(c->seq_host_all, 0, sizeof(unsigned long long) * CHIP_MAX_INFLIGHT);
    CHIP_HIP(c, hipHostMalloc(&c->topk_host, (size_t)CHIP_MAX_NQ * CHIP_MAX_TOPK * sizeof(chip_topk_entry), hipHostMallocDefault));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&c->topk_dev, c->topk_host, 0));
    for (Slot &s : c->slots) {
        CHIP_HIP(c, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        // pinned + mapped: the deciding workgroup stores the record here directly, no D2H copy kernel per tick
        CHIP_HIP(c, hipHostMalloc(&s.host, sizeof(chip_tick_result), hipHostMallocDefault));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&s.dev, s.host, 0));
        s.seq_host = c->seq_host_all + (&s - c->slots);
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&s.seq_dev, s.seq_host, 0));
    }
    if (env_int("CHIP_SCAN_STAMPS", 0)) {
        CHIP_HIP(c, hipMalloc(&c->stamps_dev, ((size_t)c->max_grid * 16 * 4 + 64) * sizeof(unsigned long long)));   // (+ 8 launch-wide stamps behind the waves')
        CHIP_HIP(c, hipMemset(c->stamps_dev, 0, ((size_t)c->max_grid * 16 * 4 + 64) * sizeof(unsigned long long)));
    }
    rc = pnp_create(c);
    if (rc != CHIP_OK) return rc;
    c->cap_hint = capacity_hint;
    if (capacity_hint > 0) {
        std::lock_guard<std::mutex> lk(c->append_mu);
        rc = ensure_capacity(c, local_count(c, capacity_hint));
        if (rc != CHIP_OK) return rc;
    }
    return CHIP_OK;
}

int ctx_create(chip_ctx **out, int32_t D, int64_t capacity_hint, int32_t device, int32_t shard_rank, int32_t shard_count, uint32_t flags)
{
    if (!out) return CHIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (D <= 0 || shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count || capacity_hint < 0) return CHIP_ERR_INVALID_ARG;
    const uint32_t store = flags & kCreateStoreMask;
    if (store == kCreateStoreMask) return CHIP_ERR_INVALID_ARG;
    if (D % 4 != 0 || (size_t)D * 4 * CHIP_MAX_NQ > 160 * 1024) return CHIP_ERR_UNSUPPORTED;
    // double rows: two query descriptors (2 x D x 8 B) must fit the 160 KiB of LDS; further ones are read in place (db_scan_topk_wide).
    // D <= 10 240 for either storage type -- the reference's default 8192-D model included (Cerebro.cpp:1021)
    if (store == CHIP_CREATE_STORE_F64 && (size_t)D * 8 * 2 > 160 * 1024) return CHIP_ERR_UNSUPPORTED;
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
    c->store_auto = store == 0;
    int rc = create_impl(c, capacity_hint, store == CHIP_CREATE_STORE_F64 ? 8 : 4);
    if (rc != CHIP_OK) { ctx_destroy(c); return rc; }
    *out = c;
    return CHIP_OK;
}

// Enqueue K1 (scan + per-workgroup top-k, on s_scan) and K2 (cross-workgroup merge [+ accept decision], on the ctx
// stream behind an event) for nq queries over the global prefix [0,k).  out (device or pinned host, optional) gets
// [nq][K]; res (optional) the decision record of Cerebro.cpp:1056.  Consecutive calls pipeline: scans run back to
// back on s_scan while the previous merge (and whatever the caller enqueues after it on the ctx stream) proceeds.
int enqueue_scan_merge(Ctx *c, int64_t k, const void *const *q, int nq, int K, int64_t l,
                       const chip_dot_params *p, chip_topk_entry *out, chip_tick_result *res, bool tick, hipStream_t *merge_stream,
                       hipEvent_t *merged_ev)
{
    const int b = (int)(c->n_enqueued++ % Ctx::kRing);
    // Ticks (queries already resident) over a short prefix alternate between two scan streams so that the ramp-down of
    // one launch overlaps the ramp-up of the next (measured, 1 -> 2 streams: 10k rows 45 -> 38 us/tick, 60k 172 -> 153,
    // 125k 317 -> 308, 500k 1191 -> 1157; at 1M the gain is < 1 %, and launches that overlap would no longer have a
    // meaningful per-launch duration for the roofline, so long scans and profiled runs stay on one stream).  Anything that
    // uploads its queries on s_scan first stays on s_scan.
    const bool short_scan = (double)local_count(c, k) * c->D * c->elem <= c->scan_overlap_bytes;
    hipStream_t s_scan = (tick && short_scan && !c->prof_on && c->s_scan2 && (c->n_enqueued & 1)) ? c->s_scan2 : c->s_scan;
    // (same-stream short ticks rotate over all tick streams: see below)
    const bool same_stream = tick && short_scan && !c->prof_on && c->nranks == 1 && !c->xchg && !c->parent && c->own_query_stream &&
                             c->s_scan2 && merge_stream && c->tick_same_stream;
    if (same_stream) {
        hipStream_t ring[4] = {c->s_scan, c->s_scan2, c->s_scan_x[0], c->s_scan_x[1]};
        int ns = 2;
        while (ns < 4 && ring[ns]) ns++;
        s_scan = ring[c->n_same_stream++ % (uint64_t)ns];
    }
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
    a.q64 = scan_q64(c, nq, !short_scan) ? 1 : 0;
    const int grid = scan_grid_for(c, a.n_rows, nq, a.q64 != 0);
    a.rows_form = scan_rows_form(c, a.n_rows, nq, grid, a.q64 != 0, tick && c->tick_sync_now);
    a.plain_loads = (double)a.n_rows * c->D * c->elem <= c->scan_plain_bytes ? 1 : 0;
    a.stamps = c->stamps_dev;
    // rows claimed within the workgroup: measured -2 % on the synchronous 29k / 45k tick (two workgroups per CU), neutral at 10k (one
    // workgroup per CU: the launch is too short for the waves to drift apart) -- profiles/r05_short_scan.md
    a.dyn_claim = (a.rows_form == 1 && (c->scan_claim == 1 || (c->scan_claim < 0 && grid > c->n_cus))) ? c->scan_depth : 0;
    // fused tick: one launch (kernels.hip fused_tick_finish) -- same-stream short ticks through the row-batched kernel, decision wanted,
    // no list output
    const bool fused = same_stream && a.rows_form > 0 && c->tick_fused && res != nullptr && out == nullptr && nq == 3 && p != nullptr;
    if (fused) {
        a.K = 1;
        a.fused_result = res;
        a.fused_ticket = c->tickets_dev + b;
        // tuning builds, CHIP_SCAN_DEPTH=7 only: rows claimed by PAIRS of workgroups (b, b + grid / 2: the older and the younger workgroup of a CU) from one counter
        if ((a.dyn_claim & 255) == 7 && c->pair_ctr_dev && (grid & 15) == 0 && grid / 2 <= Ctx::kPairCtrs) a.pair_ctr = c->pair_ctr_dev + (size_t)b * Ctx::kPairCtrs * Ctx::kPairStride;
        a.tick_l = l;
        a.locality = p->locality;
        a.thresh = p->thresh;
        a.fused_seq = c->next_seq_dev;       // (nullptr unless tick_enqueue_slot asked for a pollable completion)
        a.fused_seq_val = c->next_seq_val;
    }
    c->last_enqueue_fused = fused;
    c->next_seq_dev = nullptr;

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
        c->prof_bytes_last = (double)a.n_rows * c->D * c->elem;
        CHIP_HIP(c, hipEventRecord(e0, s_scan));
    }
    int rc = launch_scan(c, s_scan, a, nq, grid);
    if (rc != CHIP_OK) return rc;
    if (e1) CHIP_HIP(c, hipEventRecord(e1, s_scan));
    // Short ticks of a plain single-GPU ctx keep the merge on the scan's OWN stream: the tick is then two launches and two event
    // records on one stream (no cross-stream event pair), and because consecutive ticks alternate between the two scan streams,
    // merge(i) still overlaps scan(i+1).  At 10k rows the tick is host-enqueue-bound otherwise (round 2: 21-34 us of API calls per
    // tick against a ~20 us kernel).  Anything with an exchange, an external ctx stream or profiling keeps the ctx-stream merge.
    hipStream_t s_merge = same_stream ? s_scan : c->s_query;
    if (merge_stream) *merge_stream = s_merge;
    if (!same_stream) {
        CHIP_HIP(c, hipEventRecord(c->ev_scan[b], s_scan));
        if (c->ring_dev) c->last_scan_ev[s_scan == c->s_scan2 ? 1 : 0] = c->ev_scan[b];   // caller holds ring_mu (RingGuard)
        CHIP_HIP(c, hipStreamWaitEvent(c->s_query, c->ev_scan[b], 0));
    }

    if (fused) {   // the scan's last workgroup has written the decision record: the scan's end IS the tick's end
        CHIP_HIP(c, hipEventRecord(c->ev_merged[b], s_scan));
        if (merged_ev) *merged_ev = c->ev_merged[b];
        return CHIP_OK;
    }
    MergeArgs m;
    m.in = c->partial_dev[b];
    m.n_lists = grid;
    m.K = K;
    m.out = out;
    m.result = res;
    m.l = l;
    m.locality = p ? p->locality : 0;
    m.thresh = p ? p->thresh : 0.0;
    rc = launch_merge(c, s_merge, m, nq);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, hipEventRecord(c->ev_merged[b], s_merge));
    if (merged_ev) *merged_ev = same_stream ? c->ev_merged[b] : nullptr;   // same-stream tick: this IS the tick's completion event
    return CHIP_OK;
}

// Pointers of the query rows (device).  Single GPU: straight into the DB; sharded: the replicated ring.
// Sharded ctx: the query rows of a scan are read from the replicated ring, which the appender overwrites in place.  A
// querier holds ring_mu from the residency check until the scan's completion event is recorded (RingGuard); the appender
// announces the length its call will reach (rows_pending) and picks up the newest scan events under the same lock before it
// touches the ring, and its ring writes wait for those scans.  So a scan either was validated against the post-append
// length, or is ordered before the ring writes.
int query_row_ptrs(Ctx *c, const int64_t *rows, int nq, int64_t n_global, const void **q)
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
            q[i] = ring_ptr(c, g);
        }
    }
    return CHIP_OK;
}

// External query vectors -> qvec_dev in the storage type, on the scan stream: K1 reads them there (the previous synchronous
// call has fully drained, so no WAR hazard).  float <-> double conversion of a few rows is done on the host: widening is
// exact; narrowing double queries for a float DB must be lossless ((double)(float)x == x) or the call fails.
int upload_query_vectors(Ctx *c, const void *queries, int src_elem, int nq, const void **q)
{
    const size_t n = (size_t)nq * c->D;
    if (src_elem == c->elem) {
        CHIP_HIP(c, hipMemcpyAsync(c->qvec_dev, queries, n * c->elem, hipMemcpyHostToDevice, c->s_scan));
    } else {
        c->qconv.resize(n * c->elem);
        if (c->elem == 8) {
            double *d = reinterpret_cast<double *>(c->qconv.data());
            const float *s = static_cast<const float *>(queries);
            for (size_t i = 0; i < n; i++) d[i] = (double)s[i];
        } else {
            float *d = reinterpret_cast<float *>(c->qconv.data());
            const double *s = static_cast<const double *>(queries);
            for (size_t i = 0; i < n; i++) {
                d[i] = (float)s[i];
                if (!((double)d[i] == s[i])) return std::isfinite(s[i]) ? CHIP_ERR_NOT_F32 : CHIP_ERR_NONFINITE;
            }
        }
        CHIP_HIP(c, hipMemcpyAsync(c->qvec_dev, c->qconv.data(), n * c->elem, hipMemcpyHostToDevice, c->s_scan));
        CHIP_HIP(c, hipStreamSynchronize(c->s_scan));   // qconv is pageable and reused
    }
    for (int i = 0; i < nq; i++) q[i] = static_cast<char *>(c->qvec_dev) + (size_t)i * c->D * c->elem;
    return CHIP_OK;
}

// Host part of the tick (Cerebro.cpp:960-966, :1019-1022).  Returns CHIP_TICK_* in *status.  last_l is NOT written here:
// the caller commits it once the tick has actually been enqueued (or at once for TOO_SHORT), so a failed enqueue leaves the
// state as the reference's loop would (it only reaches :1098 at the end of an executed pass).
int tick_prepare(int64_t n, int64_t last_l, int64_t l, const chip_dot_params *p, int32_t *status, int64_t *k_out)
{
    if (l < 0 || l > n) return CHIP_ERR_RANGE;
    if (l - last_l < p->min_new) { *status = CHIP_TICK_SKIPPED; return CHIP_OK; }  // :962-966, last_l untouched
    if (l < 3) return CHIP_ERR_RANGE;  // needs descriptors l-1, l-2, l-3 (:987-989)
    const int64_t k = l - p->lag;      // :1019
    *k_out = k;
    *status = (k > p->min_k) ? CHIP_TICK_SCANNED : CHIP_TICK_TOO_SHORT;  // :1022
    return CHIP_OK;
}

void fill_immediate(chip_tick_result *r, int32_t status)
{
    std::memset(r, 0, sizeof *r);
    r->status = status;
    r->idx_curr = r->idx_prev = -1;
    for (int q = 0; q < 3; q++) { r->argmax[q] = -1; r->maxv[q] = -INFINITY; }
}

int64_t published_rows(const Ctx *c)
{
    std::lock_guard<std::mutex> lk(c->mu);
    return c->rows_global;
}

static int tick_enqueue_slot(Ctx *c, int64_t l, const chip_dot_params *p, Slot &s)
{
    if (s.in_flight) return CHIP_ERR_BUSY;
    int32_t status = 0;
    int64_t k = 0;
    // With an exchange attached the ranks may see different published lengths (each has its own appender): l > n on THIS rank is
    // then not a reason to leave the collective call -- the rank takes part with the failure mark (xchg_tick_enqueue).
    const int64_t n_pub = published_rows(c);
    int rc = tick_prepare(c->xchg ? INT64_MAX : n_pub, c->last_l, l, p, &status, &k);
    if (rc != CHIP_OK) return rc;
    s.poll = false;
    s.prev_last_l = c->last_l;
    s.tick_l = l;
    s.last_l_ptr = &c->last_l;
    if (status != CHIP_TICK_SCANNED) {
        if (status == CHIP_TICK_TOO_SHORT) c->last_l = l;   // :1098 (the else-branch of :1022 still ends the pass)
        fill_immediate(s.host, status);
        s.immediate = true;
        s.in_flight = true;
        return CHIP_OK;
    }
    if (c->xchg) {   // sharded ctx with its exchange inside the library: scan -> local merge -> all-gather -> merge + decision
        rc = xchg_tick_enqueue(c, l, k, p, s, l > n_pub);
        if (rc != CHIP_OK) return rc;
    } else {
        const int64_t rows[3] = {l - 1, l - 2, l - 3};  // v, vm, vmm (:987-989)
        const void *q[3];
        RingGuard rg(c);
        rc = query_row_ptrs(c, rows, 3, l, q);
        if (rc != CHIP_OK) return rc;
        if (resident_eligible(c, k)) {
            rc = resident_tick_enqueue(c, k, l, p, s);
            if (rc == CHIP_OK) {
                s.immediate = false;
                s.in_flight = true;
                c->last_l = l;             // :1098
                return CHIP_OK;
            }
            if (rc != CHIP_ERR_UNSUPPORTED && rc != kResidentPaused) return rc;
            if (rc == CHIP_ERR_UNSUPPORTED) c->tick_resident = false;      // this ctx's shape has no resident form: every tick is launched from here on
            s.resident = false;
            s.poll = false;
        }
        hipStream_t s_done = c->s_query;
        hipEvent_t merged = nullptr;
        s.poll = false;
        if (c->tick_poll) { s.seq_want = ++c->tick_seq; c->next_seq_dev = s.seq_dev; c->next_seq_val = s.seq_want; }
        rc = enqueue_scan_merge(c, k, q, 3, CHIP_DEFAULT_TOPK, l, p, nullptr, s.dev, true, &s_done, &merged);
        c->next_seq_dev = nullptr;
        if (rc != CHIP_OK) return rc;
        s.poll = c->tick_poll && c->last_enqueue_fused;   // the fused kernel stores the completion word itself
        // a same-stream tick is complete when its merge is: one event record per tick (the list buffer's merge event; the buffer is
        // not reused before kRing = 64 further enqueues, and at most CHIP_MAX_INFLIGHT - 1 = 63 ticks are uncollected)
        if (merged) s.wait_ev = merged;
        else { CHIP_HIP(c, hipEventRecord(s.done, s_done)); s.wait_ev = s.done; }
        s.immediate = false;
        s.in_flight = true;
    }
    c->last_l = l;                     // :1098
    return CHIP_OK;
}

int tick_collect_slot(Ctx *c, Slot &s, chip_tick_result *out)
{
    if (!s.in_flight) return CHIP_ERR_BUSY;
    if (!s.immediate) {
        // A fused tick's last workgroup writes the record into the slot's pinned host memory and THEN stores the slot's completion word
        // with a system-scope release: polling that word (acquire) hands the record over ~4-5 us before the stream's event would -- the
        // end-of-kernel cache maintenance, the event's barrier packet and its signal are off the synchronous tick's critical path.  The
        // event is still recorded (it orders the reuse of the list buffer); it is only waited for if the word does not arrive.
        bool done = false;
        if (s.resident) {
            const int rc = resident_collect(c, s);
            if (rc != CHIP_OK) { s.in_flight = false; s.poll = false; return rc; }
            done = true;
        } else if (s.poll) {
            const volatile unsigned long long *w = s.seq_host;
            for (long spin = 0; spin < (1L << 24) && !done; spin++) {
                done = __atomic_load_n(w, __ATOMIC_ACQUIRE) == s.seq_want;
                if (!done && (spin & 0x3fff) == 0x3fff && hipEventQuery(s.wait_ev ? s.wait_ev : s.done) != hipErrorNotReady) break;   // finished or failed: let the event say which
            }
        }
        if (!done) CHIP_HIP(c, hipEventSynchronize(s.wait_ev ? s.wait_ev : s.done));
        s.poll = false;
    }
    *out = *s.host;
    s.in_flight = false;
    if (out->status == CHIP_TICK_FAILED) {   // a shard could not take part: the tick had no effect (:1098 was not reached)
        // Roll last_l back only while this tick is still the newest one enqueued.  With ticks pipelined, a later tick may have been
        // enqueued (and may succeed) meanwhile: its commit of last_l stands, exactly as if the reference's sequential loop had
        // skipped the failed pass and run the next one.
        if (s.last_l_ptr && *s.last_l_ptr == s.tick_l) *s.last_l_ptr = s.prev_last_l;
        else if (s.last_l_ptr) {
            // A newer tick was enqueued on top of this failed one.  Should IT fail too, it must not roll back to this tick's l (a pass
            // that never reached :1098): hand it this tick's predecessor instead, so that a run of failed ticks unwinds to the last_l
            // of the newest tick that did not fail.  The successor is the in-flight slot whose prev_last_l is this l.
            for (Slot &o : c->slots)
                if (&o != &s && o.in_flight && o.last_l_ptr == s.last_l_ptr && o.prev_last_l == s.tick_l) o.prev_last_l = s.prev_last_l;
        }
        fill_immediate(out, CHIP_TICK_FAILED);
        return CHIP_ERR_SHARD_FAILED;
    }
    return CHIP_OK;
}

int sync_topk_out(Ctx *c, int nq, int K, double *scores, int64_t *idx)
{
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));  // the last block stored the list into pinned host memory
    if (c->topk_host[0].idx == -2) return CHIP_ERR_SHARD_FAILED;   // kFailedShardIdx: a shard could not take part (every rank sees it)
    for (int i = 0; i < nq * K; i++) {
        if (scores) scores[i] = c->topk_host[i].score;
        if (idx) idx[i] = c->topk_host[i].idx;
    }
    return CHIP_OK;
}

// Announce an append to queriers of a sharded ctx and order its ring writes behind the scans already enqueued.
static int ring_begin_append(Ctx *c, int64_t new_total)
{
    if (!c->ring_dev) return CHIP_OK;
    hipEvent_t ev[2];
    {
        std::lock_guard<std::mutex> rl(c->ring_mu);
        c->rows_pending = new_total;        // queriers now validate ring residency against the length after this call
        ev[0] = c->last_scan_ev[0];
        ev[1] = c->last_scan_ev[1];
    }
    for (hipEvent_t e : ev)                 // scans already enqueued read their queries before the ring is overwritten
        if (e) CHIP_HIP(c, hipStreamWaitEvent(c->s_append, e, 0));
    return CHIP_OK;
}

// ---- append, in phases (see chip_internal.h).  The caller holds the append lock and has made the device current. ----
int append_reserve(Ctx *c, int64_t first, int64_t n) { return ensure_capacity(c, local_count(c, first + n)); }

// Upload rows [from, from + count) of the call (step > 1: every step-th row, i.e. one shard's rows) in staging-sized chunks and
// run K3 on them.  ring: mirror into the replicated ring as well.
static int append_pass(Ctx *c, const void *desc, int src_elem, int64_t first, int64_t from, int64_t count, int64_t step, bool ring)
{
    const size_t row_bytes = (size_t)c->D * src_elem;
    const int64_t chunk_rows = (int64_t)(c->stage_bytes / row_bytes);
    for (int64_t done = 0; done < count; done += chunk_rows) {
        const int64_t m = (count - done) < chunk_rows ? (count - done) : chunk_rows;
        const char *src = static_cast<const char *>(desc) + (size_t)(from + done * step) * row_bytes;
        if (step == 1)
            CHIP_HIP(c, hipMemcpyAsync(c->stage_dev, src, (size_t)m * row_bytes, hipMemcpyHostToDevice, c->s_append));
        else   // one shard's rows: a strided gather out of the caller's batch, one row per pitch line
            CHIP_HIP(c, hipMemcpy2DAsync(c->stage_dev, row_bytes, src, (size_t)step * row_bytes, row_bytes, (size_t)m, hipMemcpyHostToDevice, c->s_append));
        const int r = launch_store_rows(c, c->s_append, c->stage_dev, src_elem, m, first + from + done * step, c->flags_dev, ring, step);
        if (r != CHIP_OK) return r;
        // the staging buffer is reused by the next chunk: stream order serialises copy -> kernel -> copy
    }
    return CHIP_OK;
}

// DB rows only (rows past the published length are invisible to queries); *bad gets the validation bits of THIS ctx's share:
// bit0 = a value that is not float32-representable went into float rows, bit1 = NaN / Inf.
int append_store_db(Ctx *c, const void *desc, int src_elem, int64_t first, int64_t n, bool owner_only, uint32_t *bad)
{
    *c->flags_host = 0;
    CHIP_HIP(c, hipMemsetAsync(c->flags_dev, 0, sizeof(uint32_t), c->s_append));
    int rc;
    if (owner_only && c->nranks > 1) {
        const int64_t G = c->nranks;
        const int64_t i0 = ((c->rank - first) % G + G) % G;               // first row of the batch this shard owns
        const int64_t cnt = n > i0 ? (n - i0 + G - 1) / G : 0;
        rc = append_pass(c, desc, src_elem, first, i0, cnt, G, false);
    } else {
        rc = append_pass(c, desc, src_elem, first, 0, n, 1, false);
    }
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, hipMemcpyAsync(c->flags_host, c->flags_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, c->s_append));
    CHIP_HIP(c, hipStreamSynchronize(c->s_append));
    *bad = *c->flags_host;
    return CHIP_OK;
}

// Genuinely float64 descriptors (e.g. ReljaNetVLAD's numpy WPCA output, whole_image_desc_compute_server.py:148-149): an
// undecided, still EMPTY DB becomes a double-row DB -- as the reference's MatrixXd M (Cerebro.cpp:946).
bool append_can_switch_to_double(const Ctx *c, int64_t first) { return c->store_auto && first == 0 && (size_t)c->D * 8 * 2 <= 160 * 1024; }

int append_switch_to_double(Ctx *c, int64_t n)
{
    int rc = configure_storage(c, 8);
    if (rc != CHIP_OK) return rc;
    return ensure_capacity(c, local_count(c, c->cap_hint > n ? c->cap_hint : n));
}

void append_publish(Ctx *c, int64_t new_total, bool lossy, int64_t n)
{
    std::lock_guard<std::mutex> lk(c->mu);  // publish the new length only now (rows fully resident)
    c->rows_global = new_total;
    c->rows_local = local_count(c, c->rows_global);
    if (lossy) c->lossy_rows += n;   // upper bound: rows of this call
    c->store_auto = false;           // the storage type is final once the DB holds a row
}

int append_ring_publish(Ctx *c, const void *desc, int src_elem, int64_t first, int64_t n, bool lossy)
{
    if (c->ring_dev) {  // sharded: mirror the newest rows into the replicated ring (the DB store of owned rows is idempotent)
        int rc = ring_begin_append(c, first + n);
        if (rc != CHIP_OK) return rc;
        const int64_t m = n < CHIP_RING_ROWS ? n : CHIP_RING_ROWS;
        rc = append_pass(c, desc, src_elem, first, n - m, m, 1, true);
        if (rc != CHIP_OK) return rc;
        CHIP_HIP(c, hipStreamSynchronize(c->s_append));
    }
    append_publish(c, first + n, lossy, n);
    return CHIP_OK;
}

int ctx_append(Ctx *c, const void *desc, int src_elem, int64_t n, uint32_t flags, int64_t *first_index)
{
    if (!c || !desc || n < 0) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> alk(c->append_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    const int64_t first = published_rows(c);
    if (first_index) *first_index = first;
    if (n == 0) return CHIP_OK;
    int rc = append_reserve(c, first, n);
    if (rc != CHIP_OK) return rc;
    // A ctx that is fed the whole stream by its own caller (single GPU, or one process per GPU) validates the WHOLE batch, so that
    // every rank takes the same storage-type / rejection decision without talking to the others.
    uint32_t bad = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        rc = append_store_db(c, desc, src_elem, first, n, false, &bad);
        if (rc != CHIP_OK) return rc;
        if (bad & 2u) return CHIP_ERR_NONFINITE;
        if (!(bad & 1u)) break;
        // CHIP_APPEND_ALLOW_ROUNDING is the caller saying "these ARE float32 descriptors, up to the precision they were printed / sent
        // with" (a state.json checkpoint holds 15-digit decimal text, RawFileIO.cpp:330-459): they are rounded to the float rows they
        // came from, also in an empty undecided DB.  Without the flag the data decides: an empty undecided DB takes values that are
        // not float32-representable as they are (double rows); anywhere else they are refused.
        if (flags & CHIP_APPEND_ALLOW_ROUNDING) break;
        if (attempt == 0 && append_can_switch_to_double(c, first)) {
            rc = append_switch_to_double(c, n);
            if (rc != CHIP_OK) return rc;
            continue;
        }
        return CHIP_ERR_NOT_F32;
    }
    return append_ring_publish(c, desc, src_elem, first, n, (bad & 1u) != 0);
}

int synth_generate(Ctx *c, int64_t first, int64_t n, uint64_t seed, const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant, int unit)
{
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
    int rc = ring_begin_append(c, first + n);   // same ordering against in-flight scans as ctx_append
    if (rc == CHIP_OK) rc = launch_synth(c, c->s_append, first, n, seed, pd, ps, pk, n_plant, unit);
    hipError_t e = hipStreamSynchronize(c->s_append);
    if (pd) (void)hipFree(pd);
    if (ps) (void)hipFree(ps);
    if (pk) (void)hipFree(pk);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(c, e);
    return CHIP_OK;
}

static int check_plants(int64_t first, int64_t n, const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant)
{
    for (int64_t i = 0; i < n_plant; i++) {
        if (plant_dst[i] < first || plant_dst[i] >= first + n || plant_src[i] < 0) return CHIP_ERR_RANGE;
        if (i > 0 && plant_dst[i] <= plant_dst[i - 1]) return CHIP_ERR_INVALID_ARG;
        if (plant_kind[i] != 1 && plant_kind[i] != 2) return CHIP_ERR_INVALID_ARG;
    }
    return CHIP_OK;
}

int ctx_append_synthetic(Ctx *c, int64_t n, uint64_t seed, const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant, int unit)
{
    if (!c || n < 0 || n_plant < 0 || (n_plant > 0 && (!plant_dst || !plant_src || !plant_kind))) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> alk(c->append_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    const int64_t first = published_rows(c);
    int rc = check_plants(first, n, plant_dst, plant_src, plant_kind, n_plant);
    if (rc != CHIP_OK) return rc;
    if (n == 0) return CHIP_OK;
    rc = append_reserve(c, first, n);
    if (rc != CHIP_OK) return rc;
    rc = synth_generate(c, first, n, seed, plant_dst, plant_src, plant_kind, n_plant, unit);
    if (rc != CHIP_OK) return rc;
    append_publish(c, first + n, false, n);
    return CHIP_OK;
}

// one row (storage type) to host memory, asynchronously on the ctx stream
int ctx_read_row(Ctx *c, int64_t g, int64_t total, void *out)
{
    const void *src;
    if (owns_row(c, g)) src = row_ptr_host(c, local_of(c, g));
    else if (c->ring_dev && g >= total - CHIP_RING_ROWS) src = ring_ptr(c, g);
    else return CHIP_ERR_RANGE;
    CHIP_HIP(c, hipMemcpyAsync(out, src, (size_t)c->D * c->elem, hipMemcpyDeviceToHost, c->s_query));
    return CHIP_OK;
}

static int read_rows(chip_ctx *c, const int64_t *rows, int64_t n, void *out, int out_elem)
{
    if (!c || !rows || !out || n < 0) return CHIP_ERR_INVALID_ARG;
    if (c->group) return group_read_rows(c, rows, n, out, out_elem);
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    if (out_elem < c->elem) return CHIP_ERR_NOT_F32;   // double rows do not fit a float buffer: chip_db_read_rows_f64
    const int64_t total = published_rows(c);
    const bool conv = out_elem != c->elem;             // float rows into a double buffer: widen on the host
    std::vector<float> tmp;
    if (conv) tmp.resize((size_t)n * c->D);
    for (int64_t i = 0; i < n; i++) {
        const int64_t g = rows[i];
        if (g < 0 || g >= total) return CHIP_ERR_RANGE;
        void *dst = conv ? (void *)(tmp.data() + (size_t)i * c->D) : (void *)(static_cast<char *>(out) + (size_t)i * c->D * out_elem);
        const int rc = ctx_read_row(c, g, total, dst);
        if (rc != CHIP_OK) return rc;
    }
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    if (conv)
        for (size_t i = 0; i < tmp.size(); i++) static_cast<double *>(out)[i] = (double)tmp[i];
    return CHIP_OK;
}

int merge_enqueue_slot(Ctx *c, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists, int32_t topk, Slot &s)
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
    s.wait_ev = s.done;
    s.immediate = false;
    s.in_flight = true;
    return CHIP_OK;
}

// merge of gathered [n_lists][nq][K] lists into out[nq][K] on the ctx stream, no decision
int merge_enqueue_out(Ctx *c, const void *dev_gathered, int32_t n_lists, int nq, int32_t topk, chip_topk_entry *out)
{
    MergeArgs m;
    m.in = (const chip_topk_entry *)dev_gathered;
    m.n_lists = n_lists;
    m.K = topk;
    m.out = out;
    m.result = nullptr;
    m.l = 0;
    m.locality = 0;
    m.thresh = 0.0;
    return launch_merge(c, c->s_query, m, nq);
}

// All scores of one query over this ctx's share of the prefix [0,k): u_global[local * mul + add] = score (host buffer).
int ctx_scores_local(Ctx *c, int64_t k, const void *q, double *u_global, int64_t mul, int64_t add)
{
    const int64_t n_rows = local_count(c, k);
    if (n_rows == 0) return CHIP_OK;
    if (n_rows > c->scores_cap) {
        if (c->scores_dev) { (void)hipFree(c->scores_dev); c->scores_dev = nullptr; c->scores_cap = 0; }
        const int64_t cap = n_rows + n_rows / 4 + 1024;
        CHIP_HIP(c, hipMalloc(&c->scores_dev, (size_t)cap * sizeof(double)));
        c->scores_cap = cap;
    }
    ScanArgs a;
    a.seg_table = c->seg_table_dev;
    a.seg_shift = c->seg_shift;
    a.seg_mask = c->seg_rows - 1;
    a.n_rows = n_rows;
    a.D = c->D;
    a.K = 1;
    for (int i = 0; i < CHIP_MAX_NQ; i++) a.q[i] = i == 0 ? q : nullptr;
    a.idx_mul = mul;
    a.idx_add = add;
    a.partial = nullptr;
    int rc = launch_scores(c, c->s_scan, a, c->scores_dev);
    if (rc != CHIP_OK) return rc;
    if (mul == 1 && add == 0) {
        CHIP_HIP(c, hipMemcpyAsync(u_global, c->scores_dev, (size_t)n_rows * sizeof(double), hipMemcpyDeviceToHost, c->s_scan));
        CHIP_HIP(c, hipStreamSynchronize(c->s_scan));
    } else {
        std::vector<double> tmp((size_t)n_rows);
        CHIP_HIP(c, hipMemcpyAsync(tmp.data(), c->scores_dev, (size_t)n_rows * sizeof(double), hipMemcpyDeviceToHost, c->s_scan));
        CHIP_HIP(c, hipStreamSynchronize(c->s_scan));
        for (int64_t r = 0; r < n_rows; r++) u_global[r * mul + add] = tmp[(size_t)r];
    }
    return CHIP_OK;
}

}  // namespace chip

extern "C" {

// ------------------------------------------------------------------------------------------------ lifecycle
int chip_create(chip_ctx **out, int32_t D, int64_t capacity_hint, int32_t device, int32_t shard_rank, int32_t shard_count)
{
    return ctx_create(out, D, capacity_hint, device, shard_rank, shard_count, 0);
}

int chip_create_ex(chip_ctx **out, int32_t D, int64_t capacity_hint, int32_t device, int32_t shard_rank, int32_t shard_count, uint32_t flags)
{
    if (flags & ~kCreateStoreMask) return CHIP_ERR_INVALID_ARG;
    return ctx_create(out, D, capacity_hint, device, shard_rank, shard_count, flags);
}

void chip_destroy(chip_ctx *ctx) { ctx_destroy(ctx); }

int chip_set_stream(chip_ctx *c, void *hip_stream)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group || c->xchg) return CHIP_ERR_UNSUPPORTED;   // the library owns the exchange and its stream ordering
    std::lock_guard<std::mutex> lk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    resident_stop(c);                       // ticks of a ctx on a caller's stream are launched behind that stream's work
    if (c->own_query_stream) CHIP_HIP(c, hipStreamDestroy(c->s_query));
    c->s_query = (hipStream_t)hip_stream;   // may be 0: HIP's null stream is a valid external stream
    c->own_query_stream = false;
    return CHIP_OK;
}

int chip_reset_stream(chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group) return CHIP_ERR_UNSUPPORTED;
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
    if (c->group) return group_synchronize(c);
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(c->s_append));
    CHIP_HIP(c, hipStreamSynchronize(c->s_scan));
    if (c->s_scan2) CHIP_HIP(c, hipStreamSynchronize(c->s_scan2));
    for (hipStream_t x : c->s_scan_x)
        if (x) CHIP_HIP(c, hipStreamSynchronize(x));
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    CHIP_HIP(c, hipStreamSynchronize(c->s_pnp));
    {   // a tick that went to the resident scan instance is on no stream: it is complete when its slot's completion word says so
        std::lock_guard<std::mutex> lk(c->res_mu);
        if (c->res_busy && c->res_slot) {
            const volatile unsigned long long *w = c->res_slot->seq_host;
            for (long spin = 0; spin < (1L << 28) && __atomic_load_n(w, __ATOMIC_ACQUIRE) != c->res_slot->seq_want && c->res_alive && !resident_has_left(c); spin++) {}
        }
    }
    return CHIP_OK;
}

// Public form of the pause the library takes around its own frees / allocations (include/cerebro_hip.h): an integrator brackets
// device-wide operations of OTHER libraries in the same process with it.  No-ops (CHIP_OK) on ctxs without the resident mode.
int chip_resident_pause(chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group) return CHIP_OK;                 // group / sharded ctxs never run the resident instance
    (void)hipSetDevice(c->device);
    resident_pause(c);
    return CHIP_OK;
}

int chip_resident_resume(chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group) return CHIP_OK;
    resident_resume(c);
    return CHIP_OK;
}

// ------------------------------------------------------------------------------------------------ append
int chip_db_append_f64(chip_ctx *c, const double *desc, int64_t n, uint32_t flags, int64_t *first_index)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    return c->group ? group_append(c, desc, 8, n, flags, first_index) : ctx_append(c, desc, 8, n, flags, first_index);
}

int chip_db_append_f32(chip_ctx *c, const float *desc, int64_t n, int64_t *first_index)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    return c->group ? group_append(c, desc, 4, n, 0, first_index) : ctx_append(c, desc, 4, n, 0, first_index);
}

int64_t chip_db_size(const chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    return published_rows(c);
}

int chip_db_read_rows_f32(chip_ctx *c, const int64_t *rows, int64_t n, float *out) { return read_rows(c, rows, n, out, 4); }
int chip_db_read_rows_f64(chip_ctx *c, const int64_t *rows, int64_t n, double *out) { return read_rows(c, rows, n, out, 8); }

int chip_db_append_synthetic(chip_ctx *c, int64_t n, uint64_t seed,
                             const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    return c->group ? group_append_synthetic(c, n, seed, plant_dst, plant_src, plant_kind, n_plant, 0)
                    : ctx_append_synthetic(c, n, seed, plant_dst, plant_src, plant_kind, n_plant, 0);
}

int chip_db_append_synthetic_unit(chip_ctx *c, int64_t n, uint64_t seed,
                                  const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    return c->group ? group_append_synthetic(c, n, seed, plant_dst, plant_src, plant_kind, n_plant, 1)
                    : ctx_append_synthetic(c, n, seed, plant_dst, plant_src, plant_kind, n_plant, 1);
}

// ------------------------------------------------------------------------------------------------ queries
static int check_query_args(chip_ctx *c, int64_t k, int32_t nq, int32_t topk, int64_t *n_global)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (nq < 1 || nq > CHIP_MAX_NQ || topk < 1 || topk > CHIP_MAX_TOPK) return CHIP_ERR_UNSUPPORTED;
    *n_global = published_rows(c);
    if (k < 0) return CHIP_ERR_RANGE;
    // with an exchange attached "k beyond what THIS rank has published" is a per-rank condition: the rank takes part in the
    // collective call with the failure mark (chip_multi.hip) instead of leaving it
    if (k > *n_global && !(c->xchg && !c->group)) return CHIP_ERR_RANGE;
    return CHIP_OK;
}

static int query_common(chip_ctx *c, int64_t k, const int64_t *query_rows, const void *vectors, int vec_elem, int32_t nq, int32_t topk,
                        double *scores, int64_t *idx)
{
    int64_t n = 0;
    int rc = check_query_args(c, k, nq, topk, &n);
    if (rc != CHIP_OK) return rc;
    if (!query_rows && !vectors) return CHIP_ERR_INVALID_ARG;
    if (c->group) return group_query(c, k, query_rows, vectors, vec_elem, nq, topk, scores, idx);
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    const void *q[CHIP_MAX_NQ];
    RingGuard rg(c);
    if (c->xchg) {   // collective: every rank makes the same call; per-rank conditions become the failure mark
        bool fail_local = k > n;
        rc = query_rows ? xchg_fetch_rows(c, query_rows, nq, n, q, &fail_local) : upload_query_vectors(c, vectors, vec_elem, nq, q);
        if (rc != CHIP_OK) return rc;
        return xchg_query(c, k, q, nq, topk, scores, idx, fail_local);
    }
    rc = query_rows ? query_row_ptrs(c, query_rows, nq, n, q) : upload_query_vectors(c, vectors, vec_elem, nq, q);
    if (rc != CHIP_OK) return rc;
    rc = enqueue_scan_merge(c, k, q, nq, topk, 0, nullptr, c->topk_dev, nullptr, false, nullptr, nullptr);
    if (rc != CHIP_OK) return rc;
    return sync_topk_out(c, nq, topk, scores, idx);
}

int chip_query_rows(chip_ctx *c, int64_t k, const int64_t *query_rows, int32_t nq, int32_t topk, double *scores, int64_t *idx)
{
    if (!query_rows) return CHIP_ERR_INVALID_ARG;
    return query_common(c, k, query_rows, nullptr, 0, nq, topk, scores, idx);
}

int chip_query_vectors_f32(chip_ctx *c, int64_t k, const float *queries, int32_t nq, int32_t topk, double *scores, int64_t *idx)
{
    if (!queries) return CHIP_ERR_INVALID_ARG;
    return query_common(c, k, nullptr, queries, 4, nq, topk, scores, idx);
}

int chip_query_vectors_f64(chip_ctx *c, int64_t k, const double *queries, int32_t nq, int32_t topk, double *scores, int64_t *idx)
{
    if (!queries) return CHIP_ERR_INVALID_ARG;
    return query_common(c, k, nullptr, queries, 8, nq, topk, scores, idx);
}

int chip_query_scores(chip_ctx *c, int64_t k, int64_t query_row, double *u)
{
    if (!c || !u) return CHIP_ERR_INVALID_ARG;
    const int64_t n = published_rows(c);
    if (k < 0 || query_row < 0) return CHIP_ERR_RANGE;
    const bool collective = c->xchg && !c->group;
    const bool out_of_range = k > n || query_row >= n;          // per-rank conditions when the ranks append independently
    if (out_of_range && !collective) return CHIP_ERR_RANGE;
    if (c->group) return group_scores(c, k, query_row, u);
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    const void *q[1];
    RingGuard rg(c);
    int rc;
    if (collective) {   // the query row comes from its owner (any appended row); the broadcast is made by EVERY rank, also by one
                        // that then finds its own arguments out of range -- leaving before it would put the communicator out of step
        bool fail_local = false;
        rc = xchg_fetch_rows(c, &query_row, 1, n, q, &fail_local);
        if (rc == CHIP_OK && (fail_local || out_of_range)) rc = CHIP_ERR_RANGE;
    } else {
        rc = query_row_ptrs(c, &query_row, 1, n, q);
    }
    if (rc != CHIP_OK) return rc;
    // a sharded ctx fills only the entries of the rows it owns (u[i], i % shard_count == shard_rank)
    return ctx_scores_local(c, k, q[0], u, c->nranks, c->nranks == 1 ? 0 : c->rank);
}

// ------------------------------------------------------------------------------------------------ tick
int chip_loop_tick(chip_ctx *c, int64_t l, const chip_dot_params *p, chip_tick_result *out)
{
    if (!c || !p || !out) return CHIP_ERR_INVALID_ARG;
    if (c->group) {
        const int rc = group_tick_enqueue(c, l, p, CHIP_MAX_INFLIGHT - 1);
        return rc != CHIP_OK ? rc : group_tick_collect(c, CHIP_MAX_INFLIGHT - 1, out);
    }
    if (c->nranks != 1 && !c->xchg) return CHIP_ERR_UNSUPPORTED;  // host-driven exchange: chip_scan_local + chip_merge_decide
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    Slot &s = c->slots[CHIP_MAX_INFLIGHT - 1];
    c->tick_sync_now = true;            // (query_mu held) the scan form of this tick is chosen for its latency: kernels.hip scan_rows_form
    int rc = tick_enqueue_slot(c, l, p, s);
    c->tick_sync_now = false;
    if (rc != CHIP_OK) return rc;
    return tick_collect_slot(c, s, out);
}

int chip_loop_tick_enqueue(chip_ctx *c, int64_t l, const chip_dot_params *p, int32_t slot)
{
    if (!c || !p || slot < 0 || slot >= CHIP_MAX_INFLIGHT - 1) return CHIP_ERR_INVALID_ARG;
    if (c->group) return group_tick_enqueue(c, l, p, slot);
    if (c->nranks != 1 && !c->xchg) return CHIP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    return tick_enqueue_slot(c, l, p, c->slots[slot]);
}

int chip_loop_tick_collect(chip_ctx *c, int32_t slot, chip_tick_result *out)
{
    if (!c || !out || slot < 0 || slot >= CHIP_MAX_INFLIGHT - 1) return CHIP_ERR_INVALID_ARG;
    if (c->group) return group_tick_collect(c, slot, out);
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
    if (c->group) return CHIP_ERR_UNSUPPORTED;
    if (topk < 1 || topk > CHIP_MAX_TOPK) return CHIP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    int64_t k = 0;
    int rc = tick_prepare(published_rows(c), c->last_l, l, p, status, &k);
    if (rc == CHIP_OK && *status == CHIP_TICK_TOO_SHORT) c->last_l = l;
    if (rc != CHIP_OK || *status != CHIP_TICK_SCANNED) return rc;
    const int64_t rows[3] = {l - 1, l - 2, l - 3};
    const void *q[3];
    RingGuard rg(c);
    rc = query_row_ptrs(c, rows, 3, l, q);
    if (rc != CHIP_OK) return rc;
    // scan on s_scan, then (behind an event) the local merge on the ctx stream writes this rank's 3 x topk list to
    // dev_out: everything the caller enqueues next on the ctx stream (the all-gather) is ordered after it, while the
    // next tick's scan is free to start as soon as this scan ends.
    rc = enqueue_scan_merge(c, k, q, 3, topk, l, nullptr, (chip_topk_entry *)dev_out, nullptr, true, nullptr, nullptr);
    if (rc == CHIP_OK) c->last_l = l;  // :1098
    return rc;
}

int chip_merge_decide(chip_ctx *c, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists,
                      int32_t topk, chip_tick_result *out)
{
    if (!c || !p || !dev_gathered || !out || n_lists < 1) return CHIP_ERR_INVALID_ARG;
    if (c->group) return CHIP_ERR_UNSUPPORTED;
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
    if (c->group) return CHIP_ERR_UNSUPPORTED;
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
    const Ctx *r = c->group ? group_root(const_cast<chip_ctx *>(c)) : c;
    std::lock_guard<std::mutex> lk(r->mu);
    info->abi_version = CHIP_ABI_VERSION;
    info->D = r->D;
    info->device = r->device;
    info->shard_rank = r->rank;
    info->shard_count = r->nranks;
    info->n_cus = r->n_cus;
    info->rows_global = r->rows_global;
    info->rows_local = r->rows_local;
    info->capacity_local = (int64_t)r->segs.size() * r->seg_rows;
    info->lossy_rows = r->lossy_rows;
    std::strncpy(info->arch, r->arch, sizeof(info->arch) - 1);
    info->storage_bytes = r->elem;
    info->n_devices = c->group ? group_size(c) : 1;
    info->exchange = c->group ? c->group_transport : (c->xchg ? CHIP_EXCHANGE_RCCL : CHIP_EXCHANGE_NONE);
    info->comm_ranks = exchange_comm_ranks(c->group ? r : c);
    info->comm_init_abandoned = c->comm_init_abandoned;
    info->scan_forms = scan_forms_built();
    info->test_hooks = chip_build_test_hooks();
    return CHIP_OK;
}

// Tuning aid, not part of the ABI (no declaration in cerebro_hip.h): how many ticks the resident scan instance (CHIP_TICK_RESIDENT=1) has
// served and how many instances were launched for them -- tests use it to prove which path a tick took.
int chip_debug_resident_stats(chip_ctx *c, int64_t *ticks, int64_t *launches)
{
    if (!c || !ticks || !launches || c->group) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(c->res_mu);
    *ticks = c->res_ticks;
    *launches = c->res_launches;
    return CHIP_OK;
}

// ... and where its command lines live: -1 no instance has been set up yet, 0 pinned host memory + relay, 1 workgroup 0's line in
// device memory behind the PCIe BAR + relay, 2 every workgroup's line written by the host through the BAR
int chip_debug_resident_mode(chip_ctx *c)
{
    if (!c || c->group) return -1;
    std::lock_guard<std::mutex> lk(c->res_mu);
    if (!c->s_resident) return -1;
    return c->res_direct ? 2 : (c->res_cmd_in_vram ? 1 : 0);
}

// Tuning aid, not part of the ABI (no declaration in cerebro_hip.h): with CHIP_SCAN_STAMPS=1 the row-batched scan kernel leaves four
// s_memrealtime (100 MHz) stamps per wave -- entry, queries staged, rows done, block merge done -- of the most recent launch.
int chip_debug_scan_stamps(chip_ctx *c, unsigned long long *out, int64_t n_waves)
{
    if (!c || !out || c->group || !c->stamps_dev || n_waves > (int64_t)c->max_grid * 16 + 16) return CHIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    resident_stop(c);
    CHIP_HIP(c, hipDeviceSynchronize());
    CHIP_HIP(c, hipMemcpy(out, c->stamps_dev, (size_t)n_waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CHIP_OK;
}

int chip_profile_enable(chip_ctx *c, int32_t on)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group) return group_profile_enable(c, on);
    std::lock_guard<std::mutex> qlk(c->query_mu);
    if (on) resident_stop(c);       // profiled launches want the chip as a launch finds it
    c->prof_on = on != 0;
    return CHIP_OK;
}

int chip_profile_reset(chip_ctx *c)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group) c = static_cast<chip_ctx *>(group_root(c));
    std::lock_guard<std::mutex> qlk(c->query_mu);
    c->prof_used = 0;
    return CHIP_OK;
}

int chip_profile_scan(chip_ctx *c, double *total_ms, int64_t *n_launches, double *bytes_per_launch_last, double *span_ms)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group) {   // the root device's launches (every device scans its own 1/G of the prefix)
        const int rc = group_synchronize(c);
        if (rc != CHIP_OK) return rc;
        c = static_cast<chip_ctx *>(group_root(c));
    }
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
