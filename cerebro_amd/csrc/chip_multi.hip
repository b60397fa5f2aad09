// chip_multi.hip -- the multi-GPU side of libcerebro_hip.so, inside the C ABI (include/cerebro_hip.h, "multi-GPU inside
// the library").  The descriptor DB is sharded by rows (row i on shard i % G); a tick is G local scans, a per-shard local
// top-k, ONE exchange of 3 x topk (score, index) entries per shard, and the merge + accept decision of
// src/Cerebro.cpp:1035-1056 over the G lists.  Two process layouts share the code:
//   * Exchange on a sharded ctx (one process per GPU): an RCCL communicator attached with chip_comm_init_rank; the
//     ncclAllGather is enqueued on the ctx stream between the local and the global merge, so that it overlaps the NEXT
//     tick's scan (which runs on the ctx's scan streams) -- no host synchronisation inside a tick.
//   * Group (one process, G GPUs: the shape of the reference, whose producer is one thread of one process,
//     src/cerebro_node.cpp:499): G sub-contexts, one host worker thread per device, exchange over an ncclCommInitAll
//     communicator or -- when devices repeat (a 1-GPU box running the G-way code path) or on request -- by device copies.
// The payload is 384 B per shard per tick (topk 8): latency-bound, xGMI bandwidth is irrelevant.
#include "chip_internal.h"
#include "topk_merge.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <chrono>
#include <functional>
#include <memory>
#include <new>
#include <system_error>
#include <thread>

namespace chip {

static_assert(sizeof(ncclUniqueId) == CHIP_COMM_ID_BYTES, "CHIP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

// RCCL is loaded at run time (dlopen), not linked: a machine without librccl can still load libcerebro_hip.so and use everything
// that needs no collective -- single-GPU contexts, and groups over the device-copy exchange (chip_create_multi falls back to it
// and says so in chip_get_info().exchange / chip_last_comm_error()).  <rccl/rccl.h> is used for its types only.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static const Rccl &rccl()
{
    static const Rccl r = [] {
        Rccl x;
        const char *names[] = {std::getenv("CHIP_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
        auto sym = [&](const char *n) { return dlsym(x.handle, n); };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(sym("ncclCommCount"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(sym("ncclBroadcast"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitAll && x.CommInitRank && x.CommDestroy && x.CommCount && x.AllGather && x.Broadcast && x.GetErrorString;
        return x;
    }();
    return r;
}

#define CHIP_NCCL(ctx, expr)                        \
    do {                                            \
        ncclResult_t _r = (expr);                   \
        if (_r != ncclSuccess) {                    \
            (ctx)->last_comm = (int)_r;             \
            return CHIP_ERR_COMM;                   \
        }                                           \
    } while (0)

// ---- RCCL bootstraps under a deadline ----------------------------------------------------------------------------------------
// ncclCommInitAll / ncclCommInitRank are blocking rendezvous (sockets, shared memory, peer-access probing): on a node whose RCCL
// bootstrap cannot complete they do not fail, they HANG -- and with them the caller's process.  Both run on a helper thread here;
// if the helper has not returned within the deadline (CHIP_COMM_INIT_TIMEOUT_MS, default 120 s) the call is ABANDONED: the helper
// is detached and keeps whatever it is blocked on, the library carries on without that communicator (a group falls back to the
// device-copy exchange, chip_comm_init_rank returns CHIP_ERR_COMM), chip_last_comm_error() says so and
// chip_get_info().comm_init_abandoned is set, so that the caller knows an orderly process teardown may block (leave through
// _exit).  Everything the helper touches lives in a shared_ptr'd job: it stays valid however late the helper wakes up.
constexpr int kCommInitTimedOut = 1000;     // last_comm value of an abandoned bootstrap (not an ncclResult_t)

struct CommInitJob {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclResult_t result = ncclSuccess;
    std::vector<ncclComm_t> comms;
    std::vector<int> devices;
    ncclUniqueId id;
    int n_ranks = 0, rank = 0, device = 0;
    bool all = false;                        // ncclCommInitAll over `devices` / ncclCommInitRank(rank of n_ranks) on `device`
};

// Fault injection for the two bootstraps: CHIP_TEST_COMM_INIT=hang | fail.  Compiled only into the TEST build of the library
// (-DCHIP_TEST_HOOKS: `make testlibs` -> cerebro_amd/lib/hooks/libcerebro_hip.so, which only tests/ load; chip_get_info().test_hooks
// says which build a process runs).  `make lib` -- the product -- contains none of it: a stray environment variable cannot degrade a
// deployed node.  Where compiled in, a hook still announces itself on stderr every time it fires.
static int comm_init_test_hook()
{
#ifndef CHIP_TEST_HOOKS
    return 0;
#else
    const char *t = std::getenv("CHIP_TEST_COMM_INIT");
    if (!t || !*t) return 0;
    const int mode = std::strcmp(t, "hang") == 0 ? 1 : std::strcmp(t, "fail") == 0 ? 2 : 0;
    if (mode) std::fprintf(stderr, "[cerebro_hip] TEST HOOK ACTIVE: CHIP_TEST_COMM_INIT=%s -- the RCCL bootstrap is made to %s\n", t, mode == 1 ? "hang" : "fail");
    return mode;
#endif
}

static int comm_init_timeout_ms()
{
    const int v = env_int("CHIP_COMM_INIT_TIMEOUT_MS", 120000);
    return v > 0 ? v : 120000;
}

// returns true when the helper finished in time (job->result / job->comms are final), false when it was abandoned
static bool comm_init_run(const std::shared_ptr<CommInitJob> &job)
{
    try {
    std::thread([job] {
        ncclResult_t r;
        const int hook = comm_init_test_hook();
        if (hook == 1) {
            for (;;) std::this_thread::sleep_for(std::chrono::hours(1));   // a bootstrap that never returns
        } else if (hook == 2) {
            r = ncclSystemError;
        } else if (job->all) {
            r = rccl().CommInitAll(job->comms.data(), (int)job->devices.size(), job->devices.data());
        } else {
            r = hipSetDevice(job->device) == hipSuccess ? rccl().CommInitRank(&job->comms[0], job->n_ranks, job->id, job->rank) : ncclUnhandledCudaError;
        }
        std::lock_guard<std::mutex> lk(job->m);
        job->result = r;
        job->done = true;
        // nobody is waiting any more: the communicators are left alone (destroying a clique whose creator has moved on to the copy
        // exchange could block again); the job, and with it this record of them, is released with the last reference
        job->cv.notify_all();
    }).detach();
    } catch (const std::system_error &) {   // no thread to be had: an extern "C" entry point must not throw -- report it as a failed bootstrap
        job->result = ncclSystemError;
        job->done = true;
        return true;
    }
    std::unique_lock<std::mutex> lk(job->m);
    if (job->cv.wait_for(lk, std::chrono::milliseconds(comm_init_timeout_ms()), [&] { return job->done; })) return true;
    job->abandoned = true;
    std::fprintf(stderr, "[cerebro_hip] %s did not return within %d ms (CHIP_COMM_INIT_TIMEOUT_MS): abandoned\n",
                 job->all ? "ncclCommInitAll" : "ncclCommInitRank", comm_init_timeout_ms());
    return false;
}

// Per-shard list buffers.  A tick's buffers are picked by a running counter modulo kXRing (>= the number of ticks a caller
// can keep in flight), so the buffers of tick i are not rewritten before its merge has run -- no backward dependencies
// between the streams of different devices.
struct Exchange {
    ncclComm_t comm = nullptr;
    int world = 1;
    chip_topk_entry *local_ring = nullptr;      // [kXRing][kListEntries]          this shard's lists of a tick
    chip_topk_entry *gathered_ring = nullptr;   // [kXRing][world][kListEntries]   all shards' lists ([world][nq][K] packed)
    hipEvent_t ev_local[kXRing] = {};           // copy exchange: this shard's list of tick b is written
    chip_topk_entry *failed_list = nullptr;     // [kListEntries] of (-inf, kFailedShardIdx): what a shard that cannot take part sends
    int32_t *agree_dev = nullptr;               // [1 + world]: this rank's "I can take part" word, then every rank's (xchg_agree)
    int32_t *agree_host = nullptr;              // pinned, [1 + world]
    uint64_t n = 0;                             // ticks / queries exchanged so far (one-process-per-GPU layout)
    uint64_t n_calls = 0;                       // collective calls seen by this shard (test hook below)
    int test_fail_every = 0;                    // CHIP_TEST_FAIL_SHARD="rank:every": this shard fails its validation on every
                                                // `every`-th collective call -- exercises the failure-mark path on a healthy box
    chip_topk_entry *local(int b) const { return local_ring + (size_t)b * kListEntries; }
    chip_topk_entry *gathered(int b) const { return gathered_ring + (size_t)b * world * kListEntries; }
};

static int exchange_create(Ctx *c, int world, bool need_gathered)
{
    Exchange *x = new (std::nothrow) Exchange();
    if (!x) return CHIP_ERR_OOM;
    c->xchg = x;
    x->world = world;
    CHIP_HIP(c, hipMalloc(&x->local_ring, sizeof(chip_topk_entry) * kXRing * kListEntries));
    if (need_gathered) CHIP_HIP(c, hipMalloc(&x->gathered_ring, sizeof(chip_topk_entry) * kXRing * kListEntries * (size_t)world));
    for (int i = 0; i < kXRing; i++) CHIP_HIP(c, hipEventCreateWithFlags(&x->ev_local[i], hipEventDisableTiming));
    {
        std::vector<chip_topk_entry> mark((size_t)kListEntries);
        for (chip_topk_entry &e : mark) { e.score = -INFINITY; e.idx = kFailedShardIdx; }
        CHIP_HIP(c, hipMalloc(&x->failed_list, sizeof(chip_topk_entry) * kListEntries));
        CHIP_HIP(c, hipMemcpy(x->failed_list, mark.data(), sizeof(chip_topk_entry) * kListEntries, hipMemcpyHostToDevice));
    }
    CHIP_HIP(c, hipMalloc(&x->agree_dev, sizeof(int32_t) * (size_t)(1 + world)));
    CHIP_HIP(c, hipHostMalloc(&x->agree_host, sizeof(int32_t) * (size_t)(1 + world), hipHostMallocDefault));
#ifdef CHIP_TEST_HOOKS
    if (const char *t = std::getenv("CHIP_TEST_FAIL_SHARD")) {
        int r = -1, every = 0;
        if (std::sscanf(t, "%d:%d", &r, &every) == 2 && r == c->rank && every > 0) {
            x->test_fail_every = every;
            std::fprintf(stderr, "[cerebro_hip] TEST HOOK ACTIVE: CHIP_TEST_FAIL_SHARD=%s -- shard %d fails every %d-th collective call\n", t, r, every);
        }
    }
#endif
    return CHIP_OK;
}

// test hook: does this shard pretend its validation failed on this collective call?
static bool test_fail_now(Exchange *x)
{
    x->n_calls++;
    return x->test_fail_every > 0 && x->n_calls % (uint64_t)x->test_fail_every == 0;
}

// test hook: CHIP_TEST_BATCH_OOM="rank": this rank's many-query call pretends its per-call allocations failed (announced on stderr)
static int test_oom_rank()
{
#ifndef CHIP_TEST_HOOKS
    return -1;
#else
    static const int r = [] {
        const char *t = std::getenv("CHIP_TEST_BATCH_OOM");
        if (!t || !*t) return -1;
        std::fprintf(stderr, "[cerebro_hip] TEST HOOK ACTIVE: CHIP_TEST_BATCH_OOM=%s -- that rank's many-query allocations fail\n", t);
        return std::atoi(t);
    }();
    return r;
#endif
}

void exchange_destroy(Ctx *c)
{
    Exchange *x = c->xchg;
    if (!x) return;
    (void)hipSetDevice(c->device);
    if (x->comm && rccl().ok) (void)rccl().CommDestroy(x->comm);
    if (x->local_ring) (void)hipFree(x->local_ring);
    if (x->gathered_ring) (void)hipFree(x->gathered_ring);
    if (x->failed_list) (void)hipFree(x->failed_list);
    if (x->agree_dev) (void)hipFree(x->agree_dev);
    if (x->agree_host) (void)hipHostFree(x->agree_host);
    for (hipEvent_t e : x->ev_local)
        if (e) (void)hipEventDestroy(e);
    delete x;
    c->xchg = nullptr;
}

int exchange_comm_ranks(const Ctx *c)
{
    if (!c->xchg || !c->xchg->comm) return 0;
    int n = 0;
    return rccl().ok && rccl().CommCount(c->xchg->comm, &n) == ncclSuccess ? n : 0;
}

// ------------------------------------------------------------------------------------------------ one process per GPU
// A rank whose own validation fails (its query rows have left its ring under a concurrent bulk append, its appender has not
// published row l yet, ...) must NOT leave a collective call: the other ranks have enqueued their ncclAllGather and the next
// collective of this rank would pair with it.  It takes part with the marked neutral list instead (Exchange::failed_list); the
// merge kernel of every rank sees the mark (topk_merge.h) and the call fails with CHIP_ERR_SHARD_FAILED on all of them alike --
// communicators never go out of step, the caller retries.  Only HIP / RCCL errors are hard.
// caller: tick_enqueue_slot (query_mu held, device current, tick_prepare said SCANNED)
int xchg_tick_enqueue(Ctx *c, int64_t l, int64_t k, const chip_dot_params *p, Slot &s, bool fail_local)
{
    Exchange *x = c->xchg;
    const int K = CHIP_DEFAULT_TOPK;
    const int64_t rows[3] = {l - 1, l - 2, l - 3};  // v, vm, vmm (Cerebro.cpp:987-989)
    const void *q[3];
    RingGuard rg(c);
    if (test_fail_now(x)) fail_local = true;
    if (!fail_local) {
        const int vrc = query_row_ptrs(c, rows, 3, l, q);
        if (vrc == CHIP_ERR_RANGE) fail_local = true;
        else if (vrc != CHIP_OK) return vrc;
    }
    const int b = (int)(x->n++ % kXRing);
    const chip_topk_entry *mine = x->failed_list;
    int rc_local = CHIP_OK;
    if (!fail_local) {
        rc_local = enqueue_scan_merge(c, k, q, 3, K, l, nullptr, x->local(b), nullptr, true, nullptr);   // scan streams -> local merge on the ctx stream
        if (rc_local == CHIP_OK) mine = x->local(b);
        // a HIP error / OOM of THIS rank's enqueue: the peers have posted their all-gather -- take part with the mark (they
        // get CHIP_ERR_SHARD_FAILED for this tick, the communicator stays in step), report the local error afterwards
    }
    CHIP_NCCL(c, rccl().AllGather(mine, x->gathered(b), sizeof(chip_topk_entry) * 3 * K, ncclChar, x->comm, c->s_query));
    if (rc_local != CHIP_OK) return rc_local;
    return merge_enqueue_slot(c, l, p, x->gathered(b), x->world, K, s);               // merge + decision (:1035-1056), every rank
}

// Query rows of a sharded ctx with an exchange: every row has exactly one owner that keeps it for good (row % world), so it is
// BROADCAST from there into this rank's query buffer (16 KiB at D = 4096) -- any appended row can be a query row, not only the
// newest CHIP_RING_ROWS that the tick reads from the replicated ring.  Which path a row takes must not depend on per-rank state
// (the ranks' appenders run independently), so chip_query_rows / chip_query_scores always fetch.  *fail_local: this rank owns a
// row it has not published yet.
int xchg_fetch_rows(Ctx *c, const int64_t *rows, int nq, int64_t n_local_published, const void **q, bool *fail_local)
{
    Exchange *x = c->xchg;
    const size_t rb = (size_t)c->D * c->elem;
    for (int i = 0; i < nq; i++) {
        const int64_t g = rows[i];
        if (g < 0) return CHIP_ERR_RANGE;                 // the same on every rank
        const int owner = (int)(g % x->world);
        char *dst = static_cast<char *>(c->qvec_dev) + (size_t)i * rb;
        const void *src = dst;
        if (owner == c->rank) {
            if (g >= n_local_published) *fail_local = true;   // garbage goes out, and the failure mark with it
            else src = row_ptr_host(c, local_of(c, g));
        }
        CHIP_NCCL(c, rccl().Broadcast(src, dst, rb, ncclChar, owner, x->comm, c->s_scan));
        q[i] = dst;
    }
    return CHIP_OK;
}

int xchg_query(Ctx *c, int64_t k, const void *const *q, int nq, int K, double *scores, int64_t *idx, bool fail_local)
{
    Exchange *x = c->xchg;
    if (test_fail_now(x)) fail_local = true;
    const int b = (int)(x->n++ % kXRing);
    const chip_topk_entry *mine = x->failed_list;
    int rc_local = CHIP_OK;
    if (!fail_local) {
        rc_local = enqueue_scan_merge(c, k, q, nq, K, 0, nullptr, x->local(b), nullptr, false, nullptr);
        if (rc_local == CHIP_OK) mine = x->local(b);   // else: take part with the mark, report the local error afterwards (see xchg_tick_enqueue)
    }
    CHIP_NCCL(c, rccl().AllGather(mine, x->gathered(b), sizeof(chip_topk_entry) * nq * K, ncclChar, x->comm, c->s_query));
    if (rc_local != CHIP_OK) return rc_local;
    const int rc = merge_enqueue_out(c, x->gathered(b), x->world, nq, K, c->topk_dev);
    if (rc != CHIP_OK) return rc;
    return sync_topk_out(c, nq, K, scores, idx);
}

// ------------------------------------------------------------------------------------------------ group: workers
// One host thread per sub-context: it makes that device current once and enqueues the device's share of every group call,
// so the host cost of a tick (two launches, a few events, the collective) is paid G-wide in parallel instead of serially.
// Several host threads may be inside group calls at once (the reference's appender, querier and PnP threads share the ctx), so a
// worker has a FIFO of jobs, each pointing at the completion record of the call it belongs to.  A worker runs its jobs in the
// order they were posted; calls that must stay ordered across devices (ticks / queries: they carry collectives) are posted
// under the group's query lock, hence in the same order on every worker.
struct JobGroup {
    std::mutex m;
    std::condition_variable cv;
    int remaining = 0;
    int rc = CHIP_OK;
};
struct Job {
    std::function<int()> fn;
    JobGroup *grp = nullptr;
};
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> q;
    bool quit = false;
};

struct Group {
    std::vector<chip_ctx *> subs;
    std::vector<Worker *> workers;     // workers[g] serves subs[g]; the calling thread serves subs[0] itself
    std::vector<char> same_dev;        // subs[g] lives on the root's device
    int transport = CHIP_EXCHANGE_COPY;
    uint64_t n = 0;                    // ticks / queries exchanged so far
    std::atomic<int> broken{0};        // non-zero: a call failed on SOME devices after they had diverged (rings written, lengths
                                       // published, a collective half enqueued) -- every later call returns CHIP_ERR_GROUP_BROKEN
};

static int group_break(Group *G, int rc)
{
    int expect = 0;
    G->broken.compare_exchange_strong(expect, rc);
    return rc;
}

static void worker_main(Worker *w, int device)
{
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return !w->q.empty() || w->quit; });
        if (w->q.empty()) return;   // quit, nothing pending
        Job job = std::move(w->q.front());
        w->q.pop_front();
        lk.unlock();
        const int rc = job.fn();
        {
            std::lock_guard<std::mutex> g(job.grp->m);
            if (rc != CHIP_OK && job.grp->rc == CHIP_OK) job.grp->rc = rc;
            if (--job.grp->remaining == 0) job.grp->cv.notify_all();
        }
        lk.lock();
    }
}

// fn(g) for every sub-context, concurrently; first non-OK status wins
static int run_all(Group *G, const std::function<int(int)> &fn)
{
    const int n = (int)G->subs.size();
    JobGroup jg;
    jg.remaining = n - 1;
    for (int g = 1; g < n; g++) {
        Worker *w = G->workers[g];
        std::lock_guard<std::mutex> lk(w->m);
        Job j;
        j.fn = [&fn, g] { return fn(g); };
        j.grp = &jg;
        w->q.push_back(std::move(j));
        w->cv.notify_one();
    }
    int rc = fn(0);
    std::unique_lock<std::mutex> lk(jg.m);
    jg.cv.wait(lk, [&] { return jg.remaining == 0; });
    if (rc == CHIP_OK) rc = jg.rc;
    return rc;
}

Ctx *group_root(Ctx *gc) { return gc->group->subs[0]; }
int group_size(const Ctx *gc) { return (int)gc->group->subs.size(); }

void group_destroy(Ctx *gc)
{
    Group *G = gc->group;
    if (!G) return;
    for (Worker *w : G->workers) {
        if (!w) continue;
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->quit = true;
            w->cv.notify_all();
        }
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    // communicators first (ncclCommDestroy of an ncclCommInitAll clique wants all of them alive), then the contexts
    for (chip_ctx *s : G->subs) exchange_destroy(s);
    for (chip_ctx *s : G->subs) ctx_destroy(s);
    delete G;
    gc->group = nullptr;
}

static void mirror_state(Ctx *gc)
{
    Ctx *r = group_root(gc);
    std::lock_guard<std::mutex> lk(gc->mu);
    std::lock_guard<std::mutex> lk2(r->mu);
    gc->rows_global = r->rows_global;
    gc->elem = r->elem;
    gc->store_auto = r->store_auto;
}

// ------------------------------------------------------------------------------------------------ group: DB
// M.col(_s) = desc for a group (Cerebro.cpp:1005-1006; after loadStateFromDisk the first tick copies ALL columns, :133-161 --
// the bulk path).  Every device receives only the rows it owns (a strided gather out of the caller's batch: device g gets rows
// i % G == g) plus the newest CHIP_RING_ROWS rows of the batch for its replicated query ring, so a cold start of 1M x 4096 moves
// ~1x the batch over PCIe instead of Gx.  The call is phased so that the devices cannot diverge:
//   1. reserve capacity on every device                                   -- a failure changes nothing;
//   2. store + validate the owned rows on every device (unpublished)      -- a failure changes nothing visible;
//   3. ONE decision from the OR of the per-device validation bits (non-finite / not float32-representable / switch the empty
//      undecided DB to double rows everywhere and redo 1-2);
//   4. ring update + publish the new length on every device               -- a failure here leaves devices with different
//      lengths / rings: the group is marked broken and refuses further work.
int group_append(Ctx *gc, const void *desc, int src_elem, int64_t n, uint32_t flags, int64_t *first_index)
{
    if (!desc || n < 0) return CHIP_ERR_INVALID_ARG;
    Group *G = gc->group;
    std::lock_guard<std::mutex> alk(gc->append_mu);
    if (G->broken) return CHIP_ERR_GROUP_BROKEN;
    const int64_t first = published_rows(G->subs[0]);
    if (first_index) *first_index = first;
    if (n == 0) return CHIP_OK;
    const size_t ng = G->subs.size();
    auto on_dev = [&](int g, const std::function<int(Ctx *)> &fn) -> int {
        Ctx *c = G->subs[(size_t)g];
        std::lock_guard<std::mutex> lk(c->append_mu);
        CHIP_HIP(c, hipSetDevice(c->device));
        return fn(c);
    };
    std::vector<uint32_t> bad(ng, 0u);
    uint32_t any_bad = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        int rc = run_all(G, [&](int g) { return on_dev(g, [&](Ctx *c) { return append_reserve(c, first, n); }); });
        if (rc != CHIP_OK) return rc;
        rc = run_all(G, [&](int g) { return on_dev(g, [&](Ctx *c) { return append_store_db(c, desc, src_elem, first, n, true, &bad[(size_t)g]); }); });
        if (rc != CHIP_OK) return rc;
        any_bad = 0;
        for (uint32_t b : bad) any_bad |= b;
        if (any_bad & 2u) return CHIP_ERR_NONFINITE;
        if (!(any_bad & 1u)) break;
        if (flags & CHIP_APPEND_ALLOW_ROUNDING) break;   // the caller vouches for float32 descriptors (see ctx_append)
        if (attempt == 0 && append_can_switch_to_double(G->subs[0], first)) {
            rc = run_all(G, [&](int g) { return on_dev(g, [&](Ctx *c) { return append_switch_to_double(c, n); }); });
            if (rc != CHIP_OK) return group_break(G, rc);   // some devices may hold double rows now, others float rows
            continue;
        }
        return CHIP_ERR_NOT_F32;
    }
    const bool lossy = (any_bad & 1u) != 0;
    const int rc = run_all(G, [&](int g) { return on_dev(g, [&](Ctx *c) { return append_ring_publish(c, desc, src_elem, first, n, lossy); }); });
    if (rc != CHIP_OK) return group_break(G, rc);
    mirror_state(gc);
    return CHIP_OK;
}

int group_append_synthetic(Ctx *gc, int64_t n, uint64_t seed, const int64_t *pd, const int64_t *ps, const int32_t *pk, int64_t n_plant, int unit)
{
    Group *G = gc->group;
    std::lock_guard<std::mutex> alk(gc->append_mu);
    if (G->broken) return CHIP_ERR_GROUP_BROKEN;
    if (n < 0 || n_plant < 0 || (n_plant > 0 && (!pd || !ps || !pk))) return CHIP_ERR_INVALID_ARG;
    const int64_t first = published_rows(G->subs[0]);
    for (int64_t i = 0; i < n_plant; i++) {
        if (pd[i] < first || pd[i] >= first + n || ps[i] < 0) return CHIP_ERR_RANGE;
        if (i > 0 && pd[i] <= pd[i - 1]) return CHIP_ERR_INVALID_ARG;
        if (pk[i] != 1 && pk[i] != 2) return CHIP_ERR_INVALID_ARG;
    }
    if (n == 0) return CHIP_OK;
    auto on_dev = [&](int g, const std::function<int(Ctx *)> &fn) -> int {
        Ctx *c = G->subs[(size_t)g];
        std::lock_guard<std::mutex> lk(c->append_mu);
        CHIP_HIP(c, hipSetDevice(c->device));
        return fn(c);
    };
    int rc = run_all(G, [&](int g) { return on_dev(g, [&](Ctx *c) { return append_reserve(c, first, n); }); });
    if (rc != CHIP_OK) return rc;
    // the generator writes the rings as it goes: from here on a failure on one device leaves the others ahead of it
    rc = run_all(G, [&](int g) { return on_dev(g, [&](Ctx *c) { return synth_generate(c, first, n, seed, pd, ps, pk, n_plant, unit); }); });
    if (rc != CHIP_OK) return group_break(G, rc);
    for (chip_ctx *c : G->subs) append_publish(c, first + n, false, n);
    mirror_state(gc);
    return CHIP_OK;
}

int group_read_rows(Ctx *gc, const int64_t *rows, int64_t n, void *out, int out_elem)
{
    Group *G = gc->group;
    std::lock_guard<std::mutex> qlk(gc->query_mu);
    const int D = gc->D;
    int elem;
    int64_t total;
    {
        std::lock_guard<std::mutex> lk(gc->mu);   // mirror_state writes both under this lock
        total = gc->rows_global;
        elem = gc->elem;
    }
    if (out_elem < elem) return CHIP_ERR_NOT_F32;
    const bool conv = out_elem != elem;
    std::vector<float> tmp;
    if (conv) tmp.resize((size_t)n * D);
    const int ng = (int)G->subs.size();
    for (int64_t i = 0; i < n; i++) {
        const int64_t g = rows[i];
        if (g < 0 || g >= total) return CHIP_ERR_RANGE;
        Ctx *s = G->subs[(size_t)(g % ng)];
        CHIP_HIP(s, hipSetDevice(s->device));
        void *dst = conv ? (void *)(tmp.data() + (size_t)i * D) : (void *)(static_cast<char *>(out) + (size_t)i * D * out_elem);
        const int rc = ctx_read_row(s, g, total, dst);
        if (rc != CHIP_OK) return rc;
    }
    for (chip_ctx *s : G->subs) {
        CHIP_HIP(s, hipSetDevice(s->device));
        CHIP_HIP(s, hipStreamSynchronize(s->s_query));
    }
    if (conv)
        for (size_t i = 0; i < tmp.size(); i++) static_cast<double *>(out)[i] = (double)tmp[i];
    return CHIP_OK;
}

// ------------------------------------------------------------------------------------------------ group: scan + exchange
// What one group scan needs: the prefix, the queries (rows of the DB or external vectors), K, and where the merged result
// goes (a tick slot with the accept decision, or the root's pinned top-k buffer).
struct GroupScan {
    int64_t k = 0, l = 0;
    int nq = 0, K = 0;
    const int64_t *rows = nullptr;      // query rows, or ...
    const void *vectors = nullptr;      // ... external vectors (host)
    int vec_elem = 0;
    int64_t n_global = 0;
    const chip_dot_params *p = nullptr; // non-null: tick (decision into slot)
    Slot *slot = nullptr;
    bool tick = false;
    bool from_ring = false;             // tick whose three query rows are still in every device's replicated ring (the live path)
};

// Query rows of a group scan on device g.  Ticks read them from the device's replicated ring (the live path: no copy).  Any other
// query row -- chip_query_rows / chip_query_scores, e.g. the offline all-vs-all of the faiss-style policies on a long run
// (Cerebro.cpp:506-722) -- is fetched from the sub-context that OWNS it (row % G, which keeps it for good) into this device's
// query buffer, on the stream the scan will run on: any appended row can be a query row, not only the newest CHIP_RING_ROWS.
static int sub_query_rows(Group *G, int g, const GroupScan &j, const void **q)
{
    Ctx *c = G->subs[(size_t)g];
    if (j.from_ring) return query_row_ptrs(c, j.rows, j.nq, j.n_global, q);
    const size_t rb = (size_t)c->D * c->elem;
    const int ng = (int)G->subs.size();
    for (int i = 0; i < j.nq; i++) {
        const int64_t r = j.rows[i];
        if (r < 0 || r >= j.n_global) return CHIP_ERR_RANGE;
        Ctx *o = G->subs[(size_t)(r % ng)];
        const void *src = row_ptr_host(o, local_of(o, r));      // published rows never move
        char *dst = static_cast<char *>(c->qvec_dev) + (size_t)i * rb;
        if (o->device == c->device) CHIP_HIP(c, hipMemcpyAsync(dst, src, rb, hipMemcpyDeviceToDevice, c->s_scan));
        else CHIP_HIP(c, hipMemcpyPeerAsync(dst, c->device, src, o->device, rb, c->s_scan));
        q[i] = dst;
    }
    return CHIP_OK;
}

// device g's share: local scan -> local top-k list -> (RCCL) all-gather [-> root: merge].  A device whose validation fails (RANGE:
// its query rows have left its ring under a concurrent bulk append) still takes part, with the marked neutral list, so that the
// collectives of all devices stay paired; the root's merge reports the mark and the call fails as a whole (CHIP_ERR_SHARD_FAILED).
// *hard gets set when a HIP / RCCL call failed: the devices may be out of step and the group is broken.
static int sub_scan(Group *G, int g, const GroupScan &j, int b, std::atomic<int> *hard)
{
    Ctx *c = G->subs[(size_t)g];
    Ctx *root = G->subs[0];
    Exchange *x = c->xchg;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    auto fail_hard = [&](int rc) { hard->store(1); return rc; };
    if (hipSetDevice(c->device) != hipSuccess) return fail_hard(CHIP_ERR_HIP);
    const void *q[CHIP_MAX_NQ];
    RingGuard rg(c);
    bool soft = test_fail_now(x);
    if (!soft) {
        const int vrc = j.rows ? sub_query_rows(G, g, j, q) : upload_query_vectors(c, j.vectors, j.vec_elem, j.nq, q);
        if (vrc == CHIP_ERR_RANGE || vrc == CHIP_ERR_NOT_F32 || vrc == CHIP_ERR_NONFINITE) {
            if (!j.tick) return vrc;      // argument errors of a query are the same on every device: nothing was enqueued anywhere
            soft = true;                  // a tick: per-device state (ring residency) -- take part with the mark
        } else if (vrc != CHIP_OK) return fail_hard(vrc);
    }
    const size_t list = (size_t)j.nq * j.K;
    const bool direct = G->transport == CHIP_EXCHANGE_COPY && G->same_dev[(size_t)g];   // straight into the root's gather buffer
    chip_topk_entry *dst = direct ? root->xchg->gathered(b) + (size_t)g * list : x->local(b);
    int rc;
    // (fetched query rows were copied on s_scan: such a scan must run there too, i.e. not as a stream-alternating "tick")
    if (!soft) rc = enqueue_scan_merge(c, j.k, q, j.nq, j.K, j.l, nullptr, dst, nullptr, j.tick && j.from_ring, nullptr);
    else rc = hipMemcpyAsync(dst, x->failed_list, sizeof(chip_topk_entry) * list, hipMemcpyDeviceToDevice, c->s_query) == hipSuccess ? CHIP_OK : CHIP_ERR_HIP;
    if (rc != CHIP_OK) return fail_hard(rc);
    if (G->transport == CHIP_EXCHANGE_RCCL) {
        // one thread per device, each on its own communicator of the clique: the classic multi-threaded NCCL layout (no group call)
        const ncclResult_t nr = rccl().AllGather(x->local(b), x->gathered(b), sizeof(chip_topk_entry) * list, ncclChar, x->comm, c->s_query);
        if (nr != ncclSuccess) { c->last_comm = (int)nr; return fail_hard(CHIP_ERR_COMM); }
        if (g == 0) {
            rc = j.p ? merge_enqueue_slot(c, j.l, j.p, x->gathered(b), x->world, j.K, *j.slot)
                     : merge_enqueue_out(c, x->gathered(b), x->world, j.nq, j.K, c->topk_dev);
            if (rc != CHIP_OK) return fail_hard(rc);
        }
    } else if (g != 0) {
        if (hipEventRecord(x->ev_local[b], c->s_query) != hipSuccess) return fail_hard(CHIP_ERR_HIP);
    }
    return CHIP_OK;
}

// copy exchange, root side (after every device has enqueued and recorded): wait for the lists, pull in those that live on
// other devices, merge
static int root_gather_merge(Group *G, const GroupScan &j, int b)
{
    Ctx *root = G->subs[0];
    Exchange *rx = root->xchg;
    CHIP_HIP(root, hipSetDevice(root->device));
    const size_t list = (size_t)j.nq * j.K;
    for (size_t g = 1; g < G->subs.size(); g++) {
        Ctx *c = G->subs[g];
        CHIP_HIP(root, hipStreamWaitEvent(root->s_query, c->xchg->ev_local[b], 0));
        if (!G->same_dev[g])
            CHIP_HIP(root, hipMemcpyPeerAsync(rx->gathered(b) + g * list, root->device, c->xchg->local(b), c->device,
                                              sizeof(chip_topk_entry) * list, root->s_query));
    }
    if (j.p) return merge_enqueue_slot(root, j.l, j.p, rx->gathered(b), (int)G->subs.size(), j.K, *j.slot);
    return merge_enqueue_out(root, rx->gathered(b), (int)G->subs.size(), j.nq, j.K, root->topk_dev);
}

static int group_scan(Group *G, const GroupScan &j)
{
    if (G->broken) return CHIP_ERR_GROUP_BROKEN;
    const int b = (int)(G->n++ % kXRing);
    std::atomic<int> hard{0};
    int rc = run_all(G, [&](int g) { return sub_scan(G, g, j, b, &hard); });
    if (rc != CHIP_OK) return hard.load() ? group_break(G, rc) : rc;
    if (G->transport == CHIP_EXCHANGE_COPY) {
        rc = root_gather_merge(G, j, b);
        if (rc != CHIP_OK) return group_break(G, rc);
    }
    return rc;
}

int group_tick_enqueue(Ctx *gc, int64_t l, const chip_dot_params *p, int32_t slot)
{
    Group *G = gc->group;
    Ctx *root = G->subs[0];
    std::lock_guard<std::mutex> qlk(gc->query_mu);
    if (G->broken) return CHIP_ERR_GROUP_BROKEN;
    Slot &s = root->slots[slot];
    if (s.in_flight) return CHIP_ERR_BUSY;
    int64_t n;
    {
        std::lock_guard<std::mutex> lk(gc->mu);
        n = gc->rows_global;
    }
    int32_t status = 0;
    int64_t k = 0;
    int rc = tick_prepare(n, gc->last_l, l, p, &status, &k);
    if (rc != CHIP_OK) return rc;
    s.prev_last_l = gc->last_l;
    s.tick_l = l;
    s.last_l_ptr = &gc->last_l;
    if (status != CHIP_TICK_SCANNED) {
        if (status == CHIP_TICK_TOO_SHORT) gc->last_l = l;
        fill_immediate(s.host, status);
        s.immediate = true;
        s.in_flight = true;
        return CHIP_OK;
    }
    const int64_t rows[3] = {l - 1, l - 2, l - 3};  // v, vm, vmm (Cerebro.cpp:987-989)
    GroupScan j;
    j.k = k; j.l = l; j.nq = 3; j.K = CHIP_DEFAULT_TOPK; j.rows = rows; j.n_global = l; j.p = p; j.slot = &s; j.tick = true;
    // Live ticks trail the append head by a few rows and read their three query rows from every device's replicated ring (no copy).
    // A tick further back than the ring -- replaying an old schedule over a cold-started DB (cerebro_replay --state --devices ...) --
    // fetches them from the sub-contexts that own them, exactly as chip_query_rows does; the decision is one, from the group's length.
    j.from_ring = G->subs.size() == 1 || n - l <= CHIP_RING_ROWS - 3;
    rc = group_scan(G, j);
    if (rc != CHIP_OK) return rc;
    gc->last_l = l;   // :1098
    return CHIP_OK;
}

int group_tick_collect(Ctx *gc, int32_t slot, chip_tick_result *out)
{
    Ctx *root = group_root(gc);
    std::lock_guard<std::mutex> qlk(gc->query_mu);
    CHIP_HIP(root, hipSetDevice(root->device));
    return tick_collect_slot(root, root->slots[slot], out);
}

int group_query(Ctx *gc, int64_t k, const int64_t *rows, const void *vectors, int vec_elem, int nq, int K, double *scores, int64_t *idx)
{
    Group *G = gc->group;
    Ctx *root = G->subs[0];
    std::lock_guard<std::mutex> qlk(gc->query_mu);
    if (G->broken) return CHIP_ERR_GROUP_BROKEN;
    GroupScan j;
    j.k = k; j.nq = nq; j.K = K; j.rows = rows; j.vectors = vectors; j.vec_elem = vec_elem;
    {
        std::lock_guard<std::mutex> lk(gc->mu);
        j.n_global = gc->rows_global;
    }
    int rc = group_scan(G, j);
    if (rc != CHIP_OK) return rc;
    CHIP_HIP(root, hipSetDevice(root->device));
    return sync_topk_out(root, nq, K, scores, idx);
}

int group_scores(Ctx *gc, int64_t k, int64_t query_row, double *u)
{
    Group *G = gc->group;
    std::lock_guard<std::mutex> qlk(gc->query_mu);
    int64_t n;
    {
        std::lock_guard<std::mutex> lk(gc->mu);
        n = gc->rows_global;
    }
    if (G->broken) return CHIP_ERR_GROUP_BROKEN;
    const int ng = (int)G->subs.size();
    GroupScan j;
    j.nq = 1; j.rows = &query_row; j.n_global = n; j.tick = false;
    return run_all(G, [&](int g) -> int {
        Ctx *c = G->subs[(size_t)g];
        std::lock_guard<std::mutex> lk(c->query_mu);
        CHIP_HIP(c, hipSetDevice(c->device));
        const void *q[1];
        const int rc = sub_query_rows(G, g, j, q);      // from the owner's shard into this device's query buffer (ctx_scores_local runs on s_scan too)
        if (rc != CHIP_OK) return rc;
        return ctx_scores_local(c, k, q[0], u, ng, ng == 1 ? 0 : g);   // disjoint entries of u per device
    });
}

// Many-query (MFMA) mode on a group (SURVEY 8f N4 where the DB outgrows one GPU -- the offline all-vs-all of the faiss-style policies
// on a long run, Cerebro.cpp:506-722): every device runs db_gemm_topk over ITS rows of the prefix (global indices), leaving one
// sorted [Qpad][topk] list; the root pulls the G lists side by side (device / peer copies behind events -- Q x topk x 16 B per
// device, 32 KiB at Q = 256, topk = 8: the transport does not matter) and merges them with the same exact selection that merges
// the workgroups of one device (topk_merge_batch): the result is that of one device holding the whole DB, bit for bit.
int group_query_batch(Ctx *gc, int64_t k, const float *queries, int32_t Q, int32_t topk, float *scores, int64_t *idx)
{
    Group *G = gc->group;
    Ctx *root = G->subs[0];
    std::lock_guard<std::mutex> qlk(gc->query_mu);
    if (G->broken) return CHIP_ERR_GROUP_BROKEN;
    int64_t n;
    int elem;
    {
        std::lock_guard<std::mutex> lk(gc->mu);
        n = gc->rows_global;
        elem = gc->elem;
    }
    if (elem != 4) return CHIP_ERR_UNSUPPORTED;
    if (k < 0 || k > n) return CHIP_ERR_RANGE;
    const int ng = (int)G->subs.size();
    std::vector<chip_topk_entry *> lists((size_t)ng, nullptr);
    std::vector<hipEvent_t> evs((size_t)ng, nullptr);
    std::vector<int32_t> qpads((size_t)ng, 0);
    int rc = run_all(G, [&](int g) -> int {
        Ctx *c = G->subs[(size_t)g];
        std::lock_guard<std::mutex> lk(c->query_mu);
        CHIP_HIP(c, hipSetDevice(c->device));
        int r = batch_local_enqueue(c, k, queries, Q, topk, &lists[(size_t)g], &qpads[(size_t)g]);
        if (r != CHIP_OK) return r;
        r = batch_exchange_buffers(c, g == 0 ? ng : 0, qpads[(size_t)g], topk, nullptr, nullptr, &evs[(size_t)g]);
        if (r != CHIP_OK) return r;
        CHIP_HIP(c, hipEventRecord(evs[(size_t)g], c->s_scan));
        return CHIP_OK;
    });
    if (rc != CHIP_OK) return rc;   // nothing collective was enqueued: the group stays usable
    std::lock_guard<std::mutex> rlk(root->query_mu);
    CHIP_HIP(root, hipSetDevice(root->device));
    const int32_t Qpad = qpads[0];
    chip_topk_entry *gathered = nullptr, *merged = nullptr;
    rc = batch_exchange_buffers(root, ng, Qpad, topk, &gathered, &merged, nullptr);
    if (rc != CHIP_OK) return rc;
    const size_t list_bytes = sizeof(chip_topk_entry) * (size_t)Qpad * topk;
    for (int g = 0; g < ng; g++) {
        Ctx *c = G->subs[(size_t)g];
        if (g > 0) CHIP_HIP(root, hipStreamWaitEvent(root->s_scan, evs[(size_t)g], 0));
        chip_topk_entry *dst = gathered + (size_t)g * Qpad * topk;
        if (c->device == root->device) CHIP_HIP(root, hipMemcpyAsync(dst, lists[(size_t)g], list_bytes, hipMemcpyDeviceToDevice, root->s_scan));
        else CHIP_HIP(root, hipMemcpyPeerAsync(dst, root->device, lists[(size_t)g], c->device, list_bytes, root->s_scan));
    }
    rc = batch_merge_lists(root, root->s_scan, gathered, ng, Qpad, Q, topk, merged);
    if (rc != CHIP_OK) return rc;
    rc = batch_deliver(root, merged, Q, topk, scores, idx);
    // the sub-contexts' list buffers are reused by their next call: it is posted after this one returns (group query lock), and the
    // root's copies above have completed by then (batch_deliver synchronised the root's scan stream)
    return rc;
}

// One small collective in which every rank says whether it can take part in the LARGE one that follows (4 bytes per rank, on the
// ctx stream, host-synchronised).  The many-query call allocates per-call buffers (queries, partial lists, the gathered lists:
// Q x topk x 16 B x world) and launches a GEMM before its all-gather: a rank that runs out of memory there has nothing to send and
// nothing to receive into, so it cannot "take part with the mark" the way a tick does -- instead all ranks agree first and, unless
// every rank is ready, ALL of them skip the large collective.  Costs one latency-bound collective per many-query call
// (milliseconds of GEMM).  Only an RCCL / HIP error of the agreement itself is hard.
static int xchg_agree(Ctx *c, bool ok_local, bool *all_ok)
{
    Exchange *x = c->xchg;
    x->agree_host[0] = ok_local ? 1 : 0;
    CHIP_HIP(c, hipMemcpyAsync(x->agree_dev, x->agree_host, sizeof(int32_t), hipMemcpyHostToDevice, c->s_query));
    CHIP_NCCL(c, rccl().AllGather(x->agree_dev, x->agree_dev + 1, sizeof(int32_t), ncclChar, x->comm, c->s_query));
    CHIP_HIP(c, hipMemcpyAsync(x->agree_host + 1, x->agree_dev + 1, sizeof(int32_t) * (size_t)x->world, hipMemcpyDeviceToHost, c->s_query));
    CHIP_HIP(c, hipStreamSynchronize(c->s_query));
    bool all = true;
    for (int r = 0; r < x->world; r++) all = all && x->agree_host[1 + r] == 1;
    *all_ok = all;
    return CHIP_OK;
}

// The same on a sharded ctx of the one-process-per-GPU layout with its exchange inside the library: local pass -> agreement ->
// ncclAllGather of the [Qpad][topk] lists -> merge on every rank (collective: every rank makes the same call).  A rank whose own
// arguments are out of range, whose allocations fail or whose enqueue fails says so in the agreement round: then NO rank posts the
// large all-gather, the failing rank returns its own status and the others CHIP_ERR_SHARD_FAILED -- the communicator stays in step.
int xchg_query_batch(Ctx *c, int64_t k, const float *queries, int32_t Q, int32_t topk, float *scores, int64_t *idx, bool fail_local)
{
    Exchange *x = c->xchg;
    if (c->elem != 4) return CHIP_ERR_UNSUPPORTED;      // the storage type is the same on every rank (same append stream)
    if (test_fail_now(x)) fail_local = true;
    chip_topk_entry *mine = nullptr, *gathered = nullptr, *merged = nullptr;
    int32_t Qpad = batch_qpad(Q);
    int rc_local = CHIP_OK;
    if (!fail_local) {
        rc_local = test_oom_rank() == c->rank ? CHIP_ERR_OOM : batch_exchange_buffers(c, x->world, Qpad, topk, &gathered, &merged, nullptr);
        if (rc_local == CHIP_OK) rc_local = batch_local_enqueue(c, k, queries, Q, topk, &mine, &Qpad);
        if (rc_local != CHIP_OK) fail_local = true;
    }
    bool all_ok = false;
    const int arc = xchg_agree(c, !fail_local, &all_ok);
    if (arc != CHIP_OK) return arc;
    if (!all_ok) {
        (void)hipStreamSynchronize(c->s_scan);          // whatever this rank's local pass enqueued has drained before the buffers are reused
        return rc_local != CHIP_OK ? rc_local : CHIP_ERR_SHARD_FAILED;
    }
    const size_t list_entries = (size_t)Qpad * topk;
    CHIP_NCCL(c, rccl().AllGather(mine, gathered, sizeof(chip_topk_entry) * list_entries, ncclChar, x->comm, c->s_scan));
    const int rc = batch_merge_lists(c, c->s_scan, gathered, x->world, Qpad, Q, topk, merged);
    if (rc != CHIP_OK) return rc;
    return batch_deliver(c, merged, Q, topk, scores, idx);
}

int group_synchronize(Ctx *gc)
{
    for (chip_ctx *s : gc->group->subs) {
        const int rc = chip_synchronize(s);
        if (rc != CHIP_OK) return rc;
    }
    return CHIP_OK;
}

int group_profile_enable(Ctx *gc, int on)
{
    for (chip_ctx *s : gc->group->subs) {
        std::lock_guard<std::mutex> qlk(s->query_mu);
        s->prof_on = on != 0;
    }
    return CHIP_OK;
}

}  // namespace chip

using namespace chip;

extern "C" {

int chip_last_comm_error(const chip_ctx *ctx, const char **text)
{
    if (!ctx) return (int)ncclInvalidArgument;
    int r = ctx->last_comm;
    if (ctx->group)
        for (const chip_ctx *s : ctx->group->subs)
            if (s->last_comm) r = s->last_comm;
    if (text) {
        if (r == kCommInitTimedOut) *text = "RCCL bootstrap (ncclCommInitAll / ncclCommInitRank) did not return within CHIP_COMM_INIT_TIMEOUT_MS: abandoned";
        else *text = rccl().ok ? rccl().GetErrorString((ncclResult_t)r) : (r ? "librccl could not be loaded (dlopen): no RCCL exchange available" : "no error");
    }
    return r;
}

int chip_create_multi(chip_ctx **out, int32_t D, int64_t capacity_hint, const int32_t *devices, int32_t n_devices, uint32_t flags)
{
    if (!out) return CHIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64 || capacity_hint < 0) return CHIP_ERR_INVALID_ARG;
    if (flags & ~(kCreateStoreMask | CHIP_MULTI_EXCHANGE_COPY)) return CHIP_ERR_INVALID_ARG;
    bool repeated = false;
    for (int a = 0; a < n_devices; a++)
        for (int b = a + 1; b < n_devices; b++) repeated = repeated || devices[a] == devices[b];
#ifdef CHIP_TEST_HOOKS   // test build only: the RCCL transport for a group whose devices repeat -- real RCCL refuses that, the stand-in of tests/fakerccl does not
    if (repeated && std::getenv("CHIP_TEST_RCCL_SAME_DEVICE")) {
        std::fprintf(stderr, "[cerebro_hip] TEST HOOK ACTIVE: CHIP_TEST_RCCL_SAME_DEVICE -- ncclCommInitAll over repeated devices\n");
        repeated = false;
    }
#endif
    chip_ctx *gc = new (std::nothrow) chip_ctx();
    if (!gc) return CHIP_ERR_OOM;
    Group *G = new (std::nothrow) Group();
    if (!G) { delete gc; return CHIP_ERR_OOM; }
    gc->group = G;
    gc->D = D;
    gc->nranks = n_devices;
    gc->device = devices[0];
    G->transport = ((flags & CHIP_MULTI_EXCHANGE_COPY) || repeated) ? CHIP_EXCHANGE_COPY : CHIP_EXCHANGE_RCCL;
    gc->group_transport = G->transport;
    int rc = CHIP_OK;
    for (int g = 0; g < n_devices && rc == CHIP_OK; g++) {
        chip_ctx *s = nullptr;
        rc = ctx_create(&s, D, capacity_hint, devices[g], g, n_devices, flags & kCreateStoreMask);
        if (rc != CHIP_OK) break;
        s->parent = gc;
        s->scan_reserve = env_int("CHIP_SCAN_RESERVE", 4);   // the exchange + merges run underneath the scans on every device, also at G = 1
        G->subs.push_back(s);
        G->same_dev.push_back(devices[g] == devices[0] ? 1 : 0);
        // RCCL: every rank receives the gathered lists; copies: only the root's buffer is used
        rc = exchange_create(s, n_devices, G->transport == CHIP_EXCHANGE_RCCL || g == 0);
    }
    if (rc == CHIP_OK && G->transport == CHIP_EXCHANGE_RCCL) {
        int r = (int)ncclSystemError;          // no librccl: copy exchange
        std::shared_ptr<CommInitJob> job;
        if (rccl().ok) {
            job = std::make_shared<CommInitJob>();
            job->all = true;
            job->devices.assign(devices, devices + n_devices);
            job->comms.assign((size_t)n_devices, nullptr);
            if (comm_init_run(job)) r = (int)job->result;
            else { r = kCommInitTimedOut; gc->comm_init_abandoned = 1; }
        }
        if (r != (int)ncclSuccess) {
            // No communicator (RCCL absent / misconfigured, peer access refused, a bootstrap that hangs ...): the lists are 384 B per
            // device and tick, so the device-copy exchange is a full substitute -- fall back to it instead of failing (or hanging)
            // the create; the reason stays readable through chip_last_comm_error, chip_get_info reports CHIP_EXCHANGE_COPY.
            gc->last_comm = r;
            G->transport = CHIP_EXCHANGE_COPY;
            gc->group_transport = CHIP_EXCHANGE_COPY;
            (void)hipSetDevice(devices[0]);
        } else
            for (int g = 0; g < n_devices; g++) G->subs[(size_t)g]->xchg->comm = job->comms[(size_t)g];
    }
    if (rc == CHIP_OK) {
        G->workers.assign((size_t)n_devices, nullptr);
        for (int g = 1; g < n_devices; g++) {
            Worker *w = new (std::nothrow) Worker();
            if (!w) { rc = CHIP_ERR_OOM; break; }
            G->workers[(size_t)g] = w;
            w->th = std::thread(worker_main, w, devices[g]);
        }
    }
    if (rc != CHIP_OK) {
        // keep the communicator error readable for the caller of a failed create: there is no ctx to ask, so it is lost --
        // chip_strerror(CHIP_ERR_COMM) is all they get
        ctx_destroy(gc);
        return rc;
    }
    mirror_state(gc);
    *out = gc;
    return CHIP_OK;
}

int chip_comm_unique_id(void *id_out)
{
    if (!id_out) return CHIP_ERR_INVALID_ARG;
    ncclUniqueId id;
    if (!rccl().ok || rccl().GetUniqueId(&id) != ncclSuccess) return CHIP_ERR_COMM;
    std::memcpy(id_out, &id, sizeof id);
    return CHIP_OK;
}

int chip_comm_init_rank(chip_ctx *c, const void *id_in, int32_t n_ranks, int32_t rank)
{
    if (!c || !id_in) return CHIP_ERR_INVALID_ARG;
    if (c->group || c->xchg) return CHIP_ERR_UNSUPPORTED;
    if (n_ranks != c->nranks || rank != c->rank) return CHIP_ERR_INVALID_ARG;   // the communicator must match the shard layout of chip_create
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    if (!c->own_query_stream) return CHIP_ERR_UNSUPPORTED;   // chip_set_stream and an in-library exchange exclude each other
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof id);
    if (!rccl().ok) { c->last_comm = (int)ncclSystemError; return CHIP_ERR_COMM; }
    auto job = std::make_shared<CommInitJob>();
    job->id = id;
    job->n_ranks = n_ranks;
    job->rank = rank;
    job->device = c->device;
    job->comms.assign(1, nullptr);
    if (!comm_init_run(job)) { c->last_comm = kCommInitTimedOut; c->comm_init_abandoned = 1; return CHIP_ERR_COMM; }
    if (job->result != ncclSuccess) { c->last_comm = (int)job->result; return CHIP_ERR_COMM; }
    ncclComm_t comm = job->comms[0];
    const int rc = exchange_create(c, n_ranks, true);
    if (rc != CHIP_OK) { (void)rccl().CommDestroy(comm); exchange_destroy(c); return rc; }
    c->xchg->comm = comm;
    // three small kernels per tick now go through the ctx stream underneath the scans: keep workgroup slots free for them
    c->scan_reserve = env_int("CHIP_SCAN_RESERVE", 4);
    return CHIP_OK;
}

}  // extern "C"
