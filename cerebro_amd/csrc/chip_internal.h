// chip_internal.h -- private definitions shared by the translation units of libcerebro_hip.so.
// gfx950 / CDNA4 only: wave = 64 lanes, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <mutex>
#include <condition_variable>
#include <vector>
#include "../../include/cerebro_hip.h"

namespace chip {

constexpr int kWave = 64;
constexpr int kMaxSegs = 4096;          // segment table entries (device resident)
constexpr int64_t kSegBytesTarget = 512ll << 20;  // ~512 MiB per DB segment

// ---- launch argument blocks (passed by value; all members wave-uniform) ----
struct ScanArgs {
    const void *const *seg_table;   // device array of segment base pointers (rows of float or double: Ctx::elem)
    int32_t seg_shift;              // local row r lives in segment r >> seg_shift ...
    int64_t seg_mask;               // ... at row (r & seg_mask)
    int64_t n_rows;                 // local rows [0, n_rows) are scanned
    int32_t D;
    int32_t K;
    const void *q[CHIP_MAX_NQ];     // query descriptors (device, storage type, D each, 16-B aligned)
    int64_t idx_mul, idx_add;       // global index = local * idx_mul + idx_add  (round-robin shard map)
    chip_topk_entry *partial;       // [gridDim.x][NQ][K]
    int32_t q64 = 0;                // queries staged in LDS as fp64 (scan_q64)
    int32_t rows_form = 0;          // > 0: row-batched kernel with this many rows per wave in flight (scan_rows_form)
    int32_t plain_loads = 0;        // row-batched kernel: temporal loads (the prefix fits the Infinity Cache and is re-read every tick)
    int32_t dyn_claim = 0;          // row-batched kernel, R = 1: the waves of a workgroup CLAIM their rows from an LDS counter instead of the static map (CHIP_SCAN_CLAIM=1)
    unsigned long long *stamps = nullptr;   // tuning only (CHIP_SCAN_STAMPS): 4 wall-clock stamps per wave of the row-batched kernel
    // fused tick (row-batched kernel, plain ctx): the LAST workgroup to finish reduces the per-workgroup best entries and writes the
    // decision record itself -- a tick is then ONE launch.  fused_result == nullptr: lists only, K2 follows as a launch of its own.
    chip_tick_result *fused_result = nullptr;
    int32_t *fused_ticket = nullptr;        // arrival counter of this launch's list buffer (0 at launch, reset by the last workgroup)
    uint32_t *pair_ctr = nullptr;           // [gridDim / 2] row-claim counters shared by workgroups b and b + gridDim / 2 (0 at launch, reset by the last workgroup); nullptr: rows are claimed per workgroup
    unsigned long long *fused_seq = nullptr; // completion word of the tick's slot in pinned host memory: written (system-scope RELEASE) after the
    unsigned long long fused_seq_val = 0;    // record, so that a host that polls it sees the record complete (chip_api.hip tick_collect_slot)
    int64_t tick_l = 0;
    int32_t locality = 0;
    double thresh = 0.0;
};

// ---- resident scan (opt-in: CHIP_TICK_RESIDENT=1; kernels.hip db_scan_resident, chip_api.hip resident_*) ----
// A synchronous tick over a cache-sized prefix is ~4 us of host API up to the doorbell + ~3 us from the doorbell to the first wave in
// front of a ~29 us kernel (profiles/r05_short_scan.md).  With the mode on, ONE instance of the row-batched scan stays on the chip (one
// workgroup per CU, the shape such a tick launches anyway) and runs the scan body once per COMMAND: the host writes a 64-byte line in
// pinned memory, workgroup 0 polls it over PCIe and relays it to the other workgroups through device memory.
struct ResidentCmd {           // exactly one 64-byte line: read by the device with ONE wave load, accepted when head == tail
    uint32_t head;             // command number, written first ...
    int32_t locality;
    int64_t n_rows;            // rows of the prefix;  < 0: leave
    int64_t tick_l;            // the tick's l (queries: rows l-1, l-2, l-3)
    double thresh;
    uint64_t result;           // chip_tick_result * (device address of the slot's pinned record)
    uint64_t seq_ptr;          // the slot's completion word (device address) ...
    uint64_t seq_val;          // ... and what to store there
    uint32_t dyn_claim;
    uint32_t tail;             // ... == head, written last
};
static_assert(sizeof(ResidentCmd) == 64, "one cache line, one PCIe read");
constexpr uint32_t kResidentLeave = 0xffffffffu;   // head / tail of the line that sends the workgroups home (never a command number)

struct ResidentArgs {
    ScanArgs base;                    // everything a command does not carry (segment table, D, lists, ticket, shard map)
    const uint32_t *cmd_host;         // the ResidentCmd line the host writes: pinned host memory, or device memory behind a large PCIe BAR
    uint32_t *cmd_dev;                // one 64-byte line per workgroup in device memory: workgroup 0 copies the accepted line into each
    unsigned long long *exit_host;    // the instance stores its id here (system-scope release) when it leaves
    unsigned long long instance;
    unsigned long long lease_ticks;   // 100 MHz ticks without a command after which the instance leaves by itself (a dead host must
                                      // not hold the chip; a live one relaunches on its next tick)
    uint32_t done;                    // number of the last command completed before this instance was launched
    uint32_t direct;                  // 1: the HOST writes every workgroup's line (device memory behind the PCIe BAR); workgroup 0 relays
                                      //    nothing but its own decision to leave
};

struct MergeArgs {
    const chip_topk_entry *in;      // [n_lists][NQ][K]
    int32_t n_lists;
    int32_t K;
    chip_topk_entry *out;           // [NQ][K]   (may be null)
    chip_tick_result *result;       // decision record (may be null => no decision)
    int64_t l;                      // tick position (for idx_curr)
    int32_t locality;
    double thresh;
};

struct Ctx;

// kernels.hip
int launch_resident(Ctx *c, hipStream_t s, const ResidentArgs &ra, int grid);   // CHIP_ERR_UNSUPPORTED in a build without the rows form
int launch_scan(Ctx *c, hipStream_t s, const ScanArgs &a, int nq, int grid);
int launch_merge(Ctx *c, hipStream_t s, const MergeArgs &a, int nq);
int scan_grid_for(const Ctx *c, int64_t n_rows, int nq, bool q64);
bool scan_q64(const Ctx *c, int nq, bool long_scan);
int scan_rows_form(const Ctx *c, int64_t n_rows, int nq, int grid, bool q64, bool sync_tick = false);
int scan_forms_built();              // CHIP_SCAN_FORM_* bits of this build (-DCHIP_NO_ROWS_FORM leaves the row-batched kernel out)
int launch_scores(Ctx *c, hipStream_t s, const ScanArgs &a, double *out_dev);   // K1s: all scores of one query, out[local row]
int launch_store_rows(Ctx *c, hipStream_t s, const void *src, int src_elem, int64_t n, int64_t first_global, uint32_t *flags_dev, bool write_ring,
                      int64_t row_stride = 1);
int launch_synth(Ctx *c, hipStream_t s, int64_t first_global, int64_t n, uint64_t seed,
                 const int64_t *plant_dst_dev, const int64_t *plant_src_dev, const int32_t *plant_kind_dev, int64_t n_plant, int unit);

struct Slot {
    hipEvent_t done = nullptr;
    hipEvent_t wait_ev = nullptr;       // what collect waits for: `done`, or the merge event of the tick's list buffer (same-stream ticks)
    chip_tick_result *host = nullptr;   // pinned
    chip_tick_result *dev = nullptr;
    bool in_flight = false;
    bool immediate = false;             // result already final on host (skipped / too short)
    int64_t prev_last_l = 0;            // last_l before this tick was enqueued ...
    int64_t tick_l = 0;                 // ... the l this tick committed ...
    unsigned long long *seq_host = nullptr, *seq_dev = nullptr;   // completion word of this slot (pinned, device-mapped)
    unsigned long long seq_want = 0;    // value the fused tick in flight will store there
    bool poll = false;                  // collect by polling seq_host instead of waiting for the event
    bool resident = false;              // the tick went to the resident scan instance: there is no event, the completion word is all
    int64_t *last_l_ptr = nullptr;      // ... and where to restore it if the tick comes back CHIP_TICK_FAILED (only while no newer
                                        //     tick has been enqueued: *last_l_ptr == tick_l)
};

struct Exchange;   // chip_multi.hip: how a sharded ctx trades its per-shard top-k lists (RCCL communicator / device copies)
struct Group;      // chip_multi.hip: a ctx made of G per-device sub-contexts driven from one process

struct Ctx {
    int32_t D = 0;
    int32_t device = 0;
    int32_t rank = 0, nranks = 1;
    int32_t n_cus = 0;
    char arch[32] = {0};

    // --- storage type: 4 = float rows (default), 8 = double rows.  store_auto: still undecided -- the first append of a
    //     descriptor that is not float32-representable into the EMPTY DB switches to double (chip_create semantics).
    int32_t elem = 4;
    bool store_auto = true;

    // --- DB storage: fixed-size segments, never moved once allocated ---
    int32_t seg_shift = 0;
    int64_t seg_rows = 0;
    std::vector<void *> segs;            // host copy of the table
    void **seg_table_dev = nullptr;      // device table [kMaxSegs]
    void *ring_dev = nullptr;            // [CHIP_RING_ROWS][D]  most recent rows (all ranks)
    int64_t rows_global = 0;             // published length (guarded by mu)
    int64_t rows_local = 0;
    int64_t lossy_rows = 0;
    mutable std::mutex mu;               // guards rows_*, segs growth
    std::mutex append_mu;                // serialises appenders
    mutable std::mutex query_mu;         // serialises queriers (scratch buffers are per ctx); guards last_l
    std::mutex ring_mu;                  // sharded: ring residency check .. scan event  vs  the appender's ring writes
    int64_t rows_pending = 0;            // length an in-flight append will publish (ring_mu)
    hipEvent_t last_scan_ev[2] = {};     // newest scan per scan stream (ring_mu)
    std::mutex pnp_mu;
    std::mutex icp_mu;            // ICP has its own stream and lock: it may run underneath a PnP call

    // --- streams ---
    hipStream_t s_query = nullptr, s_append = nullptr, s_pnp = nullptr;
    bool own_query_stream = true;

    // --- append staging ---
    void *stage_dev = nullptr;           // staging for host descriptors
    size_t stage_bytes = 0;
    uint32_t *flags_dev = nullptr;       // bit0: not-f32-representable, bit1: non-finite
    uint32_t *flags_host = nullptr;      // pinned

    // --- query scratch ---
    // Scans run back-to-back on s_scan; the merge of tick i runs on the ctx stream (s_query) behind ev_scan[b], so
    // it (and, sharded, the all-gather + global merge that follow it on the ctx stream) overlaps the scan of tick
    // i+1.  Partial lists live in a ring of kRing buffers; a scan waits for the merge that last read its buffer.
    // Deeper than the number of ticks a caller can keep in flight (CHIP_MAX_INFLIGHT): in steady state the merge that last
    // read a buffer is long complete when its next scan is enqueued, so no barrier packet is put in front of the scan
    // (each one costs the scan stream ~7 us; 64 x 512 KiB of HBM is nothing).
    static constexpr int kRing = 64;
    hipStream_t s_scan = nullptr;
    hipStream_t s_scan2 = nullptr;   // second scan stream for short scans (CHIP_SCAN_STREAMS=1 disables)
    hipStream_t s_scan_x[2] = {};    // third / fourth stream of the same-stream short tick (CHIP_SCAN_STREAMS, default 4)
    uint64_t n_same_stream = 0;      // same-stream ticks so far (round robin over the tick streams)
    chip_topk_entry *partial_dev[kRing] = {};   // [max_grid][CHIP_MAX_NQ][CHIP_MAX_TOPK]
    int32_t partial_lists[kRing] = {};                               // grid of the scan that filled it
    hipEvent_t ev_scan[kRing] = {};             // scan into buffer b finished
    hipEvent_t ev_merged[kRing] = {};           // merge out of buffer b finished
    int32_t *tickets_dev = nullptr;             // [kRing] arrival counters of the fused tick (one per list buffer)
    uint32_t *pair_ctr_dev = nullptr;           // [kRing][kPairCtrs][kPairStride] row-claim counters of workgroup pairs (pair-claimed stream of the fused tick)
    static constexpr int kPairCtrs = 256;
    static constexpr int kPairStride = 32;      // uint32s: every counter has a 128-byte line to itself (the pairs of a launch sit on eight XCDs)
    unsigned long long *seq_host_all = nullptr; // [CHIP_MAX_INFLIGHT] completion words of the slots (pinned)
    unsigned long long tick_seq = 0;            // last value handed out
    bool tick_poll = true;                      // CHIP_TICK_POLL: fused ticks are collected by polling the completion word
    unsigned long long *next_seq_dev = nullptr; // set by tick_enqueue_slot for the enqueue that follows; taken (and cleared) by a fused launch
    unsigned long long next_seq_val = 0;
    bool last_enqueue_fused = false;
    bool tick_fused = true;                     // CHIP_TICK_FUSED=0 disables
    // resident scan instance (CHIP_TICK_RESIDENT=1; chip_api.hip resident_*): guarded by query_mu like the slots
    bool tick_resident = false;
    double res_max_bytes = 512.0 * 1024 * 1024; // prefixes up to this size take it (CHIP_RESIDENT_MAX_MIB)
    int32_t res_lease_ms = 250;                 // CHIP_RESIDENT_LEASE_MS
    hipStream_t s_resident = nullptr;
    void *res_pinned = nullptr, *res_cmd_vram = nullptr;   // the allocations behind the pointers below
    bool res_cmd_in_vram = false;               // the command line is in device memory, written by the host through the PCIe BAR
    bool res_direct = false;                    // ... and so is every workgroup's own line (no relay): CHIP_RESIDENT_BAR=2
    ResidentCmd res_pending{};                  // the command in flight (re-posted to an instance launched for it)
    int64_t res_test_skip_master = 0;           // test hook (CHIP_TEST_RESIDENT_SKIP_MASTER)
    bool res_test_skipped = false;
    ResidentCmd *res_cmd_host = nullptr;        // the line as the host writes it + its device address
    uint32_t *res_cmd_hostdev = nullptr;
    uint32_t *res_cmd_dev = nullptr;
    unsigned long long *res_exit_host = nullptr, *res_exit_hostdev = nullptr;
    chip_topk_entry *res_partial = nullptr;
    int32_t *res_ticket = nullptr;
    unsigned long long res_instance = 0;        // id of the instance launched last (0: none yet)
    bool res_alive = false;                     // launched and not yet seen to have left
    bool res_ready = false;                     // resident_alloc has completed: every buffer below exists (published LAST)
    int32_t res_inhibit = 0;                    // > 0: no instance may be launched (resident_pause .. resident_resume); ticks are launched meanwhile
    bool res_busy = false;                      // a command is in flight (one at a time)
    uint32_t res_cmd_no = 0, res_done = 0;      // last number posted / last number seen complete
    int32_t res_grid = 0;
    Slot *res_slot = nullptr;                   // the slot of the command in flight
    int64_t res_ticks = 0, res_launches = 0;    // ticks it has served / instances launched (chip_debug_resident_stats)
    std::mutex res_mu;                          // the instance's state: queriers (query_mu held) and the appender (append_mu held) take it last
    uint64_t n_enqueued = 0;
    int32_t max_grid = 0;
    chip_topk_entry *topk_dev = nullptr;      // [CHIP_MAX_NQ][CHIP_MAX_TOPK]
    chip_topk_entry *topk_host = nullptr;     // pinned, device-visible: the last block writes results straight here

    void *qvec_dev = nullptr;                 // [CHIP_MAX_NQ][D] external query vectors (storage type)
    double *scores_dev = nullptr;             // chip_query_scores scratch (grown on demand)
    unsigned long long *stamps_dev = nullptr; // tuning only (CHIP_SCAN_STAMPS=1): [max_grid * 16 waves][4]
    int64_t scores_cap = 0;

    // --- sharded tick inside the library (chip_multi.hip) ---
    std::vector<char> qconv;                  // host conversion scratch of external query vectors
    int64_t cap_hint = 0;
    int32_t group_transport = 0;              // CHIP_EXCHANGE_* of a group ctx
    mutable int last_comm = 0;                // last ncclResult_t (1000: a bootstrap was abandoned at its deadline)
    int32_t comm_init_abandoned = 0;          // a helper thread of this ctx is still inside an RCCL bootstrap (chip_multi.hip)
    Exchange *xchg = nullptr;                 // non-null: chip_loop_tick* work on this sharded ctx
    Group *group = nullptr;                   // non-null: this ctx is a group of per-device sub-contexts
    Ctx *parent = nullptr;                    // sub-context of a group
    Slot slots[CHIP_MAX_INFLIGHT];
    int64_t last_l = 0;

    // --- scan tuning (resolved at create; CHIP_SCAN_* env overrides for A/B runs) ---
    int32_t scan_block = 0;       // 0 = auto
    int32_t scan_blocks_per_cu = 2;
    int32_t scan_reserve = 0;
    int32_t scan_variant = 0;
    bool tick_same_stream = true;  // short ticks of a plain ctx: merge on the scan's stream (CHIP_TICK_SAME_STREAM=0 disables)
    int32_t scan_depth = 1;       // CHIP_SCAN_DEPTH: batches of a wave's load stream in flight in the claimed row-batched kernel (1 or 2)
    int32_t scan_claim = -1;      // CHIP_SCAN_CLAIM: rows claimed within the workgroup (row-batched kernel, R = 1): -1 auto, 0 never (static row -> wave map), 1 always
    int32_t scan_rows = 0;        // CHIP_SCAN_ROWS: 0 = auto (prefixes up to scan_plain_bytes), 1..3 = row-batched kernel with that R for every scan, -1 = never
    double scan_plain_bytes = 768.0 * 1024 * 1024;   // prefixes up to this size: rows form, R = 1, temporal loads (CHIP_SCAN_PLAIN_MIB)
    double scan_sync_plain_bytes = 4096.0 * 1024 * 1024;   // SYNCHRONOUS ticks: the row-batched (fused, one-launch) form up to this size (CHIP_SCAN_SYNC_PLAIN_MIB)
    bool tick_sync_now = false;      // the tick being enqueued is a synchronous chip_loop_tick (query_mu held): form chosen for latency, not throughput
    double scan_half_bytes = 192.0 * 1024 * 1024;    // prefixes up to this size: launches take half of every CU's workgroup slots (CHIP_SCAN_HALF_MIB)
    int32_t scan_short_bpc = 1;       // workgroups per CU of a launch over a cache-sized prefix (CHIP_SCAN_SHORT_BPC; 0 = as any other)
    double scan_overlap_bytes = 8.0 * 1024 * 1024 * 1024;   // launches up to this size alternate between the two scan streams (CHIP_SCAN_OVERLAP_GIB)

    // --- profiling ---
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;     // start/stop pairs
    size_t prof_used = 0;
    double prof_bytes_last = 0.0;

    // --- pnp scratch (pnp.hip) ---
    void *pnp_state = nullptr;
    void *icp_state = nullptr;   // icp.hip, created on first use
    void *batch_state = nullptr; // batch.hip, created on first use

    mutable hipError_t last_hip = hipSuccess;
};

#define CHIP_HIP(ctx, expr)                                  \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) {                              \
            (ctx)->last_hip = _e;                            \
            return _e == hipErrorOutOfMemory ? CHIP_ERR_OOM : CHIP_ERR_HIP; \
        }                                                    \
    } while (0)

// global index <-> local row (round-robin sharding)
inline bool owns_row(const Ctx *c, int64_t g) { return c->nranks == 1 || (g % c->nranks) == c->rank; }
inline int64_t local_of(const Ctx *c, int64_t g) { return c->nranks == 1 ? g : g / c->nranks; }
// number of local rows among global rows [0,k)
inline int64_t local_count(const Ctx *c, int64_t k) {
    if (c->nranks == 1) return k;
    return k > c->rank ? (k - c->rank + c->nranks - 1) / c->nranks : 0;
}
inline char *row_ptr_host(const Ctx *c, int64_t local) {
    return static_cast<char *>(c->segs[(size_t)(local >> c->seg_shift)]) + (local & (c->seg_rows - 1)) * (int64_t)c->D * c->elem;
}
inline char *ring_ptr(const Ctx *c, int64_t g) {
    return static_cast<char *>(c->ring_dev) + (g % CHIP_RING_ROWS) * (int64_t)c->D * c->elem;
}

// ---- chip_api.hip: single-ctx building blocks, shared with chip_multi.hip ----
struct RingGuard {   // sharded ctx: residency check .. scan event recorded, vs the appender's ring writes (see query_row_ptrs)
    std::unique_lock<std::mutex> lk;
    explicit RingGuard(Ctx *c) : lk(c->ring_mu, std::defer_lock) { if (c->ring_dev) lk.lock(); }
};
int ctx_create(chip_ctx **out, int32_t D, int64_t capacity_hint, int32_t device, int32_t rank, int32_t nranks, uint32_t flags);
void ctx_destroy(chip_ctx *c);
int ctx_append(Ctx *c, const void *desc, int src_elem, int64_t n, uint32_t flags, int64_t *first_index);
// the phases of an append, for callers that must take the same decision on several contexts (group_append): reserve -> store the
// DB rows (unpublished; owner_only: upload only the rows this shard owns) -> [switch an empty undecided DB to double rows] ->
// mirror the newest rows into the ring and publish the length.  The caller holds the append lock.
int append_reserve(Ctx *c, int64_t first, int64_t n);
int append_store_db(Ctx *c, const void *desc, int src_elem, int64_t first, int64_t n, bool owner_only, uint32_t *bad);
bool append_can_switch_to_double(const Ctx *c, int64_t first);
int append_switch_to_double(Ctx *c, int64_t n);
int append_ring_publish(Ctx *c, const void *desc, int src_elem, int64_t first, int64_t n, bool lossy);
int synth_generate(Ctx *c, int64_t first, int64_t n, uint64_t seed, const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant, int unit);
void append_publish(Ctx *c, int64_t new_total, bool lossy, int64_t n);
int64_t published_rows(const Ctx *c);
int ctx_append_synthetic(Ctx *c, int64_t n, uint64_t seed, const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant, int unit);
int ctx_read_row(Ctx *c, int64_t g, int64_t total, void *out);          // one row, storage type, async on s_query
int query_row_ptrs(Ctx *c, const int64_t *rows, int nq, int64_t n_global, const void **q);
int upload_query_vectors(Ctx *c, const void *queries, int src_elem, int nq, const void **q);   // -> qvec_dev (on s_scan)
int enqueue_scan_merge(Ctx *c, int64_t k, const void *const *q, int nq, int K, int64_t l, const chip_dot_params *p,
                       chip_topk_entry *out, chip_tick_result *res, bool tick, hipStream_t *merge_stream, hipEvent_t *merged_ev = nullptr);
int merge_enqueue_slot(Ctx *c, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists, int32_t topk, Slot &s);
int merge_enqueue_out(Ctx *c, const void *dev_gathered, int32_t n_lists, int nq, int32_t topk, chip_topk_entry *out);
int tick_prepare(int64_t rows_global, int64_t last_l, int64_t l, const chip_dot_params *p, int32_t *status, int64_t *k_out);
void fill_immediate(chip_tick_result *r, int32_t status);
int tick_collect_slot(Ctx *c, Slot &s, chip_tick_result *out);
int sync_topk_out(Ctx *c, int nq, int K, double *scores, int64_t *idx);
int ctx_scores_local(Ctx *c, int64_t k, const void *q, double *u_global, int64_t stride_mul, int64_t stride_add);
int env_int(const char *name, int dflt);
void resident_stop(Ctx *c);    // retire the resident scan instance and wait until it has left the chip (no-op without one): called before
                               // anything that rewrites the segment table, frees device memory or wants the chip to itself
// resident_stop + NO new instance until resident_resume: ticks that arrive meanwhile are launched.  For sections that free / allocate
// device memory (hipFree waits for the whole device -- a tick thread relaunching the instance in between would make it wait for as long
// as ticks keep coming: ADVICE r5) or rewrite the segment table.  Nests; chip_resident_pause / _resume are the public form.
void resident_pause(Ctx *c);
void resident_resume(Ctx *c);
struct ResidentPause {          // scope guard: every early return of a CHIP_HIP(...) inside the section resumes
    Ctx *c;
    bool on;
    explicit ResidentPause(Ctx *ctx, bool cond = true) : c(ctx), on(cond) { if (on) resident_pause(c); }
    ~ResidentPause() { if (on) resident_resume(c); }
    ResidentPause(const ResidentPause &) = delete;
    ResidentPause &operator=(const ResidentPause &) = delete;
};
constexpr uint32_t kCreateStoreMask = 3u;   // CHIP_CREATE_STORE_F32 | CHIP_CREATE_STORE_F64

// ---- chip_multi.hip: exchange of per-shard lists inside the library (RCCL / device copies), groups of sub-contexts ----
constexpr int kXRing = 64;                  // >= CHIP_MAX_INFLIGHT: list buffers of a tick are not reused while it is in flight
constexpr int kListEntries = CHIP_MAX_NQ * CHIP_MAX_TOPK;
void exchange_destroy(Ctx *c);
int exchange_comm_ranks(const Ctx *c);      // ncclCommCount of the attached communicator, 0 if none
void group_destroy(Ctx *c);
// sharded ctx with an RCCL communicator attached (one process per GPU): scan -> local merge -> all-gather -> merge, all enqueued
int xchg_tick_enqueue(Ctx *c, int64_t l, int64_t k, const chip_dot_params *p, Slot &s, bool fail_local);
int xchg_query(Ctx *c, int64_t k, const void *const *q, int nq, int K, double *scores, int64_t *idx, bool fail_local);
int xchg_fetch_rows(Ctx *c, const int64_t *rows, int nq, int64_t n_local_published, const void **q, bool *fail_local);
// group ctx (chip_create_multi): every entry point of the header that makes sense on a group
int group_append(Ctx *gc, const void *desc, int src_elem, int64_t n, uint32_t flags, int64_t *first_index);
int group_append_synthetic(Ctx *gc, int64_t n, uint64_t seed, const int64_t *pd, const int64_t *ps, const int32_t *pk, int64_t n_plant, int unit);
int group_read_rows(Ctx *gc, const int64_t *rows, int64_t n, void *out, int out_elem);
int group_tick_enqueue(Ctx *gc, int64_t l, const chip_dot_params *p, int32_t slot);
int group_tick_collect(Ctx *gc, int32_t slot, chip_tick_result *out);
int group_query(Ctx *gc, int64_t k, const int64_t *rows, const void *vectors, int vec_elem, int nq, int K, double *scores, int64_t *idx);
int group_scores(Ctx *gc, int64_t k, int64_t query_row, double *u);
int group_synchronize(Ctx *gc);
int group_profile_enable(Ctx *gc, int on);
Ctx *group_root(Ctx *gc);
int group_size(const Ctx *gc);

// batch.hip: the many-query (MFMA) mode in pieces, so that chip_multi.hip can run it over the shards of a DB
int32_t batch_qpad(int32_t Q);            // Q rounded up to the query-tile granularity (BM) of db_gemm_topk
int batch_local_enqueue(Ctx *c, int64_t k, const float *queries, int32_t Q, int32_t topk, chip_topk_entry **out_dev, int32_t *Qpad_out);
int batch_deliver(Ctx *c, const chip_topk_entry *list_dev, int32_t Q, int32_t topk, float *scores, int64_t *idx);
int batch_exchange_buffers(Ctx *c, int n_lists, int32_t Qpad, int32_t topk, chip_topk_entry **gathered, chip_topk_entry **merged, hipEvent_t *ev_done);
int batch_merge_lists(Ctx *c, hipStream_t s, const chip_topk_entry *in, int n_lists, int32_t Qpad, int32_t Q, int32_t topk, chip_topk_entry *out);
int group_query_batch(Ctx *gc, int64_t k, const float *queries, int32_t Q, int32_t topk, float *scores, int64_t *idx);
int xchg_query_batch(Ctx *c, int64_t k, const float *queries, int32_t Q, int32_t topk, float *scores, int64_t *idx, bool fail_local);

// pnp.hip
int pnp_create(Ctx *c);
void pnp_destroy(Ctx *c);
void icp_destroy(Ctx *c);
void batch_destroy(Ctx *c);

}  // namespace chip

struct chip_ctx : chip::Ctx {};
