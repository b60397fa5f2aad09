/*
 * cerebro_hip.h -- C-ABI of libcerebro_hip.so, the MI355X (gfx950) loop-detection core.
 *
 * The reference (mpkuse/cerebro) has no FFI seam for this path: the dot-product scan is inline in
 * Cerebro::descrip_N__dot__descrip_0_N (src/Cerebro.cpp:903-1103) and the pose verifier is the static C++
 * function StaticTheiaPoseCompute::PNP (src/DlsPnpWithRansac.cpp:132-245).  This header DEFINES the seam
 * at exactly those cut lines; INTEGRATION.md shows the few-line patch a maintainer applies
 * to Cerebro.cpp / DlsPnpWithRansac.cpp to call it.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch/Eigen types.
 *   - every function returns an int status: CHIP_OK (0) or a negative CHIP_ERR_*; nothing throws or aborts
 *     (the reference uses return false / -1 / exit(n); asserts are compiled out in Release, CMakeLists.txt:44).
 *   - the caller owns every in/out buffer; the library owns device memory; nothing is retained after return.
 *   - row index i of the descriptor DB == position i of Cerebro::wholeImageComputedList
 *     (src/Cerebro.cpp:321-326); rows are append-only and never reordered.  Indices are int64 here (the
 *     reference uses int).
 *   - thread safety: one appender thread (desc_th, cerebro_node.cpp:487), one querier thread
 *     (dot_product_th, :499) and one PnP caller (loopcandidate_consumer_th, :509) may use the same ctx
 *     concurrently.  A query only ever reads rows that were fully appended before the call.
 */
#ifndef CEREBRO_HIP_H
#define CEREBRO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHIP_ABI_VERSION 7

/* ------------------------------------------------------------------------------------------ status codes */
enum {
    CHIP_OK = 0,
    CHIP_ERR_INVALID_ARG = -1,
    CHIP_ERR_NO_DEVICE = -2,        /* no usable gfx950 device / HIP runtime failure at create            */
    CHIP_ERR_HIP = -3,              /* a HIP runtime call failed; chip_last_hip_error() has the hipError_t */
    CHIP_ERR_OOM = -4,
    CHIP_ERR_NOT_F32 = -5,          /* an appended float64 is not exactly representable as float32        */
    CHIP_ERR_NONFINITE = -6,        /* NaN/Inf in an appended descriptor (Eigen maxCoeff with NaN is unspecified) */
    CHIP_ERR_RANGE = -7,            /* l / k / row index outside the appended range                       */
    CHIP_ERR_UNSUPPORTED = -8,      /* e.g. D not a multiple of 4, topk > CHIP_MAX_TOPK, nq > CHIP_MAX_NQ  */
    CHIP_ERR_TOO_FEW_POINTS = -9,   /* PnP with < 20 correspondences (DlsPnpWithRansac.cpp:136-139 returns -1) */
    CHIP_ERR_BUSY = -10,            /* async slot still in flight / not enqueued                          */
    CHIP_ERR_COMM = -11,            /* an RCCL call failed; chip_last_comm_error() has the ncclResult_t    */
    CHIP_ERR_SHARD_FAILED = -12,    /* a shard of a sharded DB could not take part in this tick / query (e.g. its query rows had
                                       left its ring under a concurrent bulk append); EVERY rank gets this status for that call,
                                       the call had no effect (last_l is not advanced) and the exchange stays in step: retry */
    CHIP_ERR_GROUP_BROKEN = -13     /* an earlier call failed on some devices of a chip_create_multi ctx after they had
                                       diverged (or its communicator failed): the ctx refuses further work -- destroy it */
};

#define CHIP_MAX_TOPK 16
#define CHIP_MAX_NQ 4
#define CHIP_DEFAULT_TOPK 8

typedef struct chip_ctx chip_ctx;

const char *chip_strerror(int status);
int chip_abi_version(void);
/* last hipError_t seen by this ctx (0 = hipSuccess) and its hipGetErrorString text */
int chip_last_hip_error(const chip_ctx *ctx, const char **text);
/* last ncclResult_t seen by this ctx (0 = ncclSuccess) and its ncclGetErrorString text */
int chip_last_comm_error(const chip_ctx *ctx, const char **text);

/* ------------------------------------------------------------------------------------------ lifecycle
 * Replaces  MatrixXd M = MatrixXd::Zero(descriptor_size, 29000)  (src/Cerebro.cpp:946): device-resident,
 * fp32, row-major [row][D], growable (capacity_hint is only the initial reservation; the 29000 ceiling of
 * the reference is not reproduced).
 *
 * Sharding (BASELINE config 4): with shard_count = G > 1 the ctx of rank r stores global rows i with
 * i % G == r (round-robin keeps every prefix [0,k) balanced) plus a replicated ring of the most recent
 * CHIP_RING_ROWS rows (64 MiB at D=4096), from which the tick's three query descriptors are read: a tick at l
 * on a one-process-per-GPU ctx therefore needs chip_db_size() - l <= CHIP_RING_ROWS - 3 (else CHIP_ERR_RANGE /
 * CHIP_ERR_SHARD_FAILED) -- always true in live operation, where ticks trail the append head by a few rows; a
 * chip_create_multi ctx has no such limit (older query rows are fetched from the devices that own them, so a whole
 * recorded schedule can be replayed over a cold-started DB).  Every rank must be fed the same append
 * stream.  One process per GPU; the per-shard top-k lists are exchanged INSIDE the library once an RCCL
 * communicator is attached (chip_comm_init_rank below), or by the HOST (any transport) between
 * chip_scan_local and chip_merge_decide.  Query rows of chip_query_rows / chip_query_scores: with an exchange
 * attached (or on a chip_create_multi ctx) ANY appended row -- it is fetched from the shard that owns it; on a
 * sharded ctx without an exchange only rows still in this rank's ring (else CHIP_ERR_RANGE).                */
#define CHIP_RING_ROWS 4096
int  chip_create(chip_ctx **out, int32_t D, int64_t capacity_hint, int32_t device, int32_t shard_rank, int32_t shard_count);
void chip_destroy(chip_ctx *ctx);

/* Storage type of the DB rows.  The reference's M is MatrixXd (src/Cerebro.cpp:946); the default NetVLAD server emits
 * float32 VALUES on the float64 wire (whole_image_desc_compute_server.py:631,648), for which float rows are lossless and
 * half the HBM traffic; the ReljaNetVLAD model (:148-149, numpy matmul with the WPCA matrix, the only 4096-D model) emits
 * genuine float64, for which the rows must be double to keep the candidate selection bit-exact.
 *   chip_create / flags 0 : decided by the data -- float rows, unless the FIRST append (into the still empty DB) carries a
 *                           value that is not float32-representable: then the DB becomes a double-row DB and that append
 *                           succeeds unrounded.  Later non-representable values in a float DB fail (CHIP_ERR_NOT_F32).
 *   CHIP_CREATE_STORE_F32 : float rows, never switches.        CHIP_CREATE_STORE_F64 : double rows from the start.
 * Double rows: scores are fp64 FMA chains (one rounding per term) in the fixed order of DESIGN.md 3 with 2 elements per lane
 * per 128-element chunk; D <= 10 240 as for float rows (two double queries fit the 160 KiB of LDS; where the three of a
 * tick do not -- D > 6824, e.g. the reference's default 8192 -- the rest is read in place, same bits); the MFMA many-query
 * mode is float-only.                                                                                                   */
#define CHIP_CREATE_STORE_F32 1u
#define CHIP_CREATE_STORE_F64 2u
int  chip_create_ex(chip_ctx **out, int32_t D, int64_t capacity_hint, int32_t device, int32_t shard_rank, int32_t shard_count,
                    uint32_t flags);

/* ------------------------------------------------------------------------------------------ multi-GPU inside the library
 * (a) ONE process, G GPUs -- the shape of the reference: the loop-candidate producer is one thread of one process
 *     (src/cerebro_node.cpp:499, src/Cerebro.cpp:903).  chip_create_multi returns ONE ctx backed by G per-device
 *     sub-contexts (rows round-robin, row i on devices[i % G]).  The entry points of the reference's path work on it
 *     unchanged: chip_db_append_* (every device is SENT only the rows it owns plus the newest CHIP_RING_ROWS of the batch: a
 *     bulk load moves ~1x the batch over PCIe, not Gx; the validation decision is one, from all devices),
 *     chip_db_read_rows_*, chip_loop_tick* = G local scans -> per-device local top-k -> exchange -> merge + decision on
 *     devices[0], chip_query_rows / _vectors_* / _scores, chip_synchronize, chip_profile_*, PnP / ICP (on devices[0]).
 *     chip_query_batch_f32 (the MFMA many-query mode: one pass per device over its rows, the per-device lists merged on devices[0]).
 *     NOT on a group ctx (CHIP_ERR_UNSUPPORTED): chip_set_stream / chip_reset_stream, chip_scan_local, chip_merge_decide*,
 *     chip_comm_init_rank (the group owns its exchange).
 *     One host worker thread per device enqueues that device's work, so the host cost of a tick does not grow with G.
 *     Failure containment: a call that fails on SOME devices before anything became visible changes nothing; a shard that
 *     cannot take part in one tick sends a marked neutral list, the tick fails everywhere alike (CHIP_ERR_SHARD_FAILED) and the
 *     exchange stays in step; a failure after the devices diverged marks the ctx broken (CHIP_ERR_GROUP_BROKEN).
 *     Exchange: an RCCL communicator over the G devices (ncclCommInitAll; ncclAllGather of 3 x topk (score,index) entries per
 *     rank per tick, enqueued in-stream between the local and the global merge) when the devices are distinct; device
 *     copies (lists written / copied straight into the root's gather buffer behind events) when CHIP_MULTI_EXCHANGE_COPY is
 *     passed or the list names a device twice (RCCL refuses two ranks on one device) -- the latter lets a 1-GPU box run the
 *     G = 2..8 code path.  RCCL is loaded with dlopen (librccl.so.1, or $CHIP_RCCL_LIBRARY), the library does not link it: if it
 *     is absent, cannot build the communicator, or its bootstrap does not return within CHIP_COMM_INIT_TIMEOUT_MS (default
 *     120 s; the blocking rendezvous runs on a helper thread and is abandoned at the deadline) the create neither fails nor
 *     hangs, it falls back to the copy exchange (chip_get_info().exchange / .comm_ranks / .comm_init_abandoned tell which one
 *     is in use, over how many ranks, and whether a helper is still stuck; chip_last_comm_error() why).  chip_comm_init_rank
 *     has the same deadline: it returns CHIP_ERR_COMM instead of hanging.
 * (b) one process PER GPU (torchrun-style launch): create each rank's ctx with chip_create(.., shard_rank, shard_count),
 *     then attach an RCCL communicator: rank 0 calls chip_comm_unique_id, distributes the 128 bytes by any means, every
 *     rank calls chip_comm_init_rank.  From then on chip_loop_tick / _enqueue / _collect and chip_query_* work on the
 *     sharded ctx (collectively: every rank must make the same calls in the same order) and return the same result on
 *     every rank.  chip_scan_local / chip_merge_decide remain for callers that bring their own transport.          */
#define CHIP_MULTI_EXCHANGE_COPY 4u
int  chip_create_multi(chip_ctx **out, int32_t D, int64_t capacity_hint, const int32_t *devices, int32_t n_devices, uint32_t flags);
#define CHIP_COMM_ID_BYTES 128
int  chip_comm_unique_id(void *id_out /* CHIP_COMM_ID_BYTES */);
int  chip_comm_init_rank(chip_ctx *ctx, const void *id /* CHIP_COMM_ID_BYTES */, int32_t n_ranks, int32_t rank);
/* Make an externally owned hipStream_t (e.g. the stream torch.distributed synchronises its collectives with) the
 * ctx stream: synchronous queries, merges and the stream-ordering promises of chip_scan_local refer to it.  The value
 * is taken literally: NULL is HIP's null stream.  chip_reset_stream returns to the ctx's private stream. */
int  chip_set_stream(chip_ctx *ctx, void *hip_stream);
int  chip_reset_stream(chip_ctx *ctx);
int  chip_synchronize(chip_ctx *ctx);

/* ------------------------------------------------------------------------------------------ DB append
 * Replaces  M.col(_s) = ...getWholeImageDescriptor()  (src/Cerebro.cpp:1005-1006) and the f64 wire type of
 * WholeImageDescriptorCompute.srv:4 / Cerebro.cpp:268-271.  n descriptors, each D contiguous values.
 * The f64->f32 narrowing is done on the device and VERIFIED lossless ((double)(float)x == x); the default
 * NetVLAD server emits float32 values (whole_image_desc_compute_server.py:631,648) so this holds.
 * On CHIP_ERR_NOT_F32 / CHIP_ERR_NONFINITE nothing is appended.                                         */
#define CHIP_APPEND_ALLOW_ROUNDING 1u   /* the caller vouches that the values ARE float32 descriptors up to the precision they were
                                           printed / transmitted with (a state.json checkpoint: 15-digit text): round to nearest
                                           instead of failing -- also in an empty undecided DB, which then stays a float DB; sets
                                           info.lossy_rows.  Without the flag the data decides (chip_create above).             */
int chip_db_append_f64(chip_ctx *ctx, const double *desc, int64_t n, uint32_t flags, int64_t *first_index);
int chip_db_append_f32(chip_ctx *ctx, const float *desc, int64_t n, int64_t *first_index);
int64_t chip_db_size(const chip_ctx *ctx);           /* global number of appended rows (== l)             */
/* Read back rows (global indices; in sharded mode only rows owned by this rank or still in the ring). */
int chip_db_read_rows_f32(chip_ctx *ctx, const int64_t *rows, int64_t n, float *out);   /* CHIP_ERR_NOT_F32 on a double-row DB */
int chip_db_read_rows_f64(chip_ctx *ctx, const int64_t *rows, int64_t n, double *out);

/* Bench/test helper: append n rows of the integer-domain synthetic generator generated ON DEVICE
 * (spec: oracle/dot_scan.c orc_synth_row_f32; SURVEY.md 8d allows on-device generation for the 1M DB).
 * plant_* (may be NULL) list planted rows as GLOBAL row indices inside the appended range, sorted by dst:
 * kind 1 = noisy copy of src (cos ~ 0.98), kind 2 = exact duplicate of src.                            */
int chip_db_append_synthetic(chip_ctx *ctx, int64_t n, uint64_t seed,
                             const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant);
/* Same rows normalised to UNIT L2 norm (ABI 7; SURVEY.md 8d: "rows = unit-L2-norm", what NetVLAD's last layer emits): the row's integers
 * v_e, S = sum v_e^2 exactly in 64-bit integers, element = (float)((double)v_e * (1 / sqrt((double)S))) -- every floating-point step one
 * correctly rounded operation, so the device rows equal oracle/dot_scan.c orc_synth_row_unit_f32 bit for bit.  Planted rows: kind 1 is the
 * unit vector along 5 src + own (cos ~ 0.98 with src), kind 2 the unit vector of src.  bench.py's headline database is made by this call. */
int chip_db_append_synthetic_unit(chip_ctx *ctx, int64_t n, uint64_t seed,
                                  const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant);

/* ------------------------------------------------------------------------------------------ scan + top-k
 * Replaces  u = v.transpose() * M.leftCols(k); maxCoeff(); last-index argmax  (src/Cerebro.cpp:1026-1043)
 * generalised to top-K (the compiled-out faiss variants use K=5, src/Cerebro.cpp:460).
 * Scores are fp64 accumulations of exact fp32 products in the fixed order of DESIGN.md 3; ordering is
 * (score descending, index DESCENDING) so K=1 is the reference's "last index attaining the max".
 * Unused slots (k < K): score = -inf, idx = -1.  scores/idx are nq*topk, query-major.                   */
int chip_query_rows(chip_ctx *ctx, int64_t k, const int64_t *query_rows, int32_t nq, int32_t topk,
                    double *scores, int64_t *idx);
int chip_query_vectors_f32(chip_ctx *ctx, int64_t k, const float *queries, int32_t nq, int32_t topk,
                           double *scores, int64_t *idx);
/* double query vectors: as they are on a double-row DB; on a float-row DB they must be float32-representable (else
 * CHIP_ERR_NOT_F32 -- a rounded query would silently change scores). */
int chip_query_vectors_f64(chip_ctx *ctx, int64_t k, const double *queries, int32_t nq, int32_t topk,
                           double *scores, int64_t *idx);
/* The whole score vector  u = v^T * M.leftCols(k)  of ONE query row (src/Cerebro.cpp:1026; the reference's debug plot consumes
 * all of u, :1047-1052) -- same arithmetic, same bits as the scores chip_query_rows selects from.  u: k doubles (host).
 * A chip_create_multi ctx fills all of u.  A sharded ctx of the one-process-per-GPU layout fills only the entries of the rows THIS
 * rank owns (u[i], i % shard_count == shard_rank), with or without an exchange; with an exchange attached the call is collective
 * (the query row is broadcast from its owner), without one the query row must still be in this rank's ring. */
int chip_query_scores(chip_ctx *ctx, int64_t k, int64_t query_row, double *u);

/* Many-query batched mode: Q query descriptors (host, Q x D fp32) against rows [0,k) in ONE pass of
 * the DB as an fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32) with a fused exact top-k.  Semantics are those of
 * the reference's compiled-out faiss variants (IndexFlatIP on float descriptors, src/Cerebro.cpp:390,422,455-472):
 * score = fp32 inner product, here defined as ONE k-ordered fmaf chain (bit-reproducible; differs from the fp64 scores
 * of chip_query_* by fp32 round-off, ~1e-7).  Ordering (score desc, index desc); unused slots -inf / -1.
 * Requires D % 32 == 0 and float rows.  Worth it from Q ~ 40 upwards (arithmetic intensity Q/2 flop/B vs the 19.7 flop/B ridge).
 * Sharded DBs: a chip_create_multi ctx runs one pass per device over the rows it owns and merges the per-device lists on
 * devices[0]; a sharded ctx with an attached communicator does the same collectively (ncclAllGather of Q x topk entries per rank,
 * the same result on every rank); a sharded ctx WITHOUT an exchange answers for its own rows only (global indices) -- the host
 * merges.  Results are those of one device holding the whole DB, bit for bit (exact selection under a total order).            */
int chip_query_batch_f32(chip_ctx *ctx, int64_t k, const float *queries, int32_t Q, int32_t topk,
                         float *scores /* Q x topk */, int64_t *idx /* Q x topk */);

/* ------------------------------------------------------------------------------------------ the tick
 * One pass of the while-loop body of Cerebro::descrip_N__dot__descrip_0_N (src/Cerebro.cpp:956-1100) for
 * l = wholeImageComputedList_size().  Defaults (chip_dot_params_default): LOCALITY_THRESH 12 (:912),
 * DOT_PROD_THRESH (double)(float)0.85 (:913,:1056), lag 50 (:914,:1019), >=3 new rows (:962), k > 5 (:1022). */
typedef struct {
    int32_t locality;
    int32_t lag;
    int32_t min_new;
    int32_t min_k;
    double  thresh;
} chip_dot_params;
void chip_dot_params_default(chip_dot_params *p);

enum { CHIP_TICK_SKIPPED = 0,   /* l - last_l < min_new: nothing done, last_l NOT advanced (:962-966)     */
       CHIP_TICK_TOO_SHORT = 1, /* ran, but k = l - lag <= min_k (:1022 else-branch); last_l = l          */
       CHIP_TICK_SCANNED = 2,   /* scan + decision executed; last_l = l                                   */
       CHIP_TICK_FAILED = 3 };  /* sharded ticks only, never returned with CHIP_OK: a shard could not take part; the collecting
                                   call returns CHIP_ERR_SHARD_FAILED on every rank and last_l is as before the tick -- unless a
                                   LATER tick had been enqueued by then (pipelined form): that tick's last_l = l stands      */

typedef struct {
    int32_t status;      /* CHIP_TICK_*                                                                   */
    int32_t found;       /* 1 iff the :1056 criterion fired -> foundLoops.push_back (:1078-1081)          */
    int64_t idx_curr;    /* l-1                (wholeImageComputedList index of t_curr)                   */
    int64_t idx_prev;    /* u_argmax           (index of t_prev)                                          */
    double  score;       /* u_max                                                                         */
    int64_t argmax[3];   /* u_argmax, um_argmax, umm_argmax                                               */
    double  maxv[3];     /* u_max, um_max, umm_max                                                        */
} chip_tick_result;

/* Synchronous tick (single-GPU ctx, shard_count == 1).
 * Environment, read at chip_create: CHIP_TICK_RESIDENT=1 turns ticks over prefixes of up to 512 MiB into commands to a scan kernel
 * that stays on the chip between ticks (no launch per tick: ~4.7 us less per call back to back, ~14 us at the reference's 10 Hz
 * cadence; same results).  The instance leaves by itself once NO tick has arrived for a whole lease (CHIP_RESIDENT_LEASE_MS,
 * default 250) -- so at 10 Hz it never leaves by itself.  The library itself pauses it around everything of its own that frees /
 * allocates device memory, grows the DB by a segment or wants the whole chip (no instance is launched until that section is over;
 * ticks that arrive meanwhile are launched).  Calls of OTHER libraries in the same process that wait for the whole device
 * (hipFree, hipMalloc of a new pool block, hipDeviceSynchronize) would wait for as long as ticks keep coming: bracket them with
 * chip_resident_pause / chip_resident_resume, or leave the mode off in such a process.  Off by default. */
int chip_loop_tick(chip_ctx *ctx, int64_t l, const chip_dot_params *p, chip_tick_result *out);
/* Retire the resident scan instance (waits until it has left the chip, <= one tick) and keep the mode from launching another one
 * until the matching chip_resident_resume; ticks in between are ordinary launches with identical results.  Calls nest.  CHIP_OK and
 * no effect on a ctx that does not run the mode.  Thread-safe against the tick, append and PnP threads (ABI 6). */
int chip_resident_pause(chip_ctx *ctx);
int chip_resident_resume(chip_ctx *ctx);
/* Pipelined form: enqueue up to CHIP_MAX_INFLIGHT - 1 ticks without host synchronisation, collect later.  Scans run
 * back to back on an internal stream; the one-workgroup merge of tick i (ctx stream) overlaps the scan of tick i+1. */
#define CHIP_MAX_INFLIGHT 64
int chip_loop_tick_enqueue(chip_ctx *ctx, int64_t l, const chip_dot_params *p, int32_t slot);
int chip_loop_tick_collect(chip_ctx *ctx, int32_t slot, chip_tick_result *out);
int64_t chip_loop_last_l(const chip_ctx *ctx);
void chip_loop_reset(chip_ctx *ctx);

/* Sharded tick, three phases (host does the exchange between 1 and 2):
 *  1. chip_scan_local: scan this rank's share of rows [0,k), k = l - lag, for the three queries l-1,l-2,l-3 and
 *     leave its 3 x topk list (chip_topk_entry, global indices) in DEVICE memory at dev_out (caller-owned,
 *     3*topk*sizeof(chip_topk_entry) bytes).  *status gets CHIP_TICK_*; when it is not CHIP_TICK_SCANNED nothing was
 *     enqueued and phases 2-3 are skipped by every rank alike.
 *     Stream semantics: the list is written by a small merge kernel on the ctx stream (chip_set_stream), so work
 *     enqueued there afterwards (the all-gather) sees it and dev_out / gathered buffers may be reused every tick.
 *     The scan itself runs on an internal stream: the next tick's scan overlaps this tick's merge + all-gather.
 *  2. host: all-gather dev_out of every rank -> gathered[G][3][topk] (on the ctx stream).
 *  3. chip_merge_decide (synchronous) or chip_merge_decide_enqueue + chip_loop_tick_collect (pipelined):
 *     merge the G lists per query, apply the :1056 criterion, return the result.                          */
typedef struct { double score; int64_t idx; } chip_topk_entry;
int chip_scan_local(chip_ctx *ctx, int64_t l, const chip_dot_params *p, int32_t topk, void *dev_out, int32_t *status);
int chip_merge_decide(chip_ctx *ctx, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists,
                      int32_t topk, chip_tick_result *out);
int chip_merge_decide_enqueue(chip_ctx *ctx, int64_t l, const chip_dot_params *p, const void *dev_gathered, int32_t n_lists,
                              int32_t topk, int32_t slot);

/* ------------------------------------------------------------------------------------------ PnP-RANSAC
 * Replaces the body of StaticTheiaPoseCompute::PNP (src/DlsPnpWithRansac.cpp:192-240): theia::Ransac over
 * the DlsPnpWithRansac estimator (src/DlsPnpWithRansac.h:42-100).
 *   X  : N x 3 row-major, 3-D points in frame a            (w_X,  DlsPnpWithRansac.cpp:196)
 *   uv : N x 2 row-major, normalized image coords in b     (c_uv_normalized, :197)
 *   T  : 4x4 COLUMN-major b_T_a (Eigen Matrix4d layout)    (c_T_w = best_rel_pose.b_T_a, :239)
 * Returns CHIP_ERR_TOO_FEW_POINTS for N < 20 (:136-139; the reference returns confidence -1).  When no
 * hypothesis yields a model, status is CHIP_OK, *confidence = 0 and T is filled with NaN (the reference
 * returns an UNINITIALISED Matrix4d, :204; its caller only NaN-checks, Cerebro.cpp:1678).
 * n_hypotheses == 0: reference-faithful adaptive loop (<= max_iterations, early termination as
 * theia::Ransac; hypotheses are generated in parallel and the sequential rule is replayed on the host).
 * n_hypotheses  > 0: benchmark mode, exactly that many hypotheses, all scored, argmin cost with the lowest
 * hypothesis index winning ties (BASELINE config 3 uses 1000).                                         */
typedef struct {
    double  error_thresh;        /* 0.03   DlsPnpWithRansac.cpp:208 */
    double  min_inlier_ratio;    /* 0.7    :209 */
    int32_t max_iterations;      /* 50     :210 */
    int32_t min_iterations;      /* 5      :211 */
    int32_t use_mle;             /* 1      :212 */
    int32_t sample_size;         /* 15     DlsPnpWithRansac.h:45 */
    double  failure_probability; /* 0.01   theia::RansacParameters default */
    uint64_t seed;               /* counter-based sampler seed (Theia's is time-seeded => nondeterministic) */
    int32_t n_hypotheses;        /* 0 = adaptive reference mode */
    int32_t sampler;             /* CHIP_SAMPLER_*: which permutation the 15 / 10 sample indices of a hypothesis are drawn from (ABI 5) */
} chip_ransac_params;
/* CHIP_SAMPLER_FRESH (default): a fresh identity permutation per hypothesis -- hypotheses are independent and are generated on the
 * device.  CHIP_SAMPLER_THEIA_PERSISTENT: theia::RandomSampler as written -- the permutation is initialised ONCE per estimation and
 * every hypothesis continues on the array the previous one left (one theia::Ransac, hence one sampler, per PNP / P3P_ICP call:
 * src/DlsPnpWithRansac.cpp:216-221, :95-100); the host sequences the swaps (S per hypothesis) and hands the kernels a sample table.
 * Both modes use the same counter-based draws (Theia's own generator is time-seeded, i.e. not reproducible).                     */
enum { CHIP_SAMPLER_FRESH = 0, CHIP_SAMPLER_THEIA_PERSISTENT = 1 };
void chip_ransac_params_default(chip_ransac_params *p);

typedef struct {
    int32_t n_iterations;     /* summary.num_iterations (:229)                                   */
    int32_t n_inliers;
    int32_t best_hypothesis;  /* index of the winning hypothesis, -1 if none                      */
    int32_t n_models;         /* hypotheses for which DlsPnp returned exactly one solution        */
    double  best_cost;
} chip_ransac_summary;

int chip_pnp_ransac(chip_ctx *ctx, const double *X, const double *uv, int32_t N, const chip_ransac_params *p,
                    double T_colmajor[16], float *confidence, uint8_t *inlier_mask /* N bytes, may be NULL */,
                    chip_ransac_summary *summary /* may be NULL */);

/* P independent estimations in one pair of launches -- e.g. the two role-swapped PNP calls the loop-candidate consumer
 * makes per image pair (src/Cerebro.cpp:1518 and :1572).  One wave per hypothesis is latency-bound at H = 1000, so
 * co-scheduled problems cost little more than one.  All problems share *p; problem i draws from seeds[i] (NULL: p->seed
 * for every problem) and its outputs (T_colmajor + 16 i, confidence[i], inlier_mask[i] (N[i] bytes; the array or any
 * entry may be NULL), summary[i]) are bit-identical to chip_pnp_ransac(X[i], uv[i], N[i]) with that seed.
 * Any N[i] < 20 fails the whole call with CHIP_ERR_TOO_FEW_POINTS before anything runs.                         */
int chip_pnp_ransac_batch(chip_ctx *ctx, int32_t P, const double *const *X, const double *const *uv, const int32_t *N,
                          const chip_ransac_params *p, const uint64_t *seeds, double *T_colmajor /* P x 16 */,
                          float *confidence /* P */, uint8_t *const *inlier_mask, chip_ransac_summary *summary /* P or NULL */);

/* ------------------------------------------------------------------------------------------ Umeyama-ICP-RANSAC
 * Replaces the RANSAC branch of StaticTheiaPoseCompute::P3P_ICP (src/DlsPnpWithRansac.cpp:65-121): theia::Ransac over
 * AlignPointCloudsUmeyamaWithRansac (src/DlsPnpWithRansac.h:104-166): 10-point sample -> AlignPointCloudsUmeyama ->
 * accept iff min(s, 1/s) > 0.9 (:137) -> b_T_a = [R t] -> L2 error (:152-164), threshold 0.1, MLE score.
 *   A, B : N x 3 row-major, the same 3-D points expressed in frames a and b (uv_X, uvd_Y, :15-16)
 * Status / T / confidence / mask / summary conventions are those of chip_pnp_ransac (N < 20 -> CHIP_ERR_TOO_FEW_POINTS,
 * :19-22).  chip_icp_params_default = chip_ransac_params_default with error_thresh 0.1 (:89) and sample_size 10 (.h:118). */
void chip_icp_params_default(chip_ransac_params *p);
int chip_icp_ransac(chip_ctx *ctx, const double *A, const double *B, int32_t N, const chip_ransac_params *p,
                    double T_colmajor[16], float *confidence, uint8_t *inlier_mask /* N bytes, may be NULL */,
                    chip_ransac_summary *summary /* may be NULL */);
/* The same estimation in two halves, so that it can run underneath something else -- the loop-candidate consumer computes
 * PNP(a->b), PNP(b->a) and P3P_ICP for one image pair (src/Cerebro.cpp:1518,1572,1629), and the ICP kernel is tiny:
 *   chip_icp_ransac_enqueue : copies A, B (the call returns once they are staged) and launches on the ctx's ICP stream;
 *   chip_icp_ransac_collect : waits for it and delivers exactly what chip_icp_ransac would have returned.
 * One estimation may be pending per ctx (a second enqueue, or a collect without enqueue: CHIP_ERR_BUSY).                 */
int chip_icp_ransac_enqueue(chip_ctx *ctx, const double *A, const double *B, int32_t N, const chip_ransac_params *p);
int chip_icp_ransac_collect(chip_ctx *ctx, double T_colmajor[16], float *confidence, uint8_t *inlier_mask /* may be NULL */,
                            chip_ransac_summary *summary /* may be NULL */);


/* ------------------------------------------------------------------------------------------ introspection */
typedef struct {
    int32_t abi_version;
    int32_t D;
    int32_t device;
    int32_t shard_rank, shard_count;
    int32_t n_cus;              /* multiProcessorCount                                             */
    int64_t rows_global;        /* == chip_db_size                                                 */
    int64_t rows_local;         /* rows stored on this rank                                        */
    int64_t capacity_local;     /* rows reserved on this rank                                      */
    int64_t lossy_rows;         /* rows appended with CHIP_APPEND_ALLOW_ROUNDING that actually rounded */
    char    arch[32];           /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"                      */
    int32_t storage_bytes;      /* 4 = float rows, 8 = double rows                                 */
    int32_t n_devices;          /* 1, or G of chip_create_multi (then the other fields describe devices[0]) */
    int32_t exchange;           /* CHIP_EXCHANGE_*                                                 */
    int32_t comm_ranks;         /* ranks of the RCCL communicator the exchange runs over (ncclCommCount), 0 = none: a caller
                                   that asked for G GPUs can PROVE the collective spans G ranks (and see a copy fallback) */
    int32_t comm_init_abandoned;/* 1: an RCCL bootstrap of this ctx (ncclCommInitAll in chip_create_multi, ncclCommInitRank in
                                   chip_comm_init_rank) did not return within CHIP_COMM_INIT_TIMEOUT_MS (default 120 s) and was
                                   abandoned on its helper thread; the ctx works (a group: over the copy exchange), but process
                                   teardown may block inside RCCL -- leave through _exit() once the work is done (ABI 4)      */
    int32_t scan_forms;         /* CHIP_SCAN_FORM_* bits: which forms of the scan kernel this build of the library contains (ABI 5).
                                   The row-batched form (short prefixes) depends on how the building hipcc allocates registers; a
                                   build whose code-object check failed is made with -DCHIP_NO_ROWS_FORM and serves every scan with
                                   the one-row kernel -- same results, short prefixes slower                                    */
    int32_t test_hooks;         /* 0: the product build (`make lib`): no fault-injection hook, no test knob is compiled in, the
                                   CHIP_TEST_* / CHIP_PNP_BACKSUB / CHIP_PNP_DEBUG_STOP environment variables are never read.
                                   1: the test build (-DCHIP_TEST_HOOKS, `make testlibs` -> cerebro_amd/lib/hooks/), which only
                                   tests/ load -- never deploy it (ABI 6)                                                       */
} chip_info;
enum { CHIP_SCAN_FORM_ONE_ROW = 1, CHIP_SCAN_FORM_ROWS = 2 };
/* the same bits without a ctx (what `make verify` and the build log ask) */
int chip_build_scan_forms(void);
/* chip_info.test_hooks without a ctx */
int chip_build_test_hooks(void);
enum { CHIP_EXCHANGE_NONE = 0, CHIP_EXCHANGE_RCCL = 1, CHIP_EXCHANGE_COPY = 2 };
int chip_get_info(const chip_ctx *ctx, chip_info *info);

/* Per-kernel timing on the ctx stream (hipEvents bracketing every scan launch). */
int chip_profile_enable(chip_ctx *ctx, int32_t on);
int chip_profile_reset(chip_ctx *ctx);
/* Sum of the per-launch durations (ms) and launch count of the dominant kernel (db_scan_topk) since reset, plus the
 * busy span first-start -> last-stop; total_ms / span_ms (the
 * measured concurrency) stays ~1: scans are serialised on one internal stream. */
int chip_profile_scan(chip_ctx *ctx, double *total_ms, int64_t *n_launches, double *bytes_per_launch_last, double *span_ms);

#ifdef __cplusplus
}
#endif
#endif
