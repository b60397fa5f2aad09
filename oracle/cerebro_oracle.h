/*
 * cerebro_oracle.h -- CPU restatement (plain C) of cerebro's loop-detection hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libcerebro_hip.so) never
 * links, loads or calls anything in this directory.
 *
 * Parity status
 *   - dot-product tick (Cerebro.cpp:903-1103): every branch below cites the reference line it
 *     restates.  The reference holds NO golden vectors/tests for this path and
 *     cannot be compiled here (ROS/Eigen/OpenCV absent), so the oracle is pinned by
 *     construction from the cited lines plus an independent numpy mirror (tests/np_mirror.py).
 *   - DLS-PnP / RANSAC (DlsPnpWithRansac.{h,cpp} + un-vendored Theia-SfM, version unpinned):
 *     **parity unpinned** for the Theia internals -- see pnp_ransac.c header.
 *
 * All functions are reentrant; no global state.
 */
#ifndef CEREBRO_ORACLE_H
#define CEREBRO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ synthetic descriptors */
/* Integer-domain generator shared (as a SPEC, not as code) with the device generator in
 * cerebro_amd/csrc/synth.hip: bit-identical on any IEEE machine because the only floating
 * point operations are one exact int->float conversion and one float multiply.            */
uint64_t orc_splitmix64(uint64_t x);
int32_t  orc_synth_i32(uint64_t seed, int64_t row, int32_t e);      /* Irwin-Hall(4) of 16-bit fields, centred */
float    orc_synth_scale(int32_t D);                                 /* (float)(1/sqrt(D*var))                  */
float    orc_synth_scale_planted(int32_t D);                         /* (float)(1/sqrt(D*var*26))               */
/* kind: 0 = plain row, 1 = noisy copy of src (5*x_src + x_noise), 2 = exact duplicate of src */
void     orc_synth_row_f32(uint64_t seed, int64_t row, int32_t D, int32_t kind, int64_t src, float *out);
/* the same row normalised to unit L2 norm: S = sum v^2 in integers, element = (float)((double)v * (1 / sqrt((double)S))) */
void     orc_synth_row_unit_f32(uint64_t seed, int64_t row, int32_t D, int32_t kind, int64_t src, float *out);

/* ------------------------------------------------------------------ dot products */
/* Fixed summation tree of SURVEY.md Appendix B (the device kernel uses the same tree):
 * lane L (0..63) accumulates elements j*256+4L+c, j ascending, c=0..3; then
 * acc[L] += acc[L^m] for m=32,16,8,4,2,1.  Products of fp32 values are exact in fp64.   */
double   orc_dot_tree_f32(const float *q, const float *row, int32_t D);
/* Double rows: lane L sums elements j*128 + 2L + c by fused multiply-add (one rounding per term), then the same butterfly. */
double   orc_dot_tree_f64(const double *q, const double *row, int32_t D);
void     orc_scan_topk_f64(const double *db, int64_t k, int32_t D, const double *queries, int32_t nq, int32_t K,
                           double *out_scores, int64_t *out_idx, int32_t nthreads);
/* all scores of one query over rows [0,k); elem = 4 (float rows / float query) or 8 */
void     orc_scores(const void *db, int32_t elem, int64_t k, int32_t D, const void *query, double *u, int32_t nthreads);
/* Plain left-to-right fp64 dot (the most literal reading of v^T * M.col(i), Cerebro.cpp:1026). */
double   orc_dot_seq_f64(const double *q, const double *col, int32_t D);

/* Top-K of scores of nq queries against rows [0,k) of a row-major fp32 DB.
 * Order: score descending, then index DESCENDING (K=1 == "last index attaining the max",
 * Cerebro.cpp:1038-1043).  Empty slots: score=-inf, idx=-1.  out arrays are nq*K.        */
void     orc_scan_topk_f32(const float *db, int64_t k, int32_t D,
                           const float *queries, int32_t nq, int32_t K,
                           double *out_scores, int64_t *out_idx);
/* Batched many-query mode (row N4): fp32 k-ordered fmaf chain per (query, row); scores returned as doubles
 * (exactly the float values); same ordering rule.  OpenMP over queries. */
float    orc_dot_fmaf_f32(const float *q, const float *row, int32_t D);
void     orc_scan_topk_fmaf_f32(const float *db, int64_t k, int32_t D, const float *queries, int32_t nq, int32_t K,
                                double *out_scores, int64_t *out_idx);
/* Same, rows produced on the fly by the synthetic generator (for 100k/1M checks without 16 GB of RAM).
 * plant_dst/src/kind describe planted rows (sorted by dst, may be NULL).  nthreads<=0 -> 1. */
void     orc_scan_topk_synth(uint64_t seed, int64_t k, int32_t D,
                             const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant,
                             const float *queries, int32_t nq, int32_t K,
                             double *out_scores, int64_t *out_idx, int32_t nthreads);
void     orc_scan_topk_synth_unit(uint64_t seed, int64_t k, int32_t D,   /* ... over orc_synth_row_unit_f32 rows */
                                  const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant,
                                  const float *queries, int32_t nq, int32_t K,
                                  double *out_scores, int64_t *out_idx, int32_t nthreads);

/* ------------------------------------------------------------------ the tick (Cerebro.cpp:956-1100) */
typedef struct {
    int32_t locality;   /* LOCALITY_THRESH = 12            Cerebro.cpp:912 */
    double  thresh;     /* (double)(float)0.85             Cerebro.cpp:913,1056 */
    int32_t lag;        /* start_adding_..._after = 50     Cerebro.cpp:914,1019 */
    int32_t min_new;    /* l - last_l < 3 -> nothing       Cerebro.cpp:962 */
    int32_t min_k;      /* if( k > 5 )                     Cerebro.cpp:1022 */
} orc_dot_params;
void orc_dot_params_default(orc_dot_params *p);

typedef struct {
    int32_t status;          /* 0 = skipped (<min_new new rows; last_l NOT advanced), 1 = ran, k<=min_k, 2 = scanned */
    int32_t found;           /* 1 iff a loop candidate was accepted */
    int64_t idx_curr;        /* l-1 */
    int64_t idx_prev;        /* u_argmax */
    double  score;           /* u_max */
    int64_t argmax[3];       /* u, um, umm argmax (or -1) */
    double  maxv[3];
} orc_tick_result;

typedef struct { int64_t last_l; } orc_loop_state;

/* One pass of the while-loop body on a row-major fp32 DB holding >= l rows. */
void orc_loop_tick_f32(orc_loop_state *st, const orc_dot_params *p, const float *db, int32_t D,
                       int64_t l, orc_tick_result *out);
void orc_loop_tick_f64(orc_loop_state *st, const orc_dot_params *p, const double *db, int32_t D,
                       int64_t l, orc_tick_result *out);

/* Reference-faithful CPU path used as bench.py's cpu_baseline (kind "port"):
 * fp64 COLUMN-major M (D x cap) exactly as Cerebro.cpp:946, three SEPARATE GEMVs (:1026-1028),
 * three maxCoeff (:1035-1037), one last-index argmax loop (:1038-1043).  Single thread,
 * like the reference's dot_product_th.  u/um/umm are caller-provided scratch of length k. */
void orc_ref_scan_f64_colmajor(const double *M, int32_t D, int64_t k,
                               const double *v, const double *vm, const double *vmm,
                               double *u, double *um, double *umm,
                               double maxv[3], int64_t argmax[3]);
/* OpenMP-over-columns variant of the same statements; bit-identical results. */
void orc_ref_scan_f64_colmajor_omp(const double *M, int32_t D, int64_t k,
                                   const double *v, const double *vm, const double *vmm,
                                   double *u, double *um, double *umm,
                                   double maxv[3], int64_t argmax[3], int32_t nthreads);
void orc_tile_columns_omp(double *M, int32_t D, int64_t k, const double *src, int64_t src_cols, int32_t nthreads);

/* Eigen-order emulation of  v.transpose() * M.leftCols(k)  (Cerebro.cpp:1026-1028): Eigen 3.3.x row-major GEMV, one packet
 * accumulator per output + predux + scalar tail; packet 2 / fma 0 = the reference's x86-64 SSE2 Release build, packet 4 / fma 1 =
 * an AVX2+FMA build, packet 1 = the sequential chain of orc_dot_seq_f64.  Parity evidence (tests/test_oracle_eigen_order.py):
 * the selection of this path vs the oracle's fixed-tree order. */
double orc_dot_eigen_gemv_f64(const double *v, const double *col, int32_t D, int32_t packet, int32_t fma_flag, int32_t aligned_start);
void orc_ref_scan_f64_eigen_order(const double *M, int32_t D, int64_t k, const double *v, const double *vm, const double *vmm,
                                  double *u, double *um, double *umm, double maxv[3], int64_t argmax[3],
                                  int32_t packet, int32_t fma_flag, int32_t nthreads);
/* The same order in the SHAPE Eigen runs it (four rows at a time, one SSE2 Packet2d accumulator each, three SEPARATE GEMVs):
 * bit-identical to orc_dot_eigen_gemv_f64(.., packet 2, fma 0, aligned_start 0) per entry; bench.py's headline cpu_baseline. */
void orc_ref_scan_f64_eigen_gemv3(const double *M, int32_t D, int64_t k, const double *v, const double *vm, const double *vmm,
                                  double *u, double *um, double *umm, double maxv[3], int64_t argmax[3], int32_t nthreads);


/* ------------------------------------------------------------------ EuRoC-shaped surrogate run (surrogate.c) */
/* one step of a unit-norm AR(1) descriptor walk in plain sequential C (bit-reproducible on any machine) */
void orc_ar1_step(const double *prev, const float *noise, double alpha, int32_t D, int32_t round_f32, double *out);
/* One tick in a chosen summation order (0 = the device's fixed tree, 1 = Eigen 3.3 SSE2 GEMV order; order 1 needs elem == 8),
 * OpenMP over rows; gap[q] (may be NULL) = best - second-best score of query q (0 for an exact tie). */
void orc_loop_tick_order(orc_loop_state *st, const orc_dot_params *p, const void *db, int32_t elem, int32_t D, int64_t l,
                         int32_t order, int32_t nthreads, orc_tick_result *out, double gap[3]);

/* ================================================================== PnP / RANSAC (pnp_ransac.c) */
typedef struct {
    double  error_thresh;        /* 0.03   DlsPnpWithRansac.cpp:208 */
    double  min_inlier_ratio;    /* 0.7    :209 */
    int32_t max_iterations;      /* 50     :210 */
    int32_t min_iterations;      /* 5      :211 */
    int32_t use_mle;             /* 1      :212 */
    int32_t sample_size;         /* 15     DlsPnpWithRansac.h:45 */
    double  failure_probability; /* 0.01   theia default */
    uint64_t seed;
    int32_t n_hypotheses;        /* 0 = adaptive reference mode, >0 = fixed count, all scored */
    int32_t sampler;             /* 0 = fresh identity permutation per hypothesis (hypotheses independent: the default);
                                    1 = theia::RandomSampler as written: ONE permutation, initialised once per estimation and carried
                                    from hypothesis to hypothesis (DlsPnpWithRansac.cpp:216-221 constructs one Ransac, hence one sampler) */
} orc_ransac_params;
typedef struct {
    int32_t n_iterations, n_inliers, best_hypothesis, n_models;
    double  best_cost;
} orc_ransac_summary;

uint64_t orc_rng_draw(uint64_t seed, uint32_t hyp, uint32_t draw);
void   orc_ransac_sample(uint64_t seed, int32_t hyp, int32_t N, int32_t S, int32_t *out);
/* theia::RandomSampler with its PERSISTENT permutation (Initialize once: 0..N-1; every Sample(): for i < S swap(idx[i],
 * idx[RandInt(i, N-1)]), subset[i] = idx[i]), over the same counter-based draws: samples of hypotheses 0..H-1 -> out[H][S]. */
void   orc_ransac_sample_persistent(uint64_t seed, int32_t H, int32_t N, int32_t S, int32_t *out);
/* the hypothesis of a GIVEN sample (sample == NULL: orc_ransac_sample(seed, hyp, ...)); the linear form is keyed by (seed, hyp) */
int    orc_pnp_hypothesis_sampled(const double *X, const double *uv, int32_t N, uint64_t seed, int32_t hyp, int32_t S,
                                  const int32_t *sample, double T[16]);
int    orc_icp_hypothesis_sampled(const double *A, const double *B, int32_t N, uint64_t seed, int32_t hyp, int32_t S,
                                  const int32_t *sample, double T[16], double *scale_out);
void   orc_dls_linear_form(uint64_t seed, int32_t hyp, double u[4]);
double orc_reproj_error(const double *T_colmajor, const double *X, const double *uv);
void   orc_score_model(const double *T_colmajor, const double *X, const double *uv, int32_t N, double thresh, int32_t use_mle,
                       double *cost, int32_t *n_inliers, uint8_t *mask);
void   orc_dls_monomial_positions(int32_t pos[8][8][8]);
void   orc_dls_cubics(const double *X, const double *uv, int32_t n, double Tfac[27], double f[3][20]);
int    orc_dls_action_matrix(const double f[3][20], const double u[4], double S[27 * 27]);
int    orc_eig27_real(const double S[27 * 27], double lambda[27], double v4[27][4]);
int    orc_dls_pnp(const double *X, const double *uv, int32_t n, const double u[4], double *Rs, double *ts, int32_t max_out);
int    orc_pnp_hypothesis(const double *X, const double *uv, int32_t N, uint64_t seed, int32_t hyp, int32_t S,
                          double T[16], int32_t *sample_out);
void   orc_ransac_params_default(orc_ransac_params *p);
int32_t orc_ransac_max_iterations(int32_t S, double ratio, double log_fail, int32_t min_it, int32_t max_it);
int    orc_pnp_ransac(const double *X, const double *uv, int32_t N, const orc_ransac_params *p,
                      double T[16], float *confidence, uint8_t *mask, orc_ransac_summary *summary);

/* Benchmark-mode hypotheses on `nthreads` threads (OpenMP over hypotheses); returns the winning hypothesis index. */
int32_t orc_pnp_hypotheses_mt(const double *X, const double *uv, int32_t N, const orc_ransac_params *p, int32_t H, int32_t nthreads,
                              int32_t *n_models_out);

/* ================================================================== Umeyama-ICP / RANSAC (icp_ransac.c) */
int    orc_umeyama(const double *a, const double *b, int32_t n, double R[9], double t[3], double *scale);
double orc_icp_error(const double *T_colmajor, const double *a, const double *b);
void   orc_icp_score_model(const double *T_colmajor, const double *A, const double *B, int32_t N, double thresh, int32_t use_mle,
                           double *cost, int32_t *n_inliers, uint8_t *mask);
int    orc_icp_hypothesis(const double *A, const double *B, int32_t N, uint64_t seed, int32_t hyp, int32_t S, double T[16], double *scale_out);
void   orc_icp_params_default(orc_ransac_params *p);
int    orc_icp_ransac(const double *A, const double *B, int32_t N, const orc_ransac_params *p,
                      double T[16], float *confidence, uint8_t *mask, orc_ransac_summary *summary);

/* -DORC_FLOP_COUNT builds only (oracle/_build/liboracle_flops.so): fp64 operations executed by the PnP solver on the calling thread
 * since the last reset, per stage: 0 cost matrix + cubics, 1 Macaulay elimination (zero multipliers skipped, as the algorithm
 * runs), 2 back-substitution + action matrix, 3 Hessenberg reduction + accumulation, 4 Francis QR, 5 real eigenvectors +
 * back-transform, 6 pose + cheirality, 7 reprojection scoring.  *dense_lu: stage 1 without the zero-multiplier skip. */
#define ORC_FLOP_STAGES 8
void orc_flop_counts(long out[ORC_FLOP_STAGES], long *dense_lu, int reset);

/* ================================================================== top-k candidate policies (policies.c, row N4) */
typedef struct { int64_t idx_curr, idx_prev; double score; } orc_policy_loop;
typedef struct { int64_t last_l, l_last_added; } orc_naive_state;                      /* zero-initialise */
#define ORC_CLIQUE_MAX_RETAINED 64
typedef struct { int64_t last_l, l_last_added; int32_t n_retained; int32_t pad_;
                 int64_t key[ORC_CLIQUE_MAX_RETAINED]; int32_t cnt[ORC_CLIQUE_MAX_RETAINED]; } orc_clique_state;   /* zero-initialise */
/* One iteration of Cerebro::faiss__naive_loopcandidate_generator (Cerebro.cpp:400-489); returns 0/1 loops written. */
int32_t orc_faiss_naive_tick(const float *db, int32_t D, int64_t l, orc_naive_state *st, orc_policy_loop *out);
/* One iteration of Cerebro::faiss_clique_loopcandidate_generator (Cerebro.cpp:541-711); rnd replaces rand() (:692).
 * Returns the number of loops produced (written up to max_out). */
int32_t orc_faiss_clique_tick(const float *db, int32_t D, int64_t l, orc_clique_state *st,
                              int (*rnd)(void *), void *rnd_arg, orc_policy_loop *out, int32_t max_out);

#ifdef __cplusplus
}
#endif
#endif
