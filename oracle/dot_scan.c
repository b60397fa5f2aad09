/*
 * dot_scan.c -- CPU oracle for cerebro's whole-image-descriptor dot-product scan + candidate select.
 *
 * TEST INFRASTRUCTURE ONLY (see cerebro_oracle.h).  Restates, line by line,
 *   /root/reference/src/Cerebro.cpp:903-1103  (Cerebro::descrip_N__dot__descrip_0_N)
 * Build: gcc -O2 -ffp-contract=off -fopenmp  (no -ffast-math: summation order is part of the spec).
 */
#include "cerebro_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- synthetic generator (spec) */
uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint64_t synth_rowkey(uint64_t seed, int64_t row)
{
    return orc_splitmix64(seed + 0x632BE59BD9B4E019ULL * (uint64_t)row);
}

static inline int32_t synth_from_key(uint64_t rowkey, int32_t e)
{
    uint64_t h = orc_splitmix64(rowkey + (uint64_t)(uint32_t)e);
    int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return s - 131070; /* centred Irwin-Hall(4) over 16-bit uniforms: |x| <= 131070 */
}

int32_t orc_synth_i32(uint64_t seed, int64_t row, int32_t e) { return synth_from_key(synth_rowkey(seed, row), e); }

#define SYNTH_VAR 1431655765.0 /* 4 * (65536^2 - 1) / 12 */
float orc_synth_scale(int32_t D) { return (float)(1.0 / sqrt((double)D * SYNTH_VAR)); }
float orc_synth_scale_planted(int32_t D) { return (float)(1.0 / sqrt((double)D * SYNTH_VAR * 26.0)); }

void orc_synth_row_f32(uint64_t seed, int64_t row, int32_t D, int32_t kind, int64_t src, float *out)
{
    const float c = orc_synth_scale(D), cp = orc_synth_scale_planted(D);
    const uint64_t key = synth_rowkey(seed, row);
    if (kind == 0) {
        for (int32_t e = 0; e < D; e++) out[e] = (float)synth_from_key(key, e) * c;
    } else {
        const uint64_t skey = synth_rowkey(seed, src);
        if (kind == 2)
            for (int32_t e = 0; e < D; e++) out[e] = (float)synth_from_key(skey, e) * c;
        else /* noisy copy: cos(angle) ~ 5/sqrt(26) = 0.98 */
            for (int32_t e = 0; e < D; e++) out[e] = (float)(5 * synth_from_key(skey, e) + synth_from_key(key, e)) * cp;
    }
}

/* Unit-L2 form ("rows = unit-L2-norm", the NetVLAD layer's own normalisation -- scripts/predict_utils.py:59-61 of the
 * reference): the same integers v_e, S = sum v_e^2 exactly (|v| <= 6 * 131070, D <= 8192: S < 2^53), inv = 1 / sqrt((double)S) -- two
 * correctly rounded operations -- and element = (float)((double)v_e * inv).  Any IEEE machine produces the same bits; the device
 * generator (kernels.hip synth_rows_unit) sums the squares in another order, which integers do not notice. */
void orc_synth_row_unit_f32(uint64_t seed, int64_t row, int32_t D, int32_t kind, int64_t src, float *out)
{
    const uint64_t key = synth_rowkey(seed, row);
    const uint64_t skey = kind ? synth_rowkey(seed, src) : 0;
    uint64_t S = 0;
    for (int32_t e = 0; e < D; e++) {
        const int64_t v = kind == 0 ? synth_from_key(key, e) : kind == 2 ? synth_from_key(skey, e) : 5 * (int64_t)synth_from_key(skey, e) + synth_from_key(key, e);
        S += (uint64_t)(v * v);
    }
    const double inv = S ? 1.0 / sqrt((double)S) : 0.0;
    for (int32_t e = 0; e < D; e++) {
        const int64_t v = kind == 0 ? synth_from_key(key, e) : kind == 2 ? synth_from_key(skey, e) : 5 * (int64_t)synth_from_key(skey, e) + synth_from_key(key, e);
        out[e] = (float)((double)v * inv);
    }
}

/* ---------------------------------------------------------------- dot products */
double orc_dot_tree_f32(const float *q, const float *row, int32_t D)
{
    double acc[64];
    for (int L = 0; L < 64; L++) acc[L] = 0.0;
    for (int32_t base = 0; base < D; base += 256)
        for (int L = 0; L < 64; L++)
            for (int c = 0; c < 4; c++) {
                int32_t e = base + 4 * L + c;
                if (e < D) acc[L] = acc[L] + (double)q[e] * (double)row[e]; /* product exact in fp64 */
            }
    for (int m = 32; m >= 1; m >>= 1) {
        double nxt[64];
        for (int L = 0; L < 64; L++) nxt[L] = acc[L] + acc[L ^ m];
        memcpy(acc, nxt, sizeof acc);
    }
    return acc[0];
}

/* Double-row DBs (genuinely float64 descriptors, e.g. ReljaNetVLAD's numpy WPCA output,
 * scripts/whole_image_desc_compute_server.py:148-149; the reference's M is MatrixXd, src/Cerebro.cpp:946).  Products of two
 * doubles are NOT exact, so the definition fixes BOTH the rounding of every term and the order: lane L accumulates elements
 * j*128 + 2L + c (j ascending, c = 0,1) by FUSED multiply-add -- acc = fma(q[e], row[e], acc), one rounding per term -- then
 * the same xor butterfly as the float path.  (The device reads 16 bytes per lane per wave load: 2 doubles.) */
double orc_dot_tree_f64(const double *q, const double *row, int32_t D)
{
    double acc[64];
    for (int L = 0; L < 64; L++) acc[L] = 0.0;
    for (int32_t base = 0; base < D; base += 128)
        for (int L = 0; L < 64; L++)
            for (int c = 0; c < 2; c++) {
                int32_t e = base + 2 * L + c;
                if (e < D) acc[L] = fma(q[e], row[e], acc[L]);
            }
    for (int m = 32; m >= 1; m >>= 1) {
        double nxt[64];
        for (int L = 0; L < 64; L++) nxt[L] = acc[L] + acc[L ^ m];
        memcpy(acc, nxt, sizeof acc);
    }
    return acc[0];
}

double orc_dot_seq_f64(const double *q, const double *col, int32_t D)
{
    double s = 0.0;
    for (int32_t e = 0; e < D; e++) s = s + q[e] * col[e];
    return s;
}

/* fp32 dot product as ONE k-ordered fmaf chain: acc = fmaf(q[e], row[e], acc), e ascending.  This is what the batched
 * many-query mode computes (v_mfma_f32_32x32x2_f32 is bit-for-bit such a chain), and what the compiled-out faiss variants
 * of the reference compute up to summation order (IndexFlatIP on X.cast<float>(), Cerebro.cpp:390,422,455). */
float orc_dot_fmaf_f32(const float *q, const float *row, int32_t D)
{
    float acc = 0.0f;
    for (int32_t e = 0; e < D; e++) acc = __builtin_fmaf(q[e], row[e], acc);
    return acc;
}

/* ---------------------------------------------------------------- top-K with (score desc, idx desc) */
static inline int key_gt(double s, int64_t i, double s2, int64_t i2) { return s > s2 || (s == s2 && i > i2); }

static void topk_init(double *sc, int64_t *ix, int32_t K)
{
    for (int32_t j = 0; j < K; j++) { sc[j] = -INFINITY; ix[j] = -1; }
}
static void topk_push(double *sc, int64_t *ix, int32_t K, double s, int64_t i)
{
    if (!(s == s)) return; /* NaN never enters (SURVEY App.B (6)) */
    if (!key_gt(s, i, sc[K - 1], ix[K - 1])) return;
    int32_t j = K - 1;
    while (j > 0 && key_gt(s, i, sc[j - 1], ix[j - 1])) { sc[j] = sc[j - 1]; ix[j] = ix[j - 1]; j--; }
    sc[j] = s; ix[j] = i;
}

void orc_scan_topk_f32(const float *db, int64_t k, int32_t D, const float *queries, int32_t nq, int32_t K,
                       double *out_scores, int64_t *out_idx)
{
    for (int32_t q = 0; q < nq; q++) topk_init(out_scores + (size_t)q * K, out_idx + (size_t)q * K, K);
    for (int64_t i = 0; i < k; i++)
        for (int32_t q = 0; q < nq; q++)
            topk_push(out_scores + (size_t)q * K, out_idx + (size_t)q * K, K,
                      orc_dot_tree_f32(queries + (size_t)q * D, db + (size_t)i * D, D), i);
}

void orc_scan_topk_fmaf_f32(const float *db, int64_t k, int32_t D, const float *queries, int32_t nq, int32_t K,
                            double *out_scores, int64_t *out_idx)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int32_t q = 0; q < nq; q++) {
        double *sc = out_scores + (size_t)q * K;
        int64_t *ix = out_idx + (size_t)q * K;
        topk_init(sc, ix, K);
        for (int64_t i = 0; i < k; i++) topk_push(sc, ix, K, (double)orc_dot_fmaf_f32(queries + (size_t)q * D, db + (size_t)i * D, D), i);
    }
}

static void scan_topk_synth_impl(uint64_t seed, int64_t k, int32_t D,
                                 const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant,
                                 const float *queries, int32_t nq, int32_t K,
                                 double *out_scores, int64_t *out_idx, int32_t nthreads, int unit);

void orc_scan_topk_synth(uint64_t seed, int64_t k, int32_t D,
                         const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant,
                         const float *queries, int32_t nq, int32_t K,
                         double *out_scores, int64_t *out_idx, int32_t nthreads)
{
    scan_topk_synth_impl(seed, k, D, plant_dst, plant_src, plant_kind, n_plant, queries, nq, K, out_scores, out_idx, nthreads, 0);
}

void orc_scan_topk_synth_unit(uint64_t seed, int64_t k, int32_t D,
                              const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant,
                              const float *queries, int32_t nq, int32_t K,
                              double *out_scores, int64_t *out_idx, int32_t nthreads)
{
    scan_topk_synth_impl(seed, k, D, plant_dst, plant_src, plant_kind, n_plant, queries, nq, K, out_scores, out_idx, nthreads, 1);
}

static void scan_topk_synth_impl(uint64_t seed, int64_t k, int32_t D,
                                 const int64_t *plant_dst, const int64_t *plant_src, const int32_t *plant_kind, int64_t n_plant,
                                 const float *queries, int32_t nq, int32_t K,
                                 double *out_scores, int64_t *out_idx, int32_t nthreads, int unit)
{
    if (nthreads <= 0) nthreads = 1;
    double *tsc = (double *)malloc(sizeof(double) * (size_t)nthreads * nq * K);
    int64_t *tix = (int64_t *)malloc(sizeof(int64_t) * (size_t)nthreads * nq * K);
    for (int32_t t = 0; t < nthreads * nq; t++) topk_init(tsc + (size_t)t * K, tix + (size_t)t * K, K);
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        float *row = (float *)malloc(sizeof(float) * (size_t)D);
        double *sc = tsc + (size_t)t * nq * K;
        int64_t *ix = tix + (size_t)t * nq * K;
        int64_t lo = k * t / nt, hi = k * (t + 1) / nt;
        /* first planted row >= lo */
        int64_t pp = 0;
        while (pp < n_plant && plant_dst[pp] < lo) pp++;
        for (int64_t i = lo; i < hi; i++) {
            int32_t kind = 0; int64_t src = -1;
            if (pp < n_plant && plant_dst[pp] == i) { kind = plant_kind[pp]; src = plant_src[pp]; pp++; }
            if (unit) orc_synth_row_unit_f32(seed, i, D, kind, src, row);
            else orc_synth_row_f32(seed, i, D, kind, src, row);
            for (int32_t q = 0; q < nq; q++)
                topk_push(sc + (size_t)q * K, ix + (size_t)q * K, K, orc_dot_tree_f32(queries + (size_t)q * D, row, D), i);
        }
        free(row);
    }
    /* merge per-thread lists (exact top-K under a total order => partition independent) */
    for (int32_t q = 0; q < nq; q++) {
        double *sc = out_scores + (size_t)q * K; int64_t *ix = out_idx + (size_t)q * K;
        topk_init(sc, ix, K);
        for (int t = 0; t < nthreads; t++)
            for (int32_t j = 0; j < K; j++) {
                int64_t id = tix[((size_t)t * nq + q) * K + j];
                if (id >= 0) topk_push(sc, ix, K, tsc[((size_t)t * nq + q) * K + j], id);
            }
    }
    free(tsc); free(tix);
}

/* top-K over double rows, threads over disjoint row ranges + exact merge (partition independent under the total order) */
void orc_scan_topk_f64(const double *db, int64_t k, int32_t D, const double *queries, int32_t nq, int32_t K,
                       double *out_scores, int64_t *out_idx, int32_t nthreads)
{
    if (nthreads <= 0) nthreads = 1;
    double *tsc = (double *)malloc(sizeof(double) * (size_t)nthreads * nq * K);
    int64_t *tix = (int64_t *)malloc(sizeof(int64_t) * (size_t)nthreads * nq * K);
    for (int32_t t = 0; t < nthreads * nq; t++) topk_init(tsc + (size_t)t * K, tix + (size_t)t * K, K);
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        int64_t lo = k * t / nt, hi = k * (t + 1) / nt;
        for (int64_t i = lo; i < hi; i++)
            for (int32_t q = 0; q < nq; q++)
                topk_push(tsc + ((size_t)t * nq + q) * K, tix + ((size_t)t * nq + q) * K, K,
                          orc_dot_tree_f64(queries + (size_t)q * D, db + (size_t)i * D, D), i);
    }
    for (int32_t q = 0; q < nq; q++) {
        double *sc = out_scores + (size_t)q * K; int64_t *ix = out_idx + (size_t)q * K;
        topk_init(sc, ix, K);
        for (int t = 0; t < nthreads; t++)
            for (int32_t j = 0; j < K; j++) {
                int64_t id = tix[((size_t)t * nq + q) * K + j];
                if (id >= 0) topk_push(sc, ix, K, tsc[((size_t)t * nq + q) * K + j], id);
            }
    }
    free(tsc); free(tix);
}

/* the whole score vector u = v^T M[:, :k] (Cerebro.cpp:1026) in the device's summation order; elem = 4 (float rows) or 8 */
void orc_scores(const void *db, int32_t elem, int64_t k, int32_t D, const void *query, double *u, int32_t nthreads)
{
    if (nthreads <= 0) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < k; i++)
        u[i] = elem == 8 ? orc_dot_tree_f64((const double *)query, (const double *)db + (size_t)i * D, D)
                         : orc_dot_tree_f32((const float *)query, (const float *)db + (size_t)i * D, D);
}

/* ---------------------------------------------------------------- the tick */
void orc_dot_params_default(orc_dot_params *p)
{
    p->locality = 12;                 /* Cerebro.cpp:912  int LOCALITY_THRESH = 12 */
    p->thresh = (double)(float)0.85;  /* Cerebro.cpp:913  float DOT_PROD_THRESH = 0.85, compared with a double at :1056 */
    p->lag = 50;                      /* Cerebro.cpp:914 */
    p->min_new = 3;                   /* Cerebro.cpp:962 */
    p->min_k = 5;                     /* Cerebro.cpp:1022 (strict >) */
}

static int64_t i64abs(int64_t a) { return a < 0 ? -a : a; }

void orc_loop_tick_f32(orc_loop_state *st, const orc_dot_params *p, const float *db, int32_t D, int64_t l, orc_tick_result *out)
{
    memset(out, 0, sizeof *out);
    out->idx_curr = out->idx_prev = -1;
    for (int q = 0; q < 3; q++) { out->argmax[q] = -1; out->maxv[q] = -INFINITY; }

    if (l - st->last_l < p->min_new) { out->status = 0; return; } /* :962-966  (continue; last_l untouched) */

    /* :987-989  v, vm, vmm = descriptors l-1, l-2, l-3 */
    const float *qv[3] = { db + (size_t)(l - 1) * D, db + (size_t)(l - 2) * D, db + (size_t)(l - 3) * D };
    /* :1005-1006 fill of M is the DB append itself (rows [last_l,l) already present in db) */
    int64_t k = l - p->lag; /* :1019 */
    out->status = 1;
    if (k > p->min_k) {     /* :1022 */
        out->status = 2;
        for (int q = 0; q < 3; q++) {
            /* :1026-1028 u = v^T M[:, :k];  :1035-1043 max + LAST index attaining it */
            double best = -INFINITY; int64_t arg = -1;
            for (int64_t i = 0; i < k; i++) {
                double s = orc_dot_tree_f32(qv[q], db + (size_t)i * D, D);
                if (s >= best) { best = s; arg = i; } /* >= : later index wins ties */
            }
            out->maxv[q] = best; out->argmax[q] = arg;
        }
        /* :1056 */
        if (i64abs(out->argmax[0] - out->argmax[1]) < p->locality &&
            i64abs(out->argmax[0] - out->argmax[2]) < p->locality && out->maxv[0] > p->thresh) {
            out->found = 1;                /* :1078-1081 foundLoops.push_back( t[l-1], t[u_argmax], u_max ) */
            out->idx_curr = l - 1;
            out->idx_prev = out->argmax[0];
            out->score = out->maxv[0];
        }
    }
    st->last_l = l; /* :1098 */
}

/* The same tick over a double-row DB: identical control flow, scores by orc_dot_tree_f64. */
void orc_loop_tick_f64(orc_loop_state *st, const orc_dot_params *p, const double *db, int32_t D, int64_t l, orc_tick_result *out)
{
    memset(out, 0, sizeof *out);
    out->idx_curr = out->idx_prev = -1;
    for (int q = 0; q < 3; q++) { out->argmax[q] = -1; out->maxv[q] = -INFINITY; }
    if (l - st->last_l < p->min_new) { out->status = 0; return; } /* :962-966 */
    const double *qv[3] = { db + (size_t)(l - 1) * D, db + (size_t)(l - 2) * D, db + (size_t)(l - 3) * D }; /* :987-989 */
    int64_t k = l - p->lag; /* :1019 */
    out->status = 1;
    if (k > p->min_k) {     /* :1022 */
        out->status = 2;
        for (int q = 0; q < 3; q++) {
            double best = -INFINITY; int64_t arg = -1;
            for (int64_t i = 0; i < k; i++) {
                double s = orc_dot_tree_f64(qv[q], db + (size_t)i * D, D);
                if (s >= best) { best = s; arg = i; } /* :1035-1043, later index wins ties */
            }
            out->maxv[q] = best; out->argmax[q] = arg;
        }
        if (i64abs(out->argmax[0] - out->argmax[1]) < p->locality &&
            i64abs(out->argmax[0] - out->argmax[2]) < p->locality && out->maxv[0] > p->thresh) { /* :1056 */
            out->found = 1;
            out->idx_curr = l - 1;
            out->idx_prev = out->argmax[0];
            out->score = out->maxv[0];
        }
    }
    st->last_l = l; /* :1098 */
}

/* ---------------------------------------------------------------- reference-faithful fp64 path (cpu_baseline) */
void orc_ref_scan_f64_colmajor(const double *M, int32_t D, int64_t k,
                               const double *v, const double *vm, const double *vmm,
                               double *u, double *um, double *umm, double maxv[3], int64_t argmax[3])
{
    /* :1026-1028 three separate products => M is streamed three times, as Eigen does */
    for (int64_t i = 0; i < k; i++) u[i] = orc_dot_seq_f64(v, M + (size_t)i * D, D);
    for (int64_t i = 0; i < k; i++) um[i] = orc_dot_seq_f64(vm, M + (size_t)i * D, D);
    for (int64_t i = 0; i < k; i++) umm[i] = orc_dot_seq_f64(vmm, M + (size_t)i * D, D);
    /* :1035-1037 maxCoeff */
    double a = u[0], b = um[0], c = umm[0];
    for (int64_t i = 1; i < k; i++) { if (u[i] > a) a = u[i]; if (um[i] > b) b = um[i]; if (umm[i] > c) c = umm[i]; }
    /* :1038-1043 */
    int64_t ia = -1, ib = -1, ic = -1;
    for (int64_t ii = 0; ii < k; ii++) {
        if (u[ii] == a) ia = ii;
        if (um[ii] == b) ib = ii;
        if (umm[ii] == c) ic = ii;
    }
    maxv[0] = a; maxv[1] = b; maxv[2] = c;
    argmax[0] = ia; argmax[1] = ib; argmax[2] = ic;
}

/* Same statements with the three products parallelised over columns (OpenMP static schedule, `nthreads` threads): what the
 * reference would do had it enabled Eigen's OpenMP GEMV (it does not, CMakeLists.txt:42).  SURVEY.md 8d (ii).  Every u[i]
 * is the same sequential dot product, so results equal the single-thread function bit for bit. */
void orc_ref_scan_f64_colmajor_omp(const double *M, int32_t D, int64_t k,
                                   const double *v, const double *vm, const double *vmm,
                                   double *u, double *um, double *umm, double maxv[3], int64_t argmax[3], int32_t nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < k; i++) u[i] = orc_dot_seq_f64(v, M + (size_t)i * D, D);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < k; i++) um[i] = orc_dot_seq_f64(vm, M + (size_t)i * D, D);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < k; i++) umm[i] = orc_dot_seq_f64(vmm, M + (size_t)i * D, D);
    double a = u[0], b = um[0], c = umm[0];
    for (int64_t i = 1; i < k; i++) { if (u[i] > a) a = u[i]; if (um[i] > b) b = um[i]; if (umm[i] > c) c = umm[i]; }
    int64_t ia = -1, ib = -1, ic = -1;
    for (int64_t ii = 0; ii < k; ii++) {
        if (u[ii] == a) ia = ii;
        if (um[ii] == b) ib = ii;
        if (umm[ii] == c) ic = ii;
    }
    maxv[0] = a; maxv[1] = b; maxv[2] = c;
    argmax[0] = ia; argmax[1] = ib; argmax[2] = ic;
}

/* First-touch helper for the multi-thread baseline: replicate a block of `src_cols` columns into M (k columns) in
 * parallel with the same static schedule, so pages land on the NUMA node of the thread that will read them. */
void orc_tile_columns_omp(double *M, int32_t D, int64_t k, const double *src, int64_t src_cols, int32_t nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < k; i++) {
        const double *s = src + (size_t)(i % src_cols) * D;
        double *d = M + (size_t)i * D;
        for (int32_t j = 0; j < D; j++) d[j] = s[j];
    }
}

/* ---------------------------------------------------------------- Eigen-order GEMV emulation (parity evidence, not a baseline)
 * What  u = v.transpose() * M.leftCols(k)  (/root/reference/src/Cerebro.cpp:1026-1028) computes in the reference's build, term
 * order included.  Eigen is not vendored and not installed here, so this restates its published algorithm; the version followed
 * is Eigen 3.3.x (ROS Kinetic / Ubuntu 16.04 ship 3.2.92 = 3.3-beta1, the reference's docker image 3.3.4), file
 * Eigen/src/Core/products/GeneralMatrixVector.h, struct general_matrix_vector_product<Index, LhsScalar, LhsMapper, RowMajor, ...>:
 *   - a row-vector times a column-major matrix is evaluated as the transposed product M^T v: a ROW-major GEMV whose "rows" are the
 *     columns of M (each D contiguous doubles) -- GeneralProduct.h / ProductEvaluators.h gemv_dense_selector<OnTheLeft, ...>;
 *   - rows are taken four at a time (rowsAtOnce = 4; the tail rows one at a time): that blocking shares the loads of v, it does not
 *     change any row's arithmetic;
 *   - per row: scalar accumulator tmp = 0 over the first `alignedStart` elements (0 here: VectorXd storage is 16-byte aligned),
 *     then ONE packet accumulator ptmp = pset1(0) over j = alignedStart .. alignedSize in steps of the packet size with
 *     ptmp = pmadd(lhs(j), rhs(j), ptmp), then tmp += predux(ptmp), then the scalar tail j = alignedSize .. depth with
 *     tmp += lhs(j) * rhs(j); finally res[i] += alpha * tmp with alpha = 1 and res zero-initialised (dst.setZero() first);
 *   - the reference is built with CMAKE_BUILD_TYPE Release and CMAKE_CXX_FLAGS " -Wl,-no-as-needed" (CMakeLists.txt:42-44: no
 *     -march) on x86-64: SSE2 packets of 2 doubles (Packet2d), pmadd = padd(pmul(a, b), c) (no FMA), and
 *     predux(Packet2d a) = a[0] + a[1] (arch/SSE/PacketMath.h).  packet = 4 / fma = 1 restate an AVX2+FMA build
 *     (Packet4d, predux = (a0 + a1) + (a2 + a3), arch/AVX/PacketMath.h) for comparison.
 * So with packet P the value is  ((s_0 + s_1) [+ (s_2 + s_3)]) + tail,  s_c = sum over j = c, c + P, ... taken in ascending order with
 * one rounding per addition.  Built with -ffp-contract=off, so `a * b + c` below is two roundings unless fma is asked for. */
double orc_dot_eigen_gemv_f64(const double *v, const double *col, int32_t D, int32_t packet, int32_t fma_flag, int32_t aligned_start)
{
    double tmp = 0.0;
    int32_t j = 0;
    if (aligned_start > D) aligned_start = D;
    for (; j < aligned_start; j++) tmp += col[j] * v[j];
    const int32_t P = packet == 4 ? 4 : (packet == 2 ? 2 : 1);
    if (P > 1) {
        const int32_t aligned_size = aligned_start + ((D - aligned_start) & ~(P - 1));
        if (aligned_size > aligned_start) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            for (; j < aligned_size; j += P)
                for (int32_t c = 0; c < P; c++) acc[c] = fma_flag ? fma(col[j + c], v[j + c], acc[c]) : col[j + c] * v[j + c] + acc[c];
            tmp += P == 2 ? acc[0] + acc[1] : (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
    }
    for (; j < D; j++) tmp += col[j] * v[j];
    return 0.0 + 1.0 * tmp;   /* res[i] += alpha * tmp */
}

/* Cerebro.cpp:1026-1043 with the three products in Eigen's order: fp64 column-major M, maxCoeff, last index attaining it. */
void orc_ref_scan_f64_eigen_order(const double *M, int32_t D, int64_t k, const double *v, const double *vm, const double *vmm,
                                  double *u, double *um, double *umm, double maxv[3], int64_t argmax[3],
                                  int32_t packet, int32_t fma_flag, int32_t nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < k; i++) {
        u[i] = orc_dot_eigen_gemv_f64(v, M + (size_t)i * D, D, packet, fma_flag, 0);
        um[i] = orc_dot_eigen_gemv_f64(vm, M + (size_t)i * D, D, packet, fma_flag, 0);
        umm[i] = orc_dot_eigen_gemv_f64(vmm, M + (size_t)i * D, D, packet, fma_flag, 0);
    }
    double a = u[0], b = um[0], c = umm[0];
    for (int64_t i = 1; i < k; i++) { if (u[i] > a) a = u[i]; if (um[i] > b) b = um[i]; if (umm[i] > c) c = umm[i]; }   /* maxCoeff */
    int64_t ia = -1, ib = -1, ic = -1;
    for (int64_t ii = 0; ii < k; ii++) {   /* :1038-1043 */
        if (u[ii] == a) ia = ii;
        if (um[ii] == b) ib = ii;
        if (umm[ii] == c) ic = ii;
    }
    maxv[0] = a; maxv[1] = b; maxv[2] = c;
    argmax[0] = ia; argmax[1] = ib; argmax[2] = ic;
}

/* ---------------------------------------------------------------- the Eigen-order GEMV at the speed Eigen runs it (cpu_baseline)
 * orc_dot_eigen_gemv_f64 above states the ORDER; this states the order AND the shape of Eigen 3.3's row-major GEMV kernel for the
 * reference's build (GeneralMatrixVector.h, RowMajor specialisation; SSE2, CMakeLists.txt:42-44 has no -march): the "rows" of
 * M^T (= columns of M, D contiguous doubles each) are taken FOUR at a time with one Packet2d accumulator each -- four independent
 * mulpd + addpd chains over the shared loads of v -- then predux (a0 + a1) per row, then the tail rows one at a time.  The
 * blocking changes no row's arithmetic: u[i] equals orc_dot_eigen_gemv_f64(v, M + i*D, D, 2, 0, 0) bit for bit
 * (tests/test_oracle_eigen_order.py), and this is what bench.py times as the headline CPU figure.  x86-64 only (the GPU boxes'
 * hosts); elsewhere it falls back to the scalar emulation. */
#if defined(__SSE2__)
#include <emmintrin.h>
static void eigen_gemv_rowmajor_sse2(const double *M, int32_t D, int64_t i0, int64_t i1, const double *v, double *u)
{
    const int32_t aligned = D & ~1;
    int64_t i = i0;
    for (; i + 4 <= i1; i += 4) {
        const double *r0 = M + (size_t)i * D, *r1 = r0 + D, *r2 = r1 + D, *r3 = r2 + D;
        __m128d p0 = _mm_setzero_pd(), p1 = p0, p2 = p0, p3 = p0;
        for (int32_t j = 0; j < aligned; j += 2) {
            const __m128d b = _mm_loadu_pd(v + j);
            p0 = _mm_add_pd(_mm_mul_pd(_mm_loadu_pd(r0 + j), b), p0);   /* pmadd without FMA: padd(pmul(a, b), c) */
            p1 = _mm_add_pd(_mm_mul_pd(_mm_loadu_pd(r1 + j), b), p1);
            p2 = _mm_add_pd(_mm_mul_pd(_mm_loadu_pd(r2 + j), b), p2);
            p3 = _mm_add_pd(_mm_mul_pd(_mm_loadu_pd(r3 + j), b), p3);
        }
        double t[4][2];
        _mm_storeu_pd(t[0], p0); _mm_storeu_pd(t[1], p1); _mm_storeu_pd(t[2], p2); _mm_storeu_pd(t[3], p3);
        const double *r[4] = {r0, r1, r2, r3};
        for (int c = 0; c < 4; c++) {
            double tmp = 0.0;
            tmp += t[c][0] + t[c][1];                                    /* tmp += predux(ptmp) */
            for (int32_t j = aligned; j < D; j++) tmp += r[c][j] * v[j];
            u[i + c] = 0.0 + 1.0 * tmp;                                  /* res[i] += alpha * tmp */
        }
    }
    for (; i < i1; i++) u[i] = orc_dot_eigen_gemv_f64(v, M + (size_t)i * D, D, 2, 0, 0);
}
#else
static void eigen_gemv_rowmajor_sse2(const double *M, int32_t D, int64_t i0, int64_t i1, const double *v, double *u)
{
    for (int64_t i = i0; i < i1; i++) u[i] = orc_dot_eigen_gemv_f64(v, M + (size_t)i * D, D, 2, 0, 0);
}
#endif

/* Cerebro.cpp:1026-1043 as the reference's build executes it: THREE separate GEMVs (M is streamed three times), three maxCoeff,
 * one last-index argmax loop.  nthreads = 1 is the reference (its dot_product_th is one thread, Eigen GEMV is not parallel without
 * OpenMP); nthreads > 1 splits every GEMV's columns statically, results unchanged. */
void orc_ref_scan_f64_eigen_gemv3(const double *M, int32_t D, int64_t k, const double *v, const double *vm, const double *vmm,
                                  double *u, double *um, double *umm, double maxv[3], int64_t argmax[3], int32_t nthreads)
{
    const double *q[3] = {v, vm, vmm};
    double *out[3] = {u, um, umm};
    if (nthreads < 1) nthreads = 1;
    for (int g = 0; g < 3; g++) {
        if (nthreads == 1) eigen_gemv_rowmajor_sse2(M, D, 0, k, q[g], out[g]);
        else {
#pragma omp parallel num_threads(nthreads)
            {
#ifdef _OPENMP
                const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
                const int t = 0, nt = 1;
#endif
                const int64_t per = ((k + nt - 1) / nt + 3) & ~(int64_t)3;   /* whole blocks of four rows per thread */
                const int64_t a = (int64_t)t * per, b = a + per < k ? a + per : k;
                if (a < b) eigen_gemv_rowmajor_sse2(M, D, a, b, q[g], out[g]);
            }
        }
    }
    double a = u[0], b = um[0], c = umm[0];
    for (int64_t i = 1; i < k; i++) { if (u[i] > a) a = u[i]; if (um[i] > b) b = um[i]; if (umm[i] > c) c = umm[i]; }   /* maxCoeff */
    int64_t ia = -1, ib = -1, ic = -1;
    for (int64_t ii = 0; ii < k; ii++) {   /* :1038-1043 */
        if (u[ii] == a) ia = ii;
        if (um[ii] == b) ib = ii;
        if (umm[ii] == c) ic = ii;
    }
    maxv[0] = a; maxv[1] = b; maxv[2] = c;
    argmax[0] = ia; argmax[1] = ib; argmax[2] = ic;
}
