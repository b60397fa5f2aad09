/*
 * surrogate.c -- TEST INFRASTRUCTURE ONLY (see cerebro_oracle.h): the pieces of the EuRoC-shaped surrogate run that must be
 * bit-reproducible on any machine (tests/euroc_surrogate.py orchestrates them with integer / constant-only Python):
 *   - orc_ar1_step: one step of a unit-norm AR(1) descriptor walk d_t = normalize(a d_{t-1} + sqrt(1 - a^2) n_t) in plain
 *     sequential C (no BLAS, no vectorised reductions whose blocking depends on the CPU);
 *   - orc_loop_tick_order: one pass of Cerebro::descrip_N__dot__descrip_0_N's loop body (/root/reference/src/Cerebro.cpp:956-1100)
 *     in a CHOSEN summation order -- the device's fixed tree (what the GPU is bit-exact against) or Eigen 3.3's SSE2 row-major
 *     GEMV order (what the reference's Release build computes, oracle/dot_scan.c orc_dot_eigen_gemv_f64) -- with OpenMP over the
 *     DB rows, plus the gap between the best and the second-best score of every query (how close the argmax was).
 * BASELINE configs 1 and 5 ask for bit-identical candidate selection on EuRoC MH-01(..05); the dataset, the NetVLAD weights and a
 * recorded loopcandidates_liverun.json are absent from this image, so the "recorded run" of the surrogate is produced by the
 * Eigen-order path here and the GPU replay is compared with it.
 */
#include "cerebro_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* out = normalize(alpha * prev + sqrt(1 - alpha^2) * noise); round_f32: every element is then rounded to float32 (a Keras NetVLAD
 * emits float32 VALUES on the float64 wire, whole_image_desc_compute_server.py:631,648); otherwise genuine doubles (ReljaNetVLAD's
 * numpy WPCA output, :148-149).  prev may be NULL (first frame: out = normalize(noise)). */
void orc_ar1_step(const double *prev, const float *noise, double alpha, int32_t D, int32_t round_f32, double *out)
{
    const double b = sqrt(1.0 - alpha * alpha);
    double ss = 0.0;
    for (int32_t i = 0; i < D; i++) {
        const double t = prev ? alpha * prev[i] + b * (double)noise[i] : (double)noise[i];
        out[i] = t;
        ss += t * t;
    }
    const double inv = 1.0 / sqrt(ss);
    for (int32_t i = 0; i < D; i++) {
        const double t = out[i] * inv;
        out[i] = round_f32 ? (double)(float)t : t;
    }
}

typedef struct { double best, second; int64_t arg; } best2;

static inline void best2_offer(best2 *b, double s, int64_t i)
{
    if (s >= b->best) { b->second = b->best; b->best = s; b->arg = i; }   /* >= : the LATER index wins ties (:1038-1043) */
    else if (s > b->second) b->second = s;
}

/* order 0: fixed tree (orc_dot_tree_f32 / _f64 by elem); order 1: Eigen SSE2 order, db must hold doubles (elem 8) */
void orc_loop_tick_order(orc_loop_state *st, const orc_dot_params *p, const void *db, int32_t elem, int32_t D, int64_t l,
                         int32_t order, int32_t nthreads, orc_tick_result *out, double gap[3])
{
    memset(out, 0, sizeof *out);
    out->idx_curr = out->idx_prev = -1;
    for (int q = 0; q < 3; q++) { out->argmax[q] = -1; out->maxv[q] = -INFINITY; if (gap) gap[q] = INFINITY; }
    if (l - st->last_l < p->min_new) { out->status = 0; return; }   /* :962-966 */
    const char *base = (const char *)db;
    const size_t rb = (size_t)D * elem;
    const void *qv[3] = { base + (size_t)(l - 1) * rb, base + (size_t)(l - 2) * rb, base + (size_t)(l - 3) * rb };   /* :987-989 */
    const int64_t k = l - p->lag;   /* :1019 */
    out->status = 1;
    if (k > p->min_k) {             /* :1022 */
        out->status = 2;
        if (nthreads < 1) nthreads = 1;
        best2 *tb = (best2 *)malloc(sizeof(best2) * 3 * (size_t)nthreads);
        for (int i = 0; i < 3 * nthreads; i++) { tb[i].best = -INFINITY; tb[i].second = -INFINITY; tb[i].arg = -1; }
#pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
            const int t = 0, nt = 1;
#endif
            const int64_t per = (k + nt - 1) / nt, a = (int64_t)t * per, b = a + per < k ? a + per : k;
            for (int64_t i = a; i < b; i++) {
                const void *row = base + (size_t)i * rb;
                for (int q = 0; q < 3; q++) {
                    double s;
                    if (order == 1) s = orc_dot_eigen_gemv_f64((const double *)qv[q], (const double *)row, D, 2, 0, 0);
                    else if (elem == 8) s = orc_dot_tree_f64((const double *)qv[q], (const double *)row, D);
                    else s = orc_dot_tree_f32((const float *)qv[q], (const float *)row, D);
                    best2_offer(&tb[t * 3 + q], s, i);
                }
            }
        }
        for (int q = 0; q < 3; q++) {   /* threads hold ascending row ranges: combine in order, later index wins ties */
            best2 g = { -INFINITY, -INFINITY, -1 };
            for (int t = 0; t < nthreads; t++) {
                const best2 *b = &tb[t * 3 + q];
                if (b->arg < 0) continue;
                /* merge two (best, second) pairs */
                double s2 = g.second > b->second ? g.second : b->second;
                if (b->best >= g.best) { if (g.best > s2) s2 = g.best; g.best = b->best; g.arg = b->arg; }
                else if (b->best > s2) s2 = b->best;
                g.second = s2;
            }
            out->maxv[q] = g.best; out->argmax[q] = g.arg;
            if (gap) gap[q] = g.best - g.second;
        }
        free(tb);
        const int64_t d1 = out->argmax[0] - out->argmax[1], d2 = out->argmax[0] - out->argmax[2];
        if ((d1 < 0 ? -d1 : d1) < p->locality && (d2 < 0 ? -d2 : d2) < p->locality && out->maxv[0] > p->thresh) {   /* :1056 */
            out->found = 1;   /* :1078-1081 */
            out->idx_curr = l - 1;
            out->idx_prev = out->argmax[0];
            out->score = out->maxv[0];
        }
    }
    st->last_l = l;   /* :1098 */
}
