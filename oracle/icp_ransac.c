/*
 * icp_ransac.c -- CPU oracle for cerebro's Umeyama-ICP-in-RANSAC leg.
 *
 * TEST INFRASTRUCTURE ONLY (see cerebro_oracle.h).  Restates
 *   /root/reference/src/DlsPnpWithRansac.h:104-166   AlignPointCloudsUmeyamaWithRansac: SampleSize 10 (:118),
 *        EstimateModel = theia::AlignPointCloudsUmeyama on the sample, accept iff min(s, 1/s) > 0.9 (:135-146; the pose
 *        keeps R and t, the scale is only a sanity gate), Error = || R a + t - b ||_2 (:152-164; the weight f is 1:
 *        its condition `z < 1 && z > 8` is never true)
 *   /root/reference/src/DlsPnpWithRansac.cpp:16-122  StaticTheiaPoseCompute::P3P_ICP: < 20 points -> -1 (:19-22),
 *        RANSAC parameters .1 / .7 / 50 / 5 / use_mle (:88-93), returns summary.confidence (:121)
 * theia::AlignPointCloudsUmeyama is un-vendored (PARITY UNPINNED, as for DlsPnp): restated from Umeyama, "Least-squares
 * estimation of transformation parameters between two point patterns", PAMI 1991 -- with the 3x3 SVD defined here as a
 * fixed-sweep cyclic Jacobi eigen-decomposition of Sigma^T Sigma (Eigen uses a two-sided Jacobi SVD; both converge to the
 * same factors, the iteration details are this oracle's definition).  RANSAC driver, sampler and RNG: pnp_ransac.c.
 */
#include "cerebro_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define JACOBI_SWEEPS 8

/* Symmetric 3x3 eigen-decomposition by cyclic Jacobi: A = V diag(w) V^T, fixed number of sweeps, pairs (0,1),(0,2),(1,2).
 * Eigenvalues sorted descending (stable selection), det(V) forced to +1 by flipping the last column. */
static void jacobi_eig3(double A[3][3], double V[3][3], double w[3])
{
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < JACOBI_SWEEPS; sweep++)
        for (int k = 0; k < 3; k++) {
            const int p = P[k], q = Q[k];
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
            /* A <- J^T A J, J = rotation in the (p,q) plane */
            for (int r = 0; r < 3; r++) { /* columns p, q */
                const double arp = A[r][p], arq = A[r][q];
                A[r][p] = c * arp - s * arq;
                A[r][q] = s * arp + c * arq;
            }
            for (int r = 0; r < 3; r++) { /* rows p, q */
                const double apr = A[p][r], aqr = A[q][r];
                A[p][r] = c * apr - s * aqr;
                A[q][r] = s * apr + c * aqr;
            }
            for (int r = 0; r < 3; r++) {
                const double vrp = V[r][p], vrq = V[r][q];
                V[r][p] = c * vrp - s * vrq;
                V[r][q] = s * vrp + c * vrq;
            }
        }
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++) /* selection sort, descending, ties keep the lower index first */
        for (int j = i + 1; j < 3; j++)
            if (A[ord[j]][ord[j]] > A[ord[i]][ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Vs[3][3];
    for (int k = 0; k < 3; k++) {
        w[k] = A[ord[k]][ord[k]];
        for (int r = 0; r < 3; r++) Vs[r][k] = V[r][ord[k]];
    }
    const double det = Vs[0][0] * (Vs[1][1] * Vs[2][2] - Vs[1][2] * Vs[2][1]) - Vs[0][1] * (Vs[1][0] * Vs[2][2] - Vs[1][2] * Vs[2][0]) +
                       Vs[0][2] * (Vs[1][0] * Vs[2][1] - Vs[1][1] * Vs[2][0]);
    if (det < 0.0)
        for (int r = 0; r < 3; r++) Vs[r][2] = -Vs[r][2];
    memcpy(V, Vs, sizeof Vs);
}

/* b ~ s R a + t  (least squares over n correspondences, unit weights).  R row-major.  Returns 0, or -1 if degenerate. */
int orc_umeyama(const double *a, const double *b, int32_t n, double R[9], double t[3], double *scale)
{
    double ma[3] = {0, 0, 0}, mb[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { ma[k] = ma[k] + a[3 * i + k]; mb[k] = mb[k] + b[3 * i + k]; }
    for (int k = 0; k < 3; k++) { ma[k] = ma[k] / (double)n; mb[k] = mb[k] / (double)n; }
    double Sg[3][3] = {{0}}, var_a = 0.0;
    for (int i = 0; i < n; i++) {
        double da[3], db[3];
        for (int k = 0; k < 3; k++) { da[k] = a[3 * i + k] - ma[k]; db[k] = b[3 * i + k] - mb[k]; }
        var_a = var_a + ((da[0] * da[0] + da[1] * da[1]) + da[2] * da[2]);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Sg[r][c] = Sg[r][c] + db[r] * da[c];
    }
    var_a = var_a / (double)n;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Sg[r][c] = Sg[r][c] / (double)n;
    /* SVD of Sigma through the eigen-decomposition of Sigma^T Sigma = V diag(sigma^2) V^T */
    double StS[3][3], V[3][3], w[3], sig[3], U[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) StS[r][c] = (Sg[0][r] * Sg[0][c] + Sg[1][r] * Sg[1][c]) + Sg[2][r] * Sg[2][c];
    jacobi_eig3(StS, V, w);
    for (int k = 0; k < 3; k++) sig[k] = sqrt(w[k] > 0.0 ? w[k] : 0.0);
    if (!(sig[1] > 1e-6 * sig[0]) || !(sig[0] > 0.0)) return -1; /* rank < 2 (sigma from sqrt of an eigenvalue: noise floor ~1e-8): rotation undetermined */
    for (int k = 0; k < 2; k++)
        for (int r = 0; r < 3; r++) U[r][k] = ((Sg[r][0] * V[0][k] + Sg[r][1] * V[1][k]) + Sg[r][2] * V[2][k]) / sig[k];
    double S22 = 1.0;
    if (sig[2] > 1e-6 * sig[0]) {
        for (int r = 0; r < 3; r++) U[r][2] = ((Sg[r][0] * V[0][2] + Sg[r][1] * V[1][2]) + Sg[r][2] * V[2][2]) / sig[2];
        const double detU = U[0][0] * (U[1][1] * U[2][2] - U[1][2] * U[2][1]) - U[0][1] * (U[1][0] * U[2][2] - U[1][2] * U[2][0]) +
                            U[0][2] * (U[1][0] * U[2][1] - U[1][1] * U[2][0]);
        if (detU < 0.0) S22 = -1.0; /* det(U) det(V) < 0, det(V) = +1 by construction */
    } else { /* rank 2: complete U to a proper rotation */
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[3 * r + c] = (U[r][0] * V[c][0] + U[r][1] * V[c][1]) + (S22 * U[r][2]) * V[c][2];
    const double s = ((sig[0] + sig[1]) + S22 * sig[2]) / var_a;
    for (int r = 0; r < 3; r++) t[r] = mb[r] - s * ((R[3 * r] * ma[0] + R[3 * r + 1] * ma[1]) + R[3 * r + 2] * ma[2]);
    *scale = s;
    return 0;
}

/* AlignPointCloudsUmeyamaWithRansac::Error (DlsPnpWithRansac.h:152-164) */
double orc_icp_error(const double *T, const double *a, const double *b)
{
    const double x = ((T[0] * a[0] + T[4] * a[1]) + T[8] * a[2]) + T[12];
    const double y = ((T[1] * a[0] + T[5] * a[1]) + T[9] * a[2]) + T[13];
    const double z = ((T[2] * a[0] + T[6] * a[1]) + T[10] * a[2]) + T[14];
    const double dx = x - b[0], dy = y - b[1], dz = z - b[2];
    return sqrt((dx * dx + dy * dy) + dz * dz);
}

void orc_icp_score_model(const double *T, const double *A, const double *B, int32_t N, double thresh, int32_t use_mle,
                         double *cost, int32_t *n_inliers, uint8_t *mask)
{
    double acc[64];
    int32_t cnt = 0;
    for (int L = 0; L < 64; L++) acc[L] = 0.0;
    for (int32_t base = 0; base < N; base += 64)
        for (int L = 0; L < 64; L++) {
            const int32_t i = base + L;
            if (i >= N) continue;
            const double r = orc_icp_error(T, A + 3 * i, B + 3 * i);
            const int in = r < thresh;
            if (mask) mask[i] = (uint8_t)in;
            cnt += in;
            acc[L] = acc[L] + (in ? r : thresh);
        }
    for (int m = 32; m >= 1; m >>= 1) {
        double nxt[64];
        for (int L = 0; L < 64; L++) nxt[L] = acc[L] + acc[L ^ m];
        memcpy(acc, nxt, sizeof acc);
    }
    *n_inliers = cnt;
    *cost = use_mle ? acc[0] : (double)(N - cnt);
}

/* EstimateModel (DlsPnpWithRansac.h:121-150) on the hypothesis' sample.  Returns 1 and b_T_a (column-major) iff accepted. */
int orc_icp_hypothesis(const double *A, const double *B, int32_t N, uint64_t seed, int32_t hyp, int32_t S, double T[16], double *scale_out)
{
    return orc_icp_hypothesis_sampled(A, B, N, seed, hyp, S, NULL, T, scale_out);
}

int orc_icp_hypothesis_sampled(const double *A, const double *B, int32_t N, uint64_t seed, int32_t hyp, int32_t S,
                               const int32_t *sample_in, double T[16], double *scale_out)
{
    int32_t sample[64];
    double sa[64 * 3], sb[64 * 3], R[9], t[3], s = 0.0;
    if (S > 64) S = 64;
    if (sample_in) memcpy(sample, sample_in, sizeof(int32_t) * (size_t)S);
    else orc_ransac_sample(seed, hyp, N, S, sample);
    for (int i = 0; i < S; i++) {
        memcpy(sa + 3 * i, A + 3 * sample[i], 3 * sizeof(double));
        memcpy(sb + 3 * i, B + 3 * sample[i], 3 * sizeof(double));
    }
    if (orc_umeyama(sa, sb, S, R, t, &s) != 0) return 0;
    if (scale_out) *scale_out = s;
    const double inv = 1.0 / s;
    if (!((s < inv ? s : inv) > 0.9)) return 0; /* min(s, 1/s) > 0.9 (:137) */
    T[0] = R[0]; T[1] = R[3]; T[2] = R[6]; T[3] = 0.0;
    T[4] = R[1]; T[5] = R[4]; T[6] = R[7]; T[7] = 0.0;
    T[8] = R[2]; T[9] = R[5]; T[10] = R[8]; T[11] = 0.0;
    T[12] = t[0]; T[13] = t[1]; T[14] = t[2]; T[15] = 1.0;
    return 1;
}

void orc_icp_params_default(orc_ransac_params *p)
{
    orc_ransac_params_default(p);
    p->error_thresh = 0.1; /* DlsPnpWithRansac.cpp:89 */
    p->sample_size = 10;   /* DlsPnpWithRansac.h:118 */
}

/* StaticTheiaPoseCompute::P3P_ICP (DlsPnpWithRansac.cpp:16-122), RANSAC branch.  Returns 0, or -9 for N < 20. */
int orc_icp_ransac(const double *A, const double *B, int32_t N, const orc_ransac_params *p,
                   double T[16], float *confidence, uint8_t *mask, orc_ransac_summary *summary)
{
    if (N < 20) return -9;
    const int32_t S = p->sample_size;
    const double log_fail = log(p->failure_probability);
    double best_cost = DBL_MAX, best_T[16];
    int32_t best_h = -1, n_models = 0, num_it = 0, max_it;
    const int bench = p->n_hypotheses > 0;
    if (bench) max_it = p->n_hypotheses;
    else {
        max_it = p->max_iterations;
        if (p->min_inlier_ratio > 0)
            max_it = orc_ransac_max_iterations(S, p->min_inlier_ratio, log_fail, p->min_iterations, p->max_iterations);
    }
    int32_t *table = NULL;   /* sampler mode 1: theia's persistent permutation (pnp_ransac.c orc_ransac_sample_persistent) */
    if (p->sampler == 1) {
        table = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_it * (size_t)S);
        orc_ransac_sample_persistent(p->seed, max_it, N, S, table);
    }
    for (num_it = 0; num_it < max_it; num_it++) {
        double Th[16], cost;
        int32_t nin;
        if (!orc_icp_hypothesis_sampled(A, B, N, p->seed, num_it, S, table ? table + (size_t)num_it * S : NULL, Th, NULL)) continue;
        n_models++;
        orc_icp_score_model(Th, A, B, N, p->error_thresh, p->use_mle, &cost, &nin, NULL);
        if (cost < best_cost) {
            best_cost = cost; best_h = num_it;
            memcpy(best_T, Th, sizeof Th);
            if (!bench) {
                const double ratio = (double)nin / (double)N;
                if (ratio < (double)S / (double)N) continue;
                const int32_t mi = orc_ransac_max_iterations(S, ratio, log_fail, p->min_iterations, p->max_iterations);
                if (mi < max_it) max_it = mi;
            }
        }
    }
    free(table);
    int32_t nin = 0;
    if (best_h >= 0) {
        double c;
        memcpy(T, best_T, sizeof best_T);
        orc_icp_score_model(best_T, A, B, N, p->error_thresh, p->use_mle, &c, &nin, mask);
        const double ratio = (double)nin / (double)N;
        *confidence = (float)(1.0 - pow(1.0 - pow(ratio, (double)S), (double)num_it));
    } else {
        for (int i = 0; i < 16; i++) T[i] = NAN;
        if (mask) memset(mask, 0, (size_t)N);
        *confidence = 0.0f;
    }
    if (summary) {
        summary->n_iterations = num_it;
        summary->n_inliers = nin;
        summary->best_hypothesis = best_h;
        summary->n_models = n_models;
        summary->best_cost = best_h >= 0 ? best_cost : INFINITY;
    }
    return 0;
}
