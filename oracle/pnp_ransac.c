/*
 * pnp_ransac.c -- CPU oracle for cerebro's DLS-PnP-in-RANSAC pose verifier.
 *
 * TEST INFRASTRUCTURE ONLY (see cerebro_oracle.h).
 *
 * What is restated from /root/reference (cited per function):
 *   src/DlsPnpWithRansac.h:31-39    CorrespondencePair_3d2d / RelativePose
 *   src/DlsPnpWithRansac.h:45       SampleSize() == 15
 *   src/DlsPnpWithRansac.h:48-72    EstimateModel: DlsPnp on the sample, accept iff exactly ONE solution
 *   src/DlsPnpWithRansac.h:75-99    Error: L1 reprojection error in normalized image coordinates
 *   src/DlsPnpWithRansac.cpp:132-245 StaticTheiaPoseCompute::PNP: <20 points -> -1, RANSAC parameters
 *                                    (.03, .7, 50, 5, use_mle), outputs best b_T_a and summary.confidence
 *
 * What is NOT under /root/reference: Theia-SfM (find_package(Theia REQUIRED), CMakeLists.txt:27; un-vendored,
 * NO VERSION PINNED: ~2018-19 master).  theia::Ransac / RandomSampler / MLEQualityMeasurement / DlsPnp and
 * Eigen's EigenSolver / PartialPivLU are restated here from their PUBLISHED algorithms:
 *   - RANSAC driver: SampleConsensusEstimator as summarised in SURVEY.md Appendix A.1;
 *   - DLS-PnP: Hesch & Roumeliotis, "A Direct Least-Squares (DLS) Method for PnP", ICCV 2011 (Cayley
 *     parameters, 3 cubics, degree-7 Macaulay resultant with a random linear form, 27x27 action matrix);
 *   - real nonsymmetric eigenproblem: Householder reduction to Hessenberg form + Francis double-shift QR
 *     with accumulated transformations + back-substitution (EISPACK orthes/ortran/hqr2, Wilkinson & Reinsch,
 *     Handbook for Automatic Computation II -- the algorithm Eigen::EigenSolver implements);
 *   - LU with partial (row) pivoting.
 * The reference holds no golden vectors, KATs or fixtures for this path and seeds nothing (Theia's sampler is
 * time-seeded), so:                 ***  PARITY UNPINNED for the Theia/Eigen internals  ***
 * (sample sequence, random linear form, root filter, eigen-solver iteration details).  This file DEFINES them:
 * counter-based RNG, fresh sampler permutation per hypothesis, fixed operation order (compile with
 * -ffp-contract=off).  The HIP kernels are checked against THIS definition (inlier masks bit-exact, poses
 * <= 1e-4 relative Frobenius); the oracle itself is validated by self-consistency (noise-free recovery) and by
 * an independent numpy implementation (tests/np_mirror_pnp.py, np.linalg.eig / np.linalg.solve).
 */
#include "cerebro_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ================================================================ operation counter (test infrastructure) */
/* -DORC_FLOP_COUNT (oracle/_build/liboracle_flops.so only): the fp64 operations this solver EXECUTES per stage,
 * counted where they happen -- every +, -, *, / and sqrt is one operation; compares, fabs, sign flips, frexp/ldexp scalings,
 * integer work and data movement are not counted.  bench.py prices the PnP kernels' fp64-vector roofline from this count
 * (tests/test_oracle_flops.py pins it) instead of an estimate.  Stage 1 counts the elimination as the reference algorithm runs
 * it (rows with a zero multiplier skipped); orc_flop_dense_lu is the same elimination without the skip (what a dense update does). */
#ifdef ORC_FLOP_COUNT
static __thread long orc_flops[ORC_FLOP_STAGES];
static __thread long orc_flop_dense_lu;
long orc_panel_rows_total, orc_panel_rows_allzero;   /* (remaining row, 4-column panel) pairs of the elimination / those whose 4 multipliers are all 0 */
#define FL(stage, n) (orc_flops[stage] += (long)(n))
void orc_flop_counts(long out[ORC_FLOP_STAGES], long *dense_lu, int reset)
{
    for (int i = 0; i < ORC_FLOP_STAGES; i++) { if (out) out[i] = orc_flops[i]; if (reset) orc_flops[i] = 0; }
    if (dense_lu) *dense_lu = orc_flop_dense_lu;
    if (reset) orc_flop_dense_lu = 0;
}
#else
#define FL(stage, n) ((void)0)
#endif
enum { FS_CUBICS = 0, FS_LU = 1, FS_ACTION = 2, FS_HESS = 3, FS_QR = 4, FS_EIGVEC = 5, FS_POSE = 6, FS_SCORE = 7 };

/* ================================================================ counter-based RNG + sampler */
uint64_t orc_rng_draw(uint64_t seed, uint32_t hyp, uint32_t draw)
{
    return orc_splitmix64(seed ^ ((uint64_t)hyp << 32) ^ (uint64_t)draw);
}

/* theia::RandomSampler restated (partial Fisher-Yates), but with a FRESH identity permutation per hypothesis
 * so that hypotheses are independent (documented departure: changes which samples are drawn, not their
 * distribution).  RandInt(lo,hi) = lo + x % (hi-lo+1). */
void orc_ransac_sample(uint64_t seed, int32_t hyp, int32_t N, int32_t S, int32_t *out)
{
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    for (int32_t i = 0; i < N; i++) idx[i] = i;
    for (int32_t i = 0; i < S; i++) {
        uint64_t x = orc_rng_draw(seed, (uint32_t)hyp, (uint32_t)i);
        int32_t j = i + (int32_t)(x % (uint64_t)(N - i));
        int32_t t = idx[i]; idx[i] = idx[j]; idx[j] = t;
    }
    for (int32_t i = 0; i < S; i++) out[i] = idx[i];
    free(idx);
}

/* theia::RandomSampler AS WRITTEN (sampler mode 1): the permutation is initialised once (Initialize: 0..N-1) and every Sample()
 * continues on the array the previous one left -- which is what the reference runs, since StaticTheiaPoseCompute::PNP builds ONE
 * theia::Ransac (hence one sampler) per estimation (DlsPnpWithRansac.cpp:216-221).  RandInt(i, N-1) = i + draw(hyp, i) % (N - i),
 * the same counter-based draws as mode 0 (Theia's mt19937 is time-seeded).  Hypothesis h then depends on all earlier ones, so the
 * device gets the table from the host; mode 0 stays the default because it lets hypotheses be generated independently. */
void orc_ransac_sample_persistent(uint64_t seed, int32_t H, int32_t N, int32_t S, int32_t *out)
{
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    for (int32_t i = 0; i < N; i++) idx[i] = i;
    for (int32_t h = 0; h < H; h++)
        for (int32_t i = 0; i < S; i++) {
            uint64_t x = orc_rng_draw(seed, (uint32_t)h, (uint32_t)i);
            int32_t j = i + (int32_t)(x % (uint64_t)(N - i));
            int32_t t = idx[i]; idx[i] = idx[j]; idx[j] = t;
            out[(size_t)h * S + i] = idx[i];
        }
    free(idx);
}

/* random linear form f0 = u0 + u1 s1 + u2 s2 + u3 s3, u_j uniform in (-100, 100) (Theia: 100*Vector4d::Random()) */
void orc_dls_linear_form(uint64_t seed, int32_t hyp, double u[4])
{
    for (int j = 0; j < 4; j++) {
        uint64_t x = orc_rng_draw(seed, (uint32_t)hyp, (uint32_t)(64 + j));
        double f = (double)(x >> 11) * (1.0 / 9007199254740992.0); /* [0,1) with 53 bits */
        u[j] = 100.0 * (2.0 * f - 1.0);
    }
}

/* ================================================================ Error (DlsPnpWithRansac.h:75-99) */
/* T is column-major 4x4 (Eigen Matrix4d).  b_X = R*a_X + t; b_X /= b_X(2); err = |x-u| + |y-v|.
 * Operation order fixed (no FMA): ((R00*X + R01*Y) + R02*Z) + tx.  Negative depth is NOT rejected (the reference
 * does not), z == 0 yields Inf/NaN which simply is not an inlier. */
double orc_reproj_error(const double *T, const double *X, const double *uv)
{
    double x = ((T[0] * X[0] + T[4] * X[1]) + T[8] * X[2]) + T[12];
    double y = ((T[1] * X[0] + T[5] * X[1]) + T[9] * X[2]) + T[13];
    double z = ((T[2] * X[0] + T[6] * X[1]) + T[10] * X[2]) + T[14];
    double xn = x / z, yn = y / z;
    FL(FS_SCORE, 18 + 2 + 3);
    return fabs(xn - uv[0]) + fabs(yn - uv[1]);
}

/* MLE (MSAC-style) cost + inliers of one model over all N points.  Fixed summation tree shared with the device:
 * lane L (0..63) accumulates points L, L+64, ... in order; then acc[L] += acc[L^m], m = 32..1. */
void orc_score_model(const double *T, const double *X, const double *uv, int32_t N, double thresh, int32_t use_mle,
                     double *cost, int32_t *n_inliers, uint8_t *mask)
{
    double acc[64];
    int32_t cnt = 0;
    for (int L = 0; L < 64; L++) acc[L] = 0.0;
    for (int32_t base = 0; base < N; base += 64)
        for (int L = 0; L < 64; L++) {
            int32_t i = base + L;
            if (i >= N) continue;
            double r = orc_reproj_error(T, X + 3 * i, uv + 2 * i);
            int in = r < thresh;
            if (mask) mask[i] = (uint8_t)in;
            cnt += in;
            acc[L] = acc[L] + (in ? r : thresh);
            FL(FS_SCORE, 1);
        }
    for (int m = 32; m >= 1; m >>= 1) {
        double nxt[64];
        for (int L = 0; L < 64; L++) nxt[L] = acc[L] + acc[L ^ m];
        FL(FS_SCORE, 1);     /* one wave-wide add per butterfly level (64 lanes, one useful result) */
        memcpy(acc, nxt, sizeof acc);
    }
    *n_inliers = cnt;
    *cost = use_mle ? acc[0] : (double)(N - cnt);
}

/* ================================================================ monomial bookkeeping for the DLS solver */
/* exponent triples (a,b,c) of s1^a s2^b s3^c */

/* index of (a,b,c) among monomials of degree <= 4 (35) / <= 3 (20): lexicographic enumeration */
static int idx_le(int a, int b, int c, int d)
{
    int k = 0;
    for (int x = 0; x <= d; x++)
        for (int y = 0; x + y <= d; y++)
            for (int z = 0; x + y + z <= d; z++) {
                if (x == a && y == b && z == c) return k;
                k++;
            }
    return -1;
}

/* Macaulay column/row position of monomial (a,b,c), degree <= 7:
 *   [0,27)   : a,b,c <= 2, index 9a+3b+c                      (the 27 "reduced" monomials, eigenvector basis)
 *   [27,93)  : the 66 remaining monomials that are NOT of the form {one exponent == 3, others <= 2}
 *   [93,120) : the 27 "boundary" monomials {exactly one exponent == 3, others <= 2}: the only rows of
 *              X = D^-1 C that B touches, so back-substitution stops after them.
 * Within each class the order is lexicographic in (a,b,c). */
static int is_reduced(int a, int b, int c) { return a <= 2 && b <= 2 && c <= 2; }
static int is_boundary(int a, int b, int c)
{
    return (a == 3 && b <= 2 && c <= 2) || (b == 3 && a <= 2 && c <= 2) || (c == 3 && a <= 2 && b <= 2);
}
void orc_dls_monomial_positions(int32_t pos[8][8][8])
{
    int n_mid = 27, n_bnd = 93;
    for (int a = 0; a <= 7; a++)
        for (int b = 0; b <= 7; b++)
            for (int c = 0; c <= 7; c++) pos[a][b][c] = -1;
    for (int a = 0; a <= 7; a++)
        for (int b = 0; a + b <= 7; b++)
            for (int c = 0; a + b + c <= 7; c++) {
                if (is_reduced(a, b, c)) pos[a][b][c] = 9 * a + 3 * b + c;
                else if (is_boundary(a, b, c)) pos[a][b][c] = n_bnd++;
                else pos[a][b][c] = n_mid++;
            }
}

/* vec(Rbar) (row-major r11 r12 r13 r21 ...) = Q * [1, s1, s2, s3, s1^2, s1s2, s1s3, s2^2, s2s3, s3^2]
 * Rbar(s) = (1 - s.s) I + 2 [s]x + 2 s s^T  (Cayley; R = Rbar / (1 + s.s)) */
static const double QMAT[9][10] = {
    /*        1   s1  s2  s3 s1s1 s1s2 s1s3 s2s2 s2s3 s3s3 */
    /*r11*/ { 1,  0,  0,  0,  1,   0,   0,  -1,   0,  -1 },
    /*r12*/ { 0,  0,  0, -2,  0,   2,   0,   0,   0,   0 },
    /*r13*/ { 0,  0,  2,  0,  0,   0,   2,   0,   0,   0 },
    /*r21*/ { 0,  0,  0,  2,  0,   2,   0,   0,   0,   0 },
    /*r22*/ { 1,  0,  0,  0, -1,   0,   0,   1,   0,  -1 },
    /*r23*/ { 0, -2,  0,  0,  0,   0,   0,   0,   2,   0 },
    /*r31*/ { 0,  0, -2,  0,  0,   0,   2,   0,   0,   0 },
    /*r32*/ { 0,  2,  0,  0,  0,   0,   0,   0,   2,   0 },
    /*r33*/ { 1,  0,  0,  0, -1,   0,   0,  -1,   0,   1 },
};
static const int M10[10][3] = { {0,0,0}, {1,0,0}, {0,1,0}, {0,0,1}, {2,0,0}, {1,1,0}, {1,0,1}, {0,2,0}, {0,1,1}, {0,0,2} };

/* ================================================================ DLS step 1-5: cost matrix and the three cubics */
/* Outputs: Tfac (3x9, t = Tfac * vec(R)), f[3][20] cubic coefficients indexed by idx_le(a,b,c,3). */
void orc_dls_cubics(const double *X, const double *uv, int32_t n, double Tfac[27], double f[3][20])
{
    double zb[64][3];
    double Szz[3][3] = {{0}}, W[3][9] = {{0}}, M9[9][9] = {{0}};
    if (n > 64) n = 64;
    for (int i = 0; i < n; i++) {
        double u = uv[2 * i], v = uv[2 * i + 1];
        double nrm = sqrt((u * u + v * v) + 1.0);
        zb[i][0] = u / nrm; zb[i][1] = v / nrm; zb[i][2] = 1.0 / nrm;
        FL(FS_CUBICS, 4 + 1 + 3);
    }
    /* H = (n I - sum z z^T)^-1 */
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Szz[a][b] = Szz[a][b] + zb[i][a] * zb[i][b];
    FL(FS_CUBICS, 18L * n + 9 /* n I - Szz */ + 27 /* cofactors */ + 5 /* det */ + 9 /* H */);
    double m[3][3], H[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) m[a][b] = (a == b ? (double)n : 0.0) - Szz[a][b];
    {
        double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
        double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
        double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
        double c10 = m[0][2] * m[2][1] - m[0][1] * m[2][2];
        double c11 = m[0][0] * m[2][2] - m[0][2] * m[2][0];
        double c12 = m[0][1] * m[2][0] - m[0][0] * m[2][1];
        double c20 = m[0][1] * m[1][2] - m[0][2] * m[1][1];
        double c21 = m[0][2] * m[1][0] - m[0][0] * m[1][2];
        double c22 = m[0][0] * m[1][1] - m[0][1] * m[1][0];
        double det = (m[0][0] * c00 + m[0][1] * c01) + m[0][2] * c02;
        H[0][0] = c00 / det; H[0][1] = c10 / det; H[0][2] = c20 / det;
        H[1][0] = c01 / det; H[1][1] = c11 / det; H[1][2] = c21 / det;
        H[2][0] = c02 / det; H[2][1] = c12 / det; H[2][2] = c22 / det;
    }
    /* W = sum (z z^T - I) L(p),  L(p) = blockdiag(p^T, p^T, p^T) (row-major vec(R)) */
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double e = zb[i][a] * zb[i][b] - (a == b ? 1.0 : 0.0);
                for (int c = 0; c < 3; c++) W[a][3 * b + c] = W[a][3 * b + c] + e * X[3 * i + c];
                FL(FS_CUBICS, 2 + 6);
            }
    for (int a = 0; a < 3; a++)
        for (int j = 0; j < 9; j++) { Tfac[9 * a + j] = (H[a][0] * W[0][j] + H[a][1] * W[1][j]) + H[a][2] * W[2][j]; FL(FS_CUBICS, 5); }
    /* M9 = sum (L+T)^T (I - z z^T) (L+T) */
    for (int i = 0; i < n; i++) {
        double A[3][9], B[3][9], P[3][3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                P[a][b] = (a == b ? 1.0 : 0.0) - zb[i][a] * zb[i][b];
                for (int c = 0; c < 3; c++) A[a][3 * b + c] = Tfac[9 * a + 3 * b + c] + (a == b ? X[3 * i + c] : 0.0);
            }
        for (int a = 0; a < 3; a++)
            for (int j = 0; j < 9; j++) B[a][j] = (P[a][0] * A[0][j] + P[a][1] * A[1][j]) + P[a][2] * A[2][j];
        for (int j = 0; j < 9; j++)
            for (int k = 0; k < 9; k++) M9[j][k] = M9[j][k] + ((A[0][j] * B[0][k] + A[1][j] * B[1][k]) + A[2][j] * B[2][k]);
        FL(FS_CUBICS, 9 * 2 /* P */ + 27 /* A */ + 27 * 5 /* B */ + 81 * 6 /* M9 */);
    }
    /* G = Q^T M9 Q;  J'(s) = m10^T G m10 (quartic) */
    double MQ[9][10], G[10][10];
    for (int j = 0; j < 9; j++)
        for (int q = 0; q < 10; q++) {
            double s = 0.0;
            for (int k = 0; k < 9; k++) s = s + M9[j][k] * QMAT[k][q];
            FL(FS_CUBICS, 18);
            MQ[j][q] = s;
        }
    for (int l = 0; l < 10; l++)
        for (int q = 0; q < 10; q++) {
            double s = 0.0;
            for (int j = 0; j < 9; j++) s = s + QMAT[j][l] * MQ[j][q];
            FL(FS_CUBICS, 18);
            G[l][q] = s;
        }
    double c4[35];
    for (int k = 0; k < 35; k++) c4[k] = 0.0;
    for (int l = 0; l < 10; l++)
        for (int q = 0; q < 10; q++) {
            int k = idx_le(M10[l][0] + M10[q][0], M10[l][1] + M10[q][1], M10[l][2] + M10[q][2], 4);
            c4[k] = c4[k] + G[l][q];
            FL(FS_CUBICS, 1);
        }
    /* f_k = dJ'/ds_k */
    for (int a = 0; a <= 3; a++)
        for (int b = 0; a + b <= 3; b++)
            for (int c = 0; a + b + c <= 3; c++) {
                int k = idx_le(a, b, c, 3);
                f[0][k] = (double)(a + 1) * c4[idx_le(a + 1, b, c, 4)];
                f[1][k] = (double)(b + 1) * c4[idx_le(a, b + 1, c, 4)];
                f[2][k] = (double)(c + 1) * c4[idx_le(a, b, c + 1, 4)];
                FL(FS_CUBICS, 3);
            }
}

/* ================================================================ DLS step 6-7: Macaulay matrix -> action matrix */
/* S (27x27 row-major) = A - B D^-1 C of the degree-7 Macaulay matrix of {f0; f1, f2, f3}.  Returns 0 on success,
 * -1 if a zero pivot is met (singular D). */
#ifdef ORC_LU_TIE_STATS
long orc_lu_tie_stats[4];   /* pivot columns seen, high-word ties, exact ties, largest elimination step with an exact tie */
#endif
int orc_dls_action_matrix(const double f[3][20], const double u[4], double S[27 * 27])
{
    static const int SH[4][3] = { {0,0,0}, {1,0,0}, {0,1,0}, {0,0,1} };  /* terms of f0: 1, s1, s2, s3 */
    int32_t pos[8][8][8];
    orc_dls_monomial_positions(pos);
    enum { NR = 93, NC = 120 };  /* [D | C]: columns 0..92 = D (Macaulay columns 27..119), 93..119 = C (columns 0..26) */
    double *E = (double *)calloc((size_t)NR * NC, sizeof(double));
    for (int a = 0; a <= 7; a++)
        for (int b = 0; a + b <= 7; b++)
            for (int c = 0; a + b + c <= 7; c++) {
                if (is_reduced(a, b, c)) continue;
                int row = pos[a][b][c] - 27;
                int which, ma = a, mb = b, mc = c;  /* multiplier monomial */
                if (a >= 3) { which = 0; ma -= 3; } else if (b >= 3) { which = 1; mb -= 3; } else { which = 2; mc -= 3; }
                for (int x = 0; x <= 3; x++)
                    for (int y = 0; x + y <= 3; y++)
                        for (int z = 0; x + y + z <= 3; z++) {
                            int col = pos[ma + x][mb + y][mc + z];
                            double v = f[which][idx_le(x, y, z, 3)];
                            if (col >= 27) E[row * NC + (col - 27)] = v; else E[row * NC + (93 + col)] = v;
                        }
            }
#ifdef ORC_FLOP_COUNT
    unsigned char allz[93];
#endif
    /* forward elimination with partial (row) pivoting; ties -> smallest row index */
    for (int k = 0; k < NR; k++) {
#ifdef ORC_FLOP_COUNT
        if (k % 4 == 0) memset(allz, 1, sizeof allz);
#endif
        int p = k;
        double best = fabs(E[k * NC + k]);
        for (int i = k + 1; i < NR; i++) {
            double v = fabs(E[i * NC + k]);
            if (v > best) { best = v; p = i; }
        }
#ifdef ORC_LU_TIE_STATS
        {   /* test infrastructure (oracle/_build/liboracle_stats.so only): does this pivot column reach the tie paths of the GPU's pivot
             * search?  `high` = two candidates agree in the high word of their magnitude (or the maximum is 0 / not finite): the
             * 64-bit compare path; `exact` = two candidates attain the maximum exactly: the smallest-logical-index rule. */
            unsigned int mh = 0; int ch = 0, ce = 0; double bb = -1.0;
            for (int i = k; i < NR; i++) {
                const double v = fabs(E[i * NC + k]);
                unsigned long long b; memcpy(&b, &v, 8);
                const unsigned int h = (unsigned int)(b >> 32);
                if (h > mh) { mh = h; ch = 1; } else if (h == mh) ch++;
                if (v > bb) { bb = v; ce = 1; } else if (v == bb) ce++;
            }
            __sync_fetch_and_add(&orc_lu_tie_stats[0], 1);
            if (ch != 1 || mh == 0 || mh >= 0x7ff00000u) __sync_fetch_and_add(&orc_lu_tie_stats[1], 1);
            if (ce > 1) { __sync_fetch_and_add(&orc_lu_tie_stats[2], 1); if (k > orc_lu_tie_stats[3]) orc_lu_tie_stats[3] = k; }
        }
#endif
        if (!(best > 0.0)) { free(E); return -1; }
        if (p != k)
            for (int j = 0; j < NC; j++) { double t = E[k * NC + j]; E[k * NC + j] = E[p * NC + j]; E[p * NC + j] = t; }
#ifdef ORC_FLOP_COUNT
        if (p != k) { unsigned char t = allz[k]; allz[k] = allz[p]; allz[p] = t; }
#endif
        double piv = E[k * NC + k];
        for (int i = k + 1; i < NR; i++) {
            double l = E[i * NC + k] / piv;
            E[i * NC + k] = l;
            FL(FS_LU, 1);
#ifdef ORC_FLOP_COUNT
            orc_flop_dense_lu += 1 + 2L * (NC - k - 1);
            if (l != 0.0) allz[i] = 0;
            if (k % 4 == 3 || k == NR - 1) { orc_panel_rows_total++; orc_panel_rows_allzero += allz[i]; }
#endif
            if (l != 0.0) {
                for (int j = k + 1; j < NC; j++) E[i * NC + j] = E[i * NC + j] - l * E[k * NC + j];
                FL(FS_LU, 2L * (NC - k - 1));
            }
        }
    }
    /* back-substitution for the last 27 unknowns only (boundary monomials), all 27 right-hand sides */
    double Xb[27][27]; /* Xb[t][c] = X[66+t][c] */
    for (int c = 0; c < 27; c++)
        for (int i = NR - 1; i >= NR - 27; i--) {
            double s = E[i * NC + 93 + c];
            for (int j = i + 1; j < NR; j++) s = s - E[i * NC + j] * Xb[j - 66][c];
            Xb[i - 66][c] = s / E[i * NC + i];
            FL(FS_ACTION, 2L * (NR - 1 - i) + 1);
        }
    free(E);
    /* S = A - B X : row m (reduced monomial) is m * f0 = u0 m + u1 m s1 + u2 m s2 + u3 m s3 */
    for (int a = 0; a <= 2; a++)
        for (int b = 0; b <= 2; b++)
            for (int c = 0; c <= 2; c++) {
                int r = 9 * a + 3 * b + c;
                double *Sr = S + 27 * r;
                for (int j = 0; j < 27; j++) Sr[j] = 0.0;
                for (int t = 0; t < 4; t++) {
                    int col = pos[a + SH[t][0]][b + SH[t][1]][c + SH[t][2]];
                    if (col < 27) { Sr[col] = Sr[col] + u[t]; FL(FS_ACTION, 1); }
                }
                for (int t = 1; t < 4; t++) {
                    int col = pos[a + SH[t][0]][b + SH[t][1]][c + SH[t][2]];
                    if (col >= 27) /* boundary monomial: position 93.. -> Xb row col-93 */
                    {
                        for (int j = 0; j < 27; j++) Sr[j] = Sr[j] - u[t] * Xb[col - 93][j];
                        FL(FS_ACTION, 54);
                    }
                }
            }
    return 0;
}

/* ================================================================ 27x27 real eigenproblem (real eigenpairs only) */
#define EN 27
/* diagnostics (tests / tuning only): QR sweeps and k-steps of the last orc_eig27_real call on this thread */
static __thread long dbg_sweeps, dbg_ksteps;
void orc_eig27_last_stats(long *sweeps, long *ksteps) { *sweeps = dbg_sweeps; *ksteps = dbg_ksteps; }
/* Householder reduction to upper Hessenberg form with accumulated transformations (orthes + ortran). */
static void hessenberg(double H[EN][EN], double V[EN][EN])
{
    double ort[EN], ortm[EN];
    const int low = 0, high = EN - 1;
    for (int m = low + 1; m <= high - 1; m++) {
        double scale = 0.0;
        for (int i = m; i <= high; i++) scale = scale + fabs(H[i][m - 1]);
        FL(FS_HESS, high - m + 1);
        ort[m] = 0.0;
        if (scale != 0.0) {
            double h = 0.0;
            for (int i = high; i >= m; i--) { ort[i] = H[i][m - 1] / scale; h = h + ort[i] * ort[i]; }
            FL(FS_HESS, 3L * (high - m + 1) + 1 /* sqrt */ + 3 /* h, ort[m] */ + 2 /* rescale */);
            double g = sqrt(h);
            if (ort[m] > 0) g = -g;
            h = h - ort[m] * g;
            ort[m] = ort[m] - g;
            for (int j = m; j < EN; j++) { /* H = (I - u u^T/h) H */
                double f = 0.0;
                for (int i = high; i >= m; i--) f = f + ort[i] * H[i][j];
                f = f / h;
                for (int i = m; i <= high; i++) H[i][j] = H[i][j] - f * ort[i];
                FL(FS_HESS, 4L * (high - m + 1) + 1);
            }
            for (int i = 0; i <= high; i++) { /* H = H (I - u u^T/h) */
                double f = 0.0;
                for (int j = high; j >= m; j--) f = f + ort[j] * H[i][j];
                f = f / h;
                for (int j = m; j <= high; j++) H[i][j] = H[i][j] - f * ort[j];
                FL(FS_HESS, 4L * (high - m + 1) + 1);
            }
            ort[m] = scale * ort[m];
            H[m][m - 1] = scale * g;
        }
        ortm[m] = ort[m]; /* head of reflector m (its tail stays in H[m+1..][m-1]) for the accumulation pass */
    }
    /* accumulate: V = product of the reflectors */
    for (int i = 0; i < EN; i++)
        for (int j = 0; j < EN; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int m = high - 1; m >= low + 1; m--) {
        if (H[m][m - 1] != 0.0) {
            double o[EN];
            o[m] = ortm[m];
            for (int i = m + 1; i <= high; i++) o[i] = H[i][m - 1];
            for (int j = m; j <= high; j++) {
                double g = 0.0;
                for (int i = m; i <= high; i++) g = g + o[i] * V[i][j];
                g = (g / o[m]) / H[m][m - 1]; /* double division avoids underflow */
                for (int i = m; i <= high; i++) V[i][j] = V[i][j] + g * o[i];
                FL(FS_HESS, 4L * (high - m + 1) + 2);
            }
        }
    }
    for (int i = 2; i < EN; i++)
        for (int j = 0; j < i - 1; j++) H[i][j] = 0.0;
}

/* Francis double-shift QR on the Hessenberg matrix with accumulation into V, then back-substitution for the REAL
 * eigenvalues only.  wr/wi: eigenvalues; H becomes quasi-triangular and its column n holds the (triangular-form)
 * eigenvector of real eigenvalue n.  Returns 0, or -1 if an eigenvalue fails to converge in 60 sweeps. */
static int francis_qr(double H[EN][EN], double V[EN][EN], double wr[EN], double wi[EN])
{
    const int nn = EN, low = 0, high = EN - 1;
    const double eps = DBL_EPSILON; /* 2^-52 */
    int n = nn - 1;
    double exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, t, w, x, y;
    double norm = 0.0;
    for (int i = 0; i < nn; i++)
        for (int j = (i - 1 > 0 ? i - 1 : 0); j < nn; j++) { norm = norm + fabs(H[i][j]); FL(FS_QR, 1); }
    int iter = 0;
    while (n >= low) {
        int l = n;
        while (l > low) {
            s = fabs(H[l - 1][l - 1]) + fabs(H[l][l]);
            if (s == 0.0) s = norm;
            FL(FS_QR, 2);
            if (fabs(H[l][l - 1]) < eps * s) break;
            l--;
        }
        if (l == n) { /* one root */
            H[n][n] = H[n][n] + exshift;
            FL(FS_QR, 1);
            wr[n] = H[n][n]; wi[n] = 0.0;
            n--; iter = 0;
        } else if (l == n - 1) { /* two roots */
            w = H[n][n - 1] * H[n - 1][n];
            p = (H[n - 1][n - 1] - H[n][n]) / 2.0;
            q = p * p + w;
            z = sqrt(fabs(q));
            H[n][n] = H[n][n] + exshift;
            H[n - 1][n - 1] = H[n - 1][n - 1] + exshift;
            x = H[n][n];
            FL(FS_QR, 1 + 2 + 2 + 1 + 2);
            if (q >= 0) { /* real pair */
                z = (p >= 0) ? p + z : p - z;
                wr[n - 1] = x + z;
                wr[n] = wr[n - 1];
                if (z != 0.0) wr[n] = x - w / z;
                wi[n - 1] = 0.0; wi[n] = 0.0;
                x = H[n][n - 1];
                s = fabs(x) + fabs(z);
                p = x / s; q = z / s;
                r = sqrt(p * p + q * q);
                p = p / r; q = q / r;
                for (int j = n - 1; j < nn; j++) { z = H[n - 1][j]; H[n - 1][j] = q * z + p * H[n][j]; H[n][j] = q * H[n][j] - p * z; }
                for (int i = 0; i <= n; i++) { z = H[i][n - 1]; H[i][n - 1] = q * z + p * H[i][n]; H[i][n] = q * H[i][n] - p * z; }
                for (int i = low; i <= high; i++) { z = V[i][n - 1]; V[i][n - 1] = q * z + p * V[i][n]; V[i][n] = q * V[i][n] - p * z; }
                FL(FS_QR, 1 + 1 + 2 + 1 + 2 + 4 + 2 + 6L * (nn - (n - 1)) + 6L * (n + 1) + 6L * (high - low + 1));
            } else { /* complex pair */
                FL(FS_QR, 2);
                wr[n - 1] = x + p; wr[n] = x + p;
                wi[n - 1] = z; wi[n] = -z;
            }
            n -= 2; iter = 0;
        } else {
            x = H[n][n]; y = 0.0; w = 0.0;
            if (l < n) { y = H[n - 1][n - 1]; w = H[n][n - 1] * H[n - 1][n]; FL(FS_QR, 1); }
            if (iter == 10) { /* Wilkinson's exceptional shift */
                exshift = exshift + x;
                for (int i = low; i <= n; i++) H[i][i] = H[i][i] - x;
                s = fabs(H[n][n - 1]) + fabs(H[n - 1][n - 2]);
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
                FL(FS_QR, 1 + (n - low + 1) + 1 + 1 + 2);
            }
            if (iter == 30) { /* second exceptional shift */
                s = (y - x) / 2.0;
                s = s * s + w;
                FL(FS_QR, 4);
                if (s > 0) {
                    FL(FS_QR, 1 + 5 + (n - low + 1) + 1);
                    s = sqrt(s);
                    if (y < x) s = -s;
                    s = x - w / ((y - x) / 2.0 + s);
                    for (int i = low; i <= n; i++) H[i][i] = H[i][i] - s;
                    exshift = exshift + s;
                    x = y = w = 0.964;
                }
            }
            iter++;
            dbg_sweeps++;
            if (iter > 60) return -1;
            int m = n - 2;
            while (m >= l) { /* two consecutive small sub-diagonal elements */
                z = H[m][m];
                r = x - z; s = y - z;
                p = (r * s - w) / H[m + 1][m] + H[m][m + 1];
                q = H[m + 1][m + 1] - z - r - s;
                r = H[m + 2][m + 1];
                FL(FS_QR, 2 + 4 + 3 + 2 /* |p|+|q|+|r| */);
                /* EISPACK divides (p,q,r) by |p|+|q|+|r| purely as overflow protection; here the scale is the power of two
                 * 2^e with |p|+|q|+|r| = f * 2^e, f in [0.5,1): the scaling is then exact (no rounding) and costs three
                 * exponent adjustments instead of three divisions on the critical path of every step. */
                s = fabs(p) + fabs(q) + fabs(r);
#ifdef ORC_EISPACK_DIVIDE   /* EISPACK hqr2 / Eigen::EigenSolver as written: kept buildable to show the two select alike */
                p = p / s; q = q / s; r = r / s;
#else
                { int e2; (void)frexp(s, &e2); p = ldexp(p, -e2); q = ldexp(q, -e2); r = ldexp(r, -e2); }
#endif
                if (m == l) break;
                FL(FS_QR, 2 + 4);
                if (fabs(H[m][m - 1]) * (fabs(q) + fabs(r)) <
                    eps * (fabs(p) * (fabs(H[m - 1][m - 1]) + fabs(z) + fabs(H[m + 1][m + 1])))) break;
                m--;
            }
            for (int i = m + 2; i <= n; i++) { H[i][i - 2] = 0.0; if (i > m + 2) H[i][i - 3] = 0.0; }
            for (int k = m; k <= n - 1; k++) { /* double QR step on rows l..n, columns m..n */
                int notlast = (k != n - 1);
                int ex = 0;
                (void)ex;
                dbg_ksteps++;
                if (k != m) {
                    p = H[k][k - 1]; q = H[k + 1][k - 1]; r = notlast ? H[k + 2][k - 1] : 0.0;
                    x = fabs(p) + fabs(q) + fabs(r);
                    FL(FS_QR, 2);
                    if (x == 0.0) continue;
#ifdef ORC_EISPACK_DIVIDE
                    p = p / x; q = q / x; r = r / x;
#else
                    (void)frexp(x, &ex);                 /* exact power-of-two scaling, see above */
                    p = ldexp(p, -ex); q = ldexp(q, -ex); r = ldexp(r, -ex);
#endif
                }
                s = sqrt(p * p + q * q + r * r);
                FL(FS_QR, 6);
                if (p < 0) s = -s;
                if (s != 0.0) {
                    FL(FS_QR, 1 + 3 + 2);
                    FL(FS_QR, (notlast ? 10L : 6L) * (nn - k) + (notlast ? 10L : 6L) * (((n < k + 3) ? n : k + 3) + 1) + (notlast ? 10L : 6L) * (high - low + 1));
#ifdef ORC_EISPACK_DIVIDE
                    if (k != m) H[k][k - 1] = -s * x;
#else
                    if (k != m) H[k][k - 1] = ldexp(-s, ex);
#endif
                    else if (l != m) H[k][k - 1] = -H[k][k - 1];
                    p = p + s;
                    x = p / s; y = q / s; z = r / s;
                    q = q / p; r = r / p;
                    for (int j = k; j < nn; j++) { /* rows */
                        double pp = H[k][j] + q * H[k + 1][j];
                        if (notlast) { pp = pp + r * H[k + 2][j]; H[k + 2][j] = H[k + 2][j] - pp * z; }
                        H[k][j] = H[k][j] - pp * x;
                        H[k + 1][j] = H[k + 1][j] - pp * y;
                    }
                    int imax = (n < k + 3) ? n : k + 3;
                    for (int i = 0; i <= imax; i++) { /* columns */
                        double pp = x * H[i][k] + y * H[i][k + 1];
                        if (notlast) { pp = pp + z * H[i][k + 2]; H[i][k + 2] = H[i][k + 2] - pp * r; }
                        H[i][k] = H[i][k] - pp;
                        H[i][k + 1] = H[i][k + 1] - pp * q;
                    }
                    for (int i = low; i <= high; i++) { /* accumulate */
                        double pp = x * V[i][k] + y * V[i][k + 1];
                        if (notlast) { pp = pp + z * V[i][k + 2]; V[i][k + 2] = V[i][k + 2] - pp * r; }
                        V[i][k] = V[i][k] - pp;
                        V[i][k + 1] = V[i][k + 1] - pp * q;
                    }
                }
            }
        }
    }
    if (norm == 0.0) return 0;
    /* back-substitution, REAL eigenvalues only (a complex pair cannot produce the real Cayley root we accept) */
    for (n = nn - 1; n >= 0; n--) {
        if (wi[n] != 0.0) continue;
        p = wr[n];
        int l = n;
        H[n][n] = 1.0;
        for (int i = n - 1; i >= 0; i--) {
            w = H[i][i] - p;
            r = 0.0;
            for (int j = l; j <= n; j++) r = r + H[i][j] * H[j][n];
            FL(FS_EIGVEC, 1 + 2L * (n - l + 1));
            if (wi[i] < 0.0) { z = w; s = r; }
            else {
                l = i;
                if (wi[i] == 0.0) {
                    if (w != 0.0) H[i][n] = -r / w; else H[i][n] = -r / (eps * norm);
                    FL(FS_EIGVEC, 1);
                } else { /* 2x2 block of a complex pair above a real eigenvalue */
                    x = H[i][i + 1]; y = H[i + 1][i];
                    q = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i];
                    t = (x * s - z * r) / q;
                    H[i][n] = t;
                    if (fabs(x) > fabs(z)) H[i + 1][n] = (-r - w * t) / x; else H[i + 1][n] = (-s - y * t) / z;
                    FL(FS_EIGVEC, 5 + 4 + 3);
                }
                t = fabs(H[i][n]);
                FL(FS_EIGVEC, 2);
                if ((eps * t) * t > 1) { for (int j = i; j <= n; j++) H[j][n] = H[j][n] / t; FL(FS_EIGVEC, n - i + 1); }
            }
        }
    }
    return 0;
}

/* Real eigenpairs of the 27x27 action matrix.  For every real eigenvalue (in Schur-form index order 0..26) returns
 * lambda and the four eigenvector entries v[0], v[1], v[3], v[9] (monomials 1, s3, s2, s1).  Returns the count,
 * or -1 on non-convergence. */
int orc_eig27_real(const double S[27 * 27], double lambda[27], double v4[27][4])
{
    static const int ROWS[4] = {0, 1, 3, 9};
    double H[EN][EN], V[EN][EN], wr[EN], wi[EN];
    for (int i = 0; i < EN; i++)
        for (int j = 0; j < EN; j++) H[i][j] = S[EN * i + j];
    memset(V, 0, sizeof V);
    dbg_sweeps = dbg_ksteps = 0;
    hessenberg(H, V);
    if (francis_qr(H, V, wr, wi) != 0) return -1;
    int cnt = 0;
    for (int n = 0; n < EN; n++) {
        if (wi[n] != 0.0) continue;
        lambda[cnt] = wr[n];
        for (int rr = 0; rr < 4; rr++) { /* back-transform only the rows we need: v = V * H[:, n] */
            int i = ROWS[rr];
            double zz = 0.0;
            for (int k = 0; k <= n; k++) zz = zz + V[i][k] * H[k][n];
            FL(FS_EIGVEC, 2L * (n + 1));
            v4[cnt][rr] = zz;
        }
        cnt++;
    }
    return cnt;
}

/* ================================================================ DLS-PnP (theia::DlsPnp restated) */
/* Quaternion (w,x,y,z) -> rotation matrix (row-major R[9]), Eigen::Quaternion::toRotationMatrix formula. */
static void quat_to_rot(double w, double x, double y, double z, double R[9])
{
    double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* All cheirality-valid real solutions of the DLS system for n correspondences.  Rs: row-major 3x3 each, ts: 3 each.
 * Returns the number of solutions found (may exceed max_out; only the first max_out are stored), or a negative
 * value on solver failure (singular D: -1, eigen non-convergence: -2). */
int orc_dls_pnp(const double *X, const double *uv, int32_t n, const double u[4], double *Rs, double *ts, int32_t max_out)
{
    double Tfac[27], f[3][20], S[27 * 27], lambda[27], v4[27][4];
    orc_dls_cubics(X, uv, n, Tfac, f);
    if (orc_dls_action_matrix(f, u, S) != 0) return -1;
    int nreal = orc_eig27_real(S, lambda, v4);
    if (nreal < 0) return -2;
    int cnt = 0;
    for (int e = 0; e < nreal; e++) {
        double s1 = v4[e][3] / v4[e][0], s2 = v4[e][2] / v4[e][0], s3 = v4[e][1] / v4[e][0];
        if (!(fabs(s1) <= DBL_MAX && fabs(s2) <= DBL_MAX && fabs(s3) <= DBL_MAX)) continue; /* inf / NaN root */
        FL(FS_POSE, 3 + 7 + 4 + 25 /* quat_to_rot: 3 + 9 + 13 */ + 3 * 18);
        double nq = sqrt(((1.0 + s1 * s1) + s2 * s2) + s3 * s3);
        double R[9], t[3];
        quat_to_rot(1.0 / nq, s1 / nq, s2 / nq, s3 / nq, R);
        for (int a = 0; a < 3; a++) {
            double s = 0.0;
            for (int j = 0; j < 9; j++) s = s + Tfac[9 * a + j] * R[j];
            t[a] = s;
        }
        int front = 1;
        for (int i = 0; i < n; i++) { /* all sample points in front of the camera */
            double zc = ((R[6] * X[3 * i] + R[7] * X[3 * i + 1]) + R[8] * X[3 * i + 2]) + t[2];
            FL(FS_POSE, 6);
            if (zc < 0) { front = 0; break; }
        }
        if (!front) continue;
        if (cnt < max_out) { memcpy(Rs + 9 * cnt, R, sizeof R); memcpy(ts + 3 * cnt, t, sizeof t); }
        cnt++;
    }
    return cnt;
}

/* ================================================================ hypothesis = sample + EstimateModel */
/* DlsPnpWithRansac::EstimateModel (DlsPnpWithRansac.h:48-72): accept iff exactly one solution; the accepted model is
 * b_T_a (column-major 4x4).  Returns 1 if a model was produced, else 0.  (The reference's success branch falls off a
 * non-void function, :62-68; its own test copy has `return true`, unittest_theia.cpp:84 -- restated as true.) */
int orc_pnp_hypothesis_sampled(const double *X, const double *uv, int32_t N, uint64_t seed, int32_t hyp, int32_t S,
                               const int32_t *sample_in, double T[16])
{
    int32_t sample[64];
    double sx[64 * 3], suv[64 * 2], u[4], R[9], t[3];
    if (S > 64) S = 64;
    if (sample_in) memcpy(sample, sample_in, sizeof(int32_t) * (size_t)S);
    else orc_ransac_sample(seed, hyp, N, S, sample);
    for (int i = 0; i < S; i++) {
        memcpy(sx + 3 * i, X + 3 * sample[i], 3 * sizeof(double));
        memcpy(suv + 2 * i, uv + 2 * sample[i], 2 * sizeof(double));
    }
    orc_dls_linear_form(seed, hyp, u);
    int ns = orc_dls_pnp(sx, suv, S, u, R, t, 1);
    if (ns != 1) return 0;
    T[0] = R[0]; T[1] = R[3]; T[2] = R[6]; T[3] = 0.0;
    T[4] = R[1]; T[5] = R[4]; T[6] = R[7]; T[7] = 0.0;
    T[8] = R[2]; T[9] = R[5]; T[10] = R[8]; T[11] = 0.0;
    T[12] = t[0]; T[13] = t[1]; T[14] = t[2]; T[15] = 1.0;
    return 1;
}

int orc_pnp_hypothesis(const double *X, const double *uv, int32_t N, uint64_t seed, int32_t hyp, int32_t S,
                       double T[16], int32_t *sample_out)
{
    int32_t sample[64];
    if (S > 64) S = 64;
    orc_ransac_sample(seed, hyp, N, S, sample);
    if (sample_out) memcpy(sample_out, sample, sizeof(int32_t) * (size_t)S);
    return orc_pnp_hypothesis_sampled(X, uv, N, seed, hyp, S, sample, T);
}

/* ================================================================ RANSAC driver (theia::Ransac::Estimate) */
void orc_ransac_params_default(orc_ransac_params *p)
{
    p->error_thresh = 0.03;        /* DlsPnpWithRansac.cpp:208 */
    p->min_inlier_ratio = 0.7;     /* :209 */
    p->max_iterations = 50;        /* :210 */
    p->min_iterations = 5;         /* :211 */
    p->use_mle = 1;                /* :212 */
    p->sample_size = 15;           /* DlsPnpWithRansac.h:45 */
    p->failure_probability = 0.01; /* theia::RansacParameters default */
    p->seed = 0x5EEDCE7EB80ULL;
    p->n_hypotheses = 0;
    p->sampler = 0;
}

int32_t orc_ransac_max_iterations(int32_t S, double ratio, double log_fail, int32_t min_it, int32_t max_it)
{
    if (ratio == 1.0) return min_it;
    double log_prob = log(1.0 - pow(ratio, (double)S)) - DBL_EPSILON;
    double itf = floor(log_fail / log_prob) + 1.0;
    int32_t it = (itf > 2.0e9) ? 2000000000 : (int32_t)itf;
    if (it < min_it) it = min_it;
    if (it > max_it) it = max_it;
    return it;
}

/* StaticTheiaPoseCompute::PNP (DlsPnpWithRansac.cpp:132-245).  Returns 0, or -9 for N < 20 (:136-139).
 * No model found: confidence 0, T = NaN (the reference leaves best_rel_pose uninitialised, :204). */
int orc_pnp_ransac(const double *X, const double *uv, int32_t N, const orc_ransac_params *p,
                   double T[16], float *confidence, uint8_t *mask, orc_ransac_summary *summary)
{
    if (N < 20) return -9;
    const int32_t S = p->sample_size;
    const double log_fail = log(p->failure_probability);
    double best_cost = DBL_MAX, best_T[16];
    int32_t best_h = -1, n_models = 0, num_it = 0;
    int32_t max_it;
    const int bench = p->n_hypotheses > 0;
    if (bench) max_it = p->n_hypotheses;
    else {
        max_it = p->max_iterations;
        if (p->min_inlier_ratio > 0)
            max_it = orc_ransac_max_iterations(S, p->min_inlier_ratio, log_fail, p->min_iterations, p->max_iterations);
    }
    int32_t *table = NULL;   /* sampler mode 1: the samples of hypotheses 0..max_it-1 of the ONE persistent permutation */
    if (p->sampler == 1) {
        table = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_it * (size_t)S);
        orc_ransac_sample_persistent(p->seed, max_it, N, S, table);
    }
    for (num_it = 0; num_it < max_it; num_it++) {
        double Th[16], cost;
        int32_t nin;
        if (!orc_pnp_hypothesis_sampled(X, uv, N, p->seed, num_it, S, table ? table + (size_t)num_it * S : NULL, Th)) continue;
        n_models++;
        orc_score_model(Th, X, uv, N, p->error_thresh, p->use_mle, &cost, &nin, NULL);
        if (cost < best_cost) { /* strict: first best wins */
            best_cost = cost; best_h = num_it;
            memcpy(best_T, Th, sizeof Th);
            if (!bench) {
                double ratio = (double)nin / (double)N;
                if (ratio < (double)S / (double)N) continue;
                int32_t mi = orc_ransac_max_iterations(S, ratio, log_fail, p->min_iterations, p->max_iterations);
                if (mi < max_it) max_it = mi;
            }
        }
    }
    free(table);
    int32_t nin = 0;
    if (best_h >= 0) {
        double c;
        memcpy(T, best_T, sizeof best_T);
        orc_score_model(best_T, X, uv, N, p->error_thresh, p->use_mle, &c, &nin, mask);
        double ratio = (double)nin / (double)N;
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

/* Benchmark-mode throughput of the same hypotheses on `nthreads` host threads (OpenMP over the independent hypotheses):
 * bench.py's all-cores CPU figure for the PnP leg (SURVEY.md 8d "RANSAC CPU: oracle, single thread and all cores over
 * hypotheses").  Returns the index of the first hypothesis with the smallest cost (the benchmark-mode winner), so the
 * caller can check it against orc_pnp_ransac. */
int32_t orc_pnp_hypotheses_mt(const double *X, const double *uv, int32_t N, const orc_ransac_params *p, int32_t H, int32_t nthreads,
                              int32_t *n_models_out)
{
    double *cost = (double *)malloc(sizeof(double) * (size_t)H);
    if (!cost) return -1;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int32_t h = 0; h < H; h++) {
        double Th[16], c;
        int32_t nin;
        cost[h] = INFINITY;
        if (!orc_pnp_hypothesis(X, uv, N, p->seed, h, p->sample_size, Th, NULL)) continue;
        orc_score_model(Th, X, uv, N, p->error_thresh, p->use_mle, &c, &nin, NULL);
        cost[h] = c;
    }
    int32_t best = -1, nm = 0;
    double bc = DBL_MAX;
    for (int32_t h = 0; h < H; h++) {
        if (cost[h] != INFINITY) nm++;
        if (cost[h] < bc) { bc = cost[h]; best = h; }
    }
    if (n_models_out) *n_models_out = nm;
    free(cost);
    return best;
}
