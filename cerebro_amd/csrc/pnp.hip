// pnp.hip -- batched DLS-PnP-in-RANSAC on gfx950 (replaces the body of StaticTheiaPoseCompute::PNP,
// /root/reference/src/DlsPnpWithRansac.cpp:192-240, i.e. theia::Ransac over the DlsPnpWithRansac estimator of
// src/DlsPnpWithRansac.h:42-100).  All hypotheses of a call are generated and scored in parallel:
//
//   K4+K5a  pnp_build_solve   one 448-thread workgroup per hypothesis (LDS ~50 KB, 128 VGPRs: 2 per CU): wave-parallel
//                             counter-based sampler, DLS cost matrix -> three Cayley cubics -> degree-7 Macaulay matrix
//                             [D|C] (93x120) held in the REGISTERS of waves 0..5 (31 rows x 1 column per thread) ->
//                             blocked LU with partial pivoting and lookahead: wave 6 factorises the next 4-column panel
//                             (DPP wave max + logical-position tie rule, readlane broadcasts) while waves 0..5 apply the
//                             previous panel; only panel columns / pivot rows / multipliers cross LDS
//                             -> 27x27 action matrix S = A - B D^-1 C.
//   K5b+K6  pnp_eig_score     one WAVE per hypothesis: Householder Hessenberg + Francis double-shift QR with
//                             accumulated transformations (matrices in LDS, lanes over independent rows/columns),
//                             lane-per-eigenvector back-substitution, Cayley roots -> R,t, cheirality, accept iff
//                             exactly one solution (DlsPnpWithRansac.h:62); then L1 reprojection error over all N
//                             correspondences (DlsPnpWithRansac.h:75-99), MLE cost, __ballot inlier words.
//   K7      host              argmin / theia's sequential early-termination rule replayed over the H results.
//
// Numerics: fp64, -ffp-contract=off, IEEE divide/sqrt; every sum runs in the order DESIGN.md 5 fixes (the same
// order as the CPU oracle), lanes only parallelise over independent outputs, so poses and inlier masks are
// reproducible bit for bit.  Neither HBM nor MFMA bound: 0.865 M counted fp64 operations per hypothesis (profiles/pnp_flops.json;
// 1.30 M with the dense elimination the device runs) = 1.5 % of the fp64-vector peak single-problem, 2.7 % batched; the vector pipe
// is 31 % (pnp_build_solve) / 63 % (pnp_eig_score) busy in the batched call (profiles/pnp_pmc.json): dependent chains of lone waves
// (DESIGN.md 5).  Up to kPnpMaxBatch problems share one pair of launches.
#include "ransac_common.h"
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <type_traits>
#include <vector>

namespace chip {

// ------------------------------------------------------------------------------------------------ tables
struct PnpTables {
    double qmat[9][10];       // vec(Rbar) = Q * [1 s1 s2 s3 s1^2 s1s2 s1s3 s2^2 s2s3 s3^2]
    int8_t pair2c4[100];      // quartic monomial index of m10[l]*m10[q]
    int8_t c4_cnt[35];        // inverse of pair2c4: how many (l,q) pairs feed quartic monomial k ...
    int8_t c4_src[35][16];    // ... and which ones, ascending (the order the sum is taken in)
    int8_t fsrc[3][20];       // f_k[mono] = fmul * c4[fsrc]
    int8_t fmul[3][20];
    int8_t row_which[93];     // Macaulay row (non-reduced monomial) -> which cubic multiplies it
    int16_t row_dst[93][20];  // destination column in E = [D | C] of each of the 20 terms
    int16_t s_col[27][4];     // Macaulay column of m * {1, s1, s2, s3} for reduced monomial m
    int8_t col_term[93][128]; // inverse of row_dst: which of the row's 20 terms lands in column c (-1: structural zero)
};

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

static void build_tables(PnpTables &t)
{
    static const double Q[9][10] = {
        {1, 0, 0, 0, 1, 0, 0, -1, 0, -1}, {0, 0, 0, -2, 0, 2, 0, 0, 0, 0}, {0, 0, 2, 0, 0, 0, 2, 0, 0, 0},
        {0, 0, 0, 2, 0, 2, 0, 0, 0, 0},   {1, 0, 0, 0, -1, 0, 0, 1, 0, -1}, {0, -2, 0, 0, 0, 0, 0, 0, 2, 0},
        {0, 0, -2, 0, 0, 0, 2, 0, 0, 0},  {0, 2, 0, 0, 0, 0, 0, 0, 2, 0},  {1, 0, 0, 0, -1, 0, 0, -1, 0, 1}};
    static const int M10[10][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {2, 0, 0}, {1, 1, 0}, {1, 0, 1}, {0, 2, 0}, {0, 1, 1}, {0, 0, 2}};
    static const int SH[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    std::memcpy(t.qmat, Q, sizeof Q);
    for (int l = 0; l < 10; l++)
        for (int q = 0; q < 10; q++)
            t.pair2c4[l * 10 + q] = (int8_t)idx_le(M10[l][0] + M10[q][0], M10[l][1] + M10[q][1], M10[l][2] + M10[q][2], 4);
    for (int k = 0; k < 35; k++) {
        t.c4_cnt[k] = 0;
        for (int e = 0; e < 100; e++)
            if (t.pair2c4[e] == k && t.c4_cnt[k] < 16) t.c4_src[k][t.c4_cnt[k]++] = (int8_t)e;
    }
    for (int a = 0; a <= 3; a++)
        for (int b = 0; a + b <= 3; b++)
            for (int c = 0; a + b + c <= 3; c++) {
                const int k = idx_le(a, b, c, 3);
                t.fsrc[0][k] = (int8_t)idx_le(a + 1, b, c, 4); t.fmul[0][k] = (int8_t)(a + 1);
                t.fsrc[1][k] = (int8_t)idx_le(a, b + 1, c, 4); t.fmul[1][k] = (int8_t)(b + 1);
                t.fsrc[2][k] = (int8_t)idx_le(a, b, c + 1, 4); t.fmul[2][k] = (int8_t)(c + 1);
            }
    // Macaulay positions: [0,27) reduced (9a+3b+c), [27,93) others, [93,120) boundary {one exponent == 3, rest <= 2}
    static int pos[8][8][8];
    int n_mid = 27, n_bnd = 93;
    for (int a = 0; a <= 7; a++)
        for (int b = 0; a + b <= 7; b++)
            for (int c = 0; a + b + c <= 7; c++) {
                const bool red = a <= 2 && b <= 2 && c <= 2;
                const bool bnd = (a == 3 && b <= 2 && c <= 2) || (b == 3 && a <= 2 && c <= 2) || (c == 3 && a <= 2 && b <= 2);
                pos[a][b][c] = red ? 9 * a + 3 * b + c : (bnd ? n_bnd++ : n_mid++);
            }
    for (int a = 0; a <= 7; a++)
        for (int b = 0; a + b <= 7; b++)
            for (int c = 0; a + b + c <= 7; c++) {
                if (a <= 2 && b <= 2 && c <= 2) continue;
                const int row = pos[a][b][c] - 27;
                int which, ma = a, mb = b, mc = c;
                if (a >= 3) { which = 0; ma -= 3; } else if (b >= 3) { which = 1; mb -= 3; } else { which = 2; mc -= 3; }
                t.row_which[row] = (int8_t)which;
                for (int x = 0; x <= 3; x++)
                    for (int y = 0; x + y <= 3; y++)
                        for (int z = 0; x + y + z <= 3; z++) {
                            const int col = pos[ma + x][mb + y][mc + z];
                            t.row_dst[row][idx_le(x, y, z, 3)] = (int16_t)(col >= 27 ? col - 27 : 93 + col);
                        }
            }
    for (int a = 0; a <= 2; a++)
        for (int b = 0; b <= 2; b++)
            for (int c = 0; c <= 2; c++)
                for (int s = 0; s < 4; s++) t.s_col[9 * a + 3 * b + c][s] = (int16_t)pos[a + SH[s][0]][b + SH[s][1]][c + SH[s][2]];
    for (int r = 0; r < 93; r++) {
        for (int c = 0; c < 128; c++) t.col_term[r][c] = -1;
        for (int term = 0; term < 20; term++) t.col_term[r][t.row_dst[r][term]] = (int8_t)term;   // a later term overwrites, as the fill did
    }
}

// ------------------------------------------------------------------------------------------------ device helpers
// wave-wide max of a NON-NEGATIVE double (or NaN-free |x|) with DPP row ops; result broadcast via readlane(63).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const double o = __longlong_as_double(((long long)hi2 << 32) | (unsigned int)lo2);
    return o > v ? o : v;
}
// wave-wide max of a 32-bit unsigned value: six v_max_u32 with DPP source modifiers -- ONE instruction per step, issued from inline asm
// (hipcc expands the update_dpp builtin + max into v_mov / s_nop / v_mov_dpp / v_max, three dependent instructions per step).  A lane
// whose DPP source does not exist is not written (bound_ctrl off) and keeps its own value; the result is read from lane 63.
__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// wave-wide max of a NON-NEGATIVE, non-NaN double: such doubles order like their bit patterns, so the max is found on the high
// words first and on the low words of the lanes that attain it second -- 12 one-instruction DPP steps instead of 6 x (two DPP moves,
// a 64-bit compare, two selects).  Same value as the 64-bit ladder below (kept for reference / A-B).
__device__ __forceinline__ double wave_max_nonneg_2x32(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned hi = (unsigned)(b >> 32), lo = (unsigned)b;
    const unsigned mhi = wave_umax(hi);
    const unsigned mlo = wave_umax(hi == mhi ? lo : 0u);
    return __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
}

__device__ __forceinline__ double wave_max_nonneg(double v)
{
    v = dpp_max_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row max
    v = dpp_max_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
    v = dpp_max_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave max
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// value of `v` in lane j (wave-uniform j): two v_readlane_b32, result in SGPRs
__device__ __forceinline__ double lane_value_f64(double v, int j)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), j);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), j);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ------------------------------------------------------------------------------------------------ K4 + K5a
constexpr int kNR = 93, kNC = 120;

// One RANSAC problem of a launch.  A launch covers up to kPnpMaxBatch independent problems (same parameters, own
// correspondences and seed): workgroup b works on hypothesis b % H of problem b / H; all per-hypothesis arrays are
// indexed by b.  A single wave per hypothesis leaves the SIMDs latency-bound at H = 1000, so co-scheduling the
// independent estimations of a loop candidate (PNP a->b, PNP b->a: Cerebro.cpp:1518,1572) is nearly free.
struct PnpProblem {
    const double *X;    // N x 3
    const double *uv;   // N x 2
    int32_t N, pad_;
    uint64_t seed;
};
constexpr int kPnpMaxBatch = 8;

struct SolveArgs {
    PnpProblem prob[kPnpMaxBatch];
    int32_t H, S;
    int32_t slot0;          // first slot of this launch (a batch may be issued as several launch pairs)
    int32_t factor_prio;    // > 0: the LU's factor wave runs at raised issue priority (s_setprio)
    const PnpTables *tab;
    double *Sg;         // [H][729]  action matrices
    double *Tg;         // [H][27]   translation factor (t = Tfac * vec(R))
    int32_t *sample;    // [H][kSampleMax]
    const int32_t *sample_in;   // CHIP_SAMPLER_THEIA_PERSISTENT: [P * H][kSampleMax] made by the host (pinned, device-mapped); else nullptr
    int32_t *ok;        // [H] 1 = S valid, 0 = singular D
    unsigned long long *stamps;   // tuning only (CHIP_PNP_STAMPS): [H][8] s_memtime at the phase boundaries of pnp_build_solve
    // The correspondences arrive in pinned, device-mapped HOST memory (no H2D copy in front of the launch: 6 us of copy + ~10 us of
    // dependent-launch latency on a 0.5 ms call).  prob[].X / .uv of THIS kernel point there: a workgroup reads its 15 sample points across
    // PCIe (one ~2 us round trip underneath the sampler), and all workgroups together copy the n_in doubles to in_dev, which is what
    // pnp_eig_score's prob[] points at -- it scores every hypothesis against all N points and must not do that over PCIe.
    const double *in_host;
    double *in_dev;
    int64_t n_in;
};

constexpr int kSolveThreads = 448;   // waves 0..5 hold the Macaulay block (3 row groups x 128 columns), wave 6 factorises panels
constexpr int kPanel = 4;   // pivot columns factorised per panel of the blocked LU
// dynamic LDS of pnp_build_solve (must match the carve-up at the top of the kernel)
constexpr size_t kSolveLds = sizeof(double) * (kSampleMax * 8 + 9 + 9 + 27 + 27 + 81 + 90 + 100 + 36 + 60 + 4 + 27 * 27 +
                                               2 * 96 * kPanel * 2 + 2 * kPanel * 128 + 2 * kPanel * kPanel + 27 * kNC) +
                             sizeof(int) * (16 + 4 + 2 * kPanel) + 64;

// two workgroups per CU (7 waves each, 4 SIMDs): at most 128 VGPRs -- asked for explicitly, the budget is not left to chance
// STAMP = true (CHIP_PNP_STAMPS=1, tuning only): shader-clock stamps at the phase boundaries; compiled out of the product kernel
// (they were run-time-guarded before: four guarded blocks per panel in every wave of the LU loop).
template <bool STAMP>
__global__ __launch_bounds__(kSolveThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void pnp_build_solve(SolveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *sm = reinterpret_cast<double *>(smem);
    double *sx = sm;            sm += kSampleMax * 3;             // sample points
    double *suv = sm;           sm += kSampleMax * 2;
    double *zb = sm;            sm += kSampleMax * 3;             // bearings
    double *Szz = sm;           sm += 9;
    double *Hm = sm;            sm += 9;
    double *W = sm;             sm += 27;
    double *Tf = sm;            sm += 27;
    double *M9 = sm;            sm += 81;
    double *MQ = sm;            sm += 90;
    double *G = sm;             sm += 100;
    double *c4 = sm;            sm += 36;
    double *fc = sm;            sm += 60;                         // f[3][20]
    double *uu = sm;            sm += 4;
    double *Xb = sm;            sm += 27 * 27;                    // X[66+t][c]
    double *panel = sm;         sm += 2 * 96 * kPanel;            // [parity][physical row][c]: pivot columns of panels p, p+1
    double *Lp = sm;            sm += 2 * 96 * kPanel;            // [parity][physical row][c]: multipliers of a panel
    double *prow_raw = sm;      sm += 2 * kPanel * 128;           // [parity][c][column]: a panel's pivot rows as they were before the panel
    double *Lsub = sm;          sm += 2 * kPanel * kPanel;        // [parity][c][c' < c]: multipliers among a panel's pivot rows
    double *Urows = sm;         sm += 27 * kNC;                   // pivot rows of steps 66..92 = rows of U needed by the back-substitution
    int *smp = reinterpret_cast<int *>(sm);                       // [16]
    int *flag = smp + 16;                                          // [4] : singular, spare
    int *prow_s = flag + 4;                                        // [parity][kPanel] physical pivot rows of a panel
    unsigned long long *nzm = reinterpret_cast<unsigned long long *>(prow_s + 2 * kPanel);   // [parity][2] rows (0..63 | 64..92) with a non-zero multiplier in the panel (32 of the 64 spare bytes)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int slot = blockIdx.x + a.slot0;
    const int pi = slot / a.H;
    const int hyp = slot - pi * a.H;                               // hypothesis index within its problem (keys the RNG)
    const PnpProblem pr = a.prob[pi];
    const int n = a.S;
    const PnpTables &tb = *a.tab;

#define SOLVE_STAMP(i) do { if (STAMP && tid == 0) a.stamps[(size_t)slot * 24 + (i)] = __builtin_readcyclecounter(); } while (0)
    SOLVE_STAMP(0);
    if (a.in_host != nullptr)   // this launch's share of the host -> device copy of the correspondences (see SolveArgs)
        for (int64_t e = (int64_t)blockIdx.x * kSolveThreads + tid; e < a.n_in; e += (int64_t)gridDim.x * kSolveThreads) a.in_dev[e] = a.in_host[e];
    // ---- sampler: partial Fisher-Yates over a virtual identity permutation (theia::RandomSampler restated) ----
    if (wave == 0) {
        // (sample_in: the persistent-permutation mode -- the host sequenced the swaps, this hypothesis reads its row of the table)
        const int sv = a.sample_in ? (lane < n ? a.sample_in[(size_t)slot * kSampleMax + lane] : 0) : ransac_sample_wave(pr.seed, hyp, pr.N, n, lane);
        if (lane < n) smp[lane] = sv;
        if (lane < 4) {   // random linear form f0 (Theia: 100 * Vector4d::Random())
            const uint64_t x = rng_draw(pr.seed, (uint32_t)hyp, (uint32_t)(64 + lane));
            const double f = (double)(x >> 11) * (1.0 / 9007199254740992.0);
            uu[lane] = 100.0 * (2.0 * f - 1.0);
        }
        if (lane == 0) flag[0] = 0;
    }
    __syncthreads();
    if (tid < n) {
        const int s = smp[tid];
        a.sample[slot * kSampleMax + tid] = s;
        const double X0 = pr.X[3 * s], X1 = pr.X[3 * s + 1], X2 = pr.X[3 * s + 2];
        const double u = pr.uv[2 * s], v = pr.uv[2 * s + 1];
        sx[3 * tid] = X0; sx[3 * tid + 1] = X1; sx[3 * tid + 2] = X2;
        suv[2 * tid] = u; suv[2 * tid + 1] = v;
        const double nrm = sqrt((u * u + v * v) + 1.0);
        zb[3 * tid] = u / nrm; zb[3 * tid + 1] = v / nrm; zb[3 * tid + 2] = 1.0 / nrm;
    }
    __syncthreads();
    // ---- H = (n I - sum z z^T)^-1 ; W = sum (z z^T - I) L(p) ----
    if (tid < 9) {
        const int aa = tid / 3, bb = tid % 3;
        double s = 0.0;
        for (int i = 0; i < n; i++) s = s + zb[3 * i + aa] * zb[3 * i + bb];
        Szz[tid] = s;
    } else if (tid >= 64 && tid < 64 + 27) {
        const int t = tid - 64, aa = t / 9, j = t % 9, bb = j / 3, cc = j % 3;
        double s = 0.0;
        for (int i = 0; i < n; i++) {
            const double e = zb[3 * i + aa] * zb[3 * i + bb] - (aa == bb ? 1.0 : 0.0);
            s = s + e * sx[3 * i + cc];
        }
        W[t] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double m[3][3];
        for (int aa = 0; aa < 3; aa++)
            for (int bb = 0; bb < 3; bb++) m[aa][bb] = (aa == bb ? (double)n : 0.0) - Szz[3 * aa + bb];
        const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
        const double c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
        const double c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
        const double c10 = m[0][2] * m[2][1] - m[0][1] * m[2][2];
        const double c11 = m[0][0] * m[2][2] - m[0][2] * m[2][0];
        const double c12 = m[0][1] * m[2][0] - m[0][0] * m[2][1];
        const double c20 = m[0][1] * m[1][2] - m[0][2] * m[1][1];
        const double c21 = m[0][2] * m[1][0] - m[0][0] * m[1][2];
        const double c22 = m[0][0] * m[1][1] - m[0][1] * m[1][0];
        const double det = (m[0][0] * c00 + m[0][1] * c01) + m[0][2] * c02;
        Hm[0] = c00 / det; Hm[1] = c10 / det; Hm[2] = c20 / det;
        Hm[3] = c01 / det; Hm[4] = c11 / det; Hm[5] = c21 / det;
        Hm[6] = c02 / det; Hm[7] = c12 / det; Hm[8] = c22 / det;
    }
    __syncthreads();
    if (tid < 27) {
        const int aa = tid / 9, j = tid % 9;
        Tf[tid] = (Hm[3 * aa] * W[j] + Hm[3 * aa + 1] * W[9 + j]) + Hm[3 * aa + 2] * W[18 + j];
    }
    __syncthreads();
    if (tid < 27) a.Tg[slot * 27 + tid] = Tf[tid];
    SOLVE_STAMP(1);   // sampler, bearings, H, W, T done
    // ---- M9 = sum (L+T)^T (I - z z^T) (L+T) ----
    if (tid < 81) {
        const int j = tid / 9, k = tid % 9;
        const int jb = j / 3, jc = j % 3, kb = k / 3, kc = k % 3;
        double acc = 0.0;
        for (int i = 0; i < n; i++) {
            double Aj[3], Ak[3], Bk[3];
            for (int r = 0; r < 3; r++) {
                Aj[r] = Tf[9 * r + j] + (r == jb ? sx[3 * i + jc] : 0.0);
                Ak[r] = Tf[9 * r + k] + (r == kb ? sx[3 * i + kc] : 0.0);
            }
            for (int r = 0; r < 3; r++) {
                const double p0 = (r == 0 ? 1.0 : 0.0) - zb[3 * i + r] * zb[3 * i + 0];
                const double p1 = (r == 1 ? 1.0 : 0.0) - zb[3 * i + r] * zb[3 * i + 1];
                const double p2 = (r == 2 ? 1.0 : 0.0) - zb[3 * i + r] * zb[3 * i + 2];
                Bk[r] = (p0 * Ak[0] + p1 * Ak[1]) + p2 * Ak[2];
            }
            acc = acc + ((Aj[0] * Bk[0] + Aj[1] * Bk[1]) + Aj[2] * Bk[2]);
        }
        M9[tid] = acc;
    }
    __syncthreads();
    if (tid < 90) {  // MQ = M9 Q
        const int j = tid / 10, q = tid % 10;
        double s = 0.0;
        for (int k = 0; k < 9; k++) s = s + M9[9 * j + k] * tb.qmat[k][q];
        MQ[tid] = s;
    }
    __syncthreads();
    if (tid < 100) {  // G = Q^T MQ
        const int l = tid / 10, q = tid % 10;
        double s = 0.0;
        for (int j = 0; j < 9; j++) s = s + tb.qmat[j][l] * MQ[10 * j + q];
        G[tid] = s;
    }
    __syncthreads();
    if (tid < 35) {  // quartic coefficients: c4[k] = sum of G[l][q] with m10[l]*m10[q] == monomial k, in (l,q) row-major order
        double s = 0.0;
        const int cnt = tb.c4_cnt[tid];
        for (int e = 0; e < cnt; e++) s = s + G[tb.c4_src[tid][e]];
        c4[tid] = s;
    }
    __syncthreads();
    if (tid < 60) {  // f_k = dJ'/ds_k
        const int k = tid / 20, mth = tid % 20;
        fc[tid] = (double)tb.fmul[k][mth] * c4[tb.fsrc[k][mth]];
    }
    __syncthreads();
    SOLVE_STAMP(2);   // cost matrix -> cubics done
    // ---- Macaulay [D | C] (93 x 120) lives in REGISTERS of waves 0..5: thread (tx = column, ty = row group) holds rows
    //      ty, ty+3, ..., ty+90 of its column in er[0..30] (static slot indices); wave 6 never touches it. ----
    const int tx = tid & 127, ty = tid >> 7;          // ty == 3 <=> the factor wave
    const bool is_factor = wave == 6;
    double er[31];
    unsigned live = 0;   // bit s set <=> physical row ty + 3*s has not been used as a pivot yet (wave-uniform)
    if (!is_factor) {
        // row r of the Macaulay block = cubic `row_which[r]` times a monomial: its 20 coefficients land in the columns listed
        // by row_dst; col_term is the inverse map, so each thread fetches "its" term of each of its rows directly
        signed char term[31];
#pragma unroll
        for (int sl = 0; sl < 31; sl++) term[sl] = tb.col_term[ty + 3 * sl][tx];
#pragma unroll
        for (int sl = 0; sl < 31; sl++) {
            const int row = ty + 3 * sl;
            live |= 1u << sl;
            er[sl] = term[sl] >= 0 ? fc[20 * tb.row_which[row] + term[sl]] : 0.0;
        }
        if (tx < 2 * kPanel) {   // columns of panels 0 and 1 -> panel[0], panel[1]
            double *dst = panel + (tx / kPanel) * 96 * kPanel + (tx % kPanel);
#pragma unroll
            for (int sl = 0; sl < 31; sl++) dst[(ty + 3 * sl) * kPanel] = er[sl];
        }
    } else {
#pragma unroll
        for (int sl = 0; sl < 31; sl++) er[sl] = 0.0;
    }
    if (tid == 0) flag[0] = 0;
    if (tid < 2 * kPanel) prow_s[tid] = -1;   // "not decided yet" (polled by the matrix waves, see the LU loop)
    __syncthreads();

    SOLVE_STAMP(3);   // Macaulay block filled
    // ---- Blocked right-looking LU with partial (row) pivoting, no physical swaps, with LOOKAHEAD: the panel of kPanel pivot
    //      columns is factorised by a wave of its own (wave 6, two physical rows per lane, registers / DPP / readlane only)
    //      while waves 0..5 apply the previous panel to the trailing matrix.  Per panel p:
    //   P3  (waves 0..5) the owners of panel p's pivot rows publish them as they were before the panel;   -- barrier --
    //   P4  (waves 0..5) every thread rebuilds the pivot rows' values for its column (triangular solve with the multipliers
    //       among the pivot rows), applies the kPanel rank-1 updates to its 31 register-resident rows ONE AFTER THE OTHER (so
    //       every element sees exactly the operation sequence of the unblocked elimination), and the owners of panel p+2's
    //       columns dump them;
    //   F   (wave 6, concurrently with P4) brings its copy of panel p+1's columns up to date with panel p (same operations),
    //       then factorises it: pivot = max |column| over the rows not yet used, ties -> smallest LOGICAL index (lp0/lp1 track
    //       where the reference's row swaps would have put every row), multipliers by IEEE division, remaining panel
    //       columns updated; publishes multipliers / pivot rows / sub-multipliers of panel p+1.                -- barrier --
    //   The critical path is F's chain (one pivot-search latency chain per column); the trailing update hides behind it. ----
    bool singular = false;
    constexpr int kPanels = (kNR + kPanel - 1) / kPanel;
    // factor-wave state.  The factor wave does not hold matrix rows, so its panel lives in the (otherwise idle) er[] registers
    // -- the kernel's register budget is set by the matrix waves, and two workgroups per CU need <= 128 VGPRs:
    //   A0(c)/A1(c): panel entries of physical rows lane / lane+64;  L0(c)/L1(c): their multipliers;
    //   LS(c,c2), c2 < c: multipliers among the panel's pivot rows;  B0/B1: the next panel while it is brought up to date
#define A0(c) er[(c)]
#define A1(c) er[4 + (c)]
#define L0(c) er[8 + (c)]
#define L1(c) er[12 + (c)]
#define LS(c, c2) er[16 + ((c) * ((c) - 1)) / 2 + (c2)]
#define B0(c) er[22 + (c)]
#define B1(c) er[26 + (c)]
    // Which physical rows are still candidates is wave-uniform state: two lane masks in SGPRs (rows lane / lane + 64), used directly as
    // select conditions (inverse ballot) and updated by scalar bit instructions.  The reference's row swaps are not tracked per column any
    // more: where a row WOULD sit logically only matters when two candidates tie exactly (smallest logical index wins), which generic
    // data never does (0 of 93 000 columns of config 3's 1000 hypotheses) -- that path replays the swaps from the pivot history instead.
    // The factor wave is bound by the issue rate of a lone wave (~9 cycles per instruction, dependent or not: ~150 instructions and
    // ~1350 cycles per column before this), so what counts is the number of instructions on the per-column path.
    const bool has1 = lane + 64 < kNR;
    // (the branch on the wave's role is divergent as far as the compiler knows, so what the factor wave carries around the LU loop lives
    //  in VGPRs: the masks are moved to SGPRs once per panel -- declaring the role uniform instead costs the matrix waves 129 spills)
    unsigned long long alive0_v = ~0ull, alive1_v = (1ull << (kNR - 64)) - 1ull;
    int hist0 = 0, hist1 = 0;   // lane j: physical pivot row of elimination step j / j + 64 (read by the tie path only)
    auto uniform_u64 = [](unsigned long long v) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };

    // tuning only (CHIP_PNP_STAMPS): sub-phase shader-clock totals of the factor wave
    // (compiled in with -DCHIP_PNP_FSTAMPS only: eight 64-bit accumulators push the kernel past 128 VGPRs = one workgroup per CU,
    // and every stamp costs ~60 cycles on the chain it measures)
#ifdef CHIP_PNP_FSTAMPS
    unsigned long long f_t = __builtin_readcyclecounter(), f_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool f_stamp = STAMP && wave == 6;
#define F_STAMP(i) do { if (f_stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); f_acc[i] += t_ - f_t; f_t = t_; } } while (0)
#else
#define F_STAMP(i) do { } while (0)
#endif
    // factorise the panel held in a0/a1 (columns k .. k+BW-1), publish into parity set `par`.  BW is a compile-time constant (4, or 1
    // for the last panel: 93 = 23 x 4 + 1): no per-column guards.  A zero / non-finite pivot column marks the block singular and the
    // panel is finished with whatever row was picked -- garbage, but in-range garbage: the workgroup leaves at the top of the next
    // panel (flag[0]) exactly as before, and nothing of a rejected block is looked at.
    auto factor_panel = [&](auto bw_tag, int k, int par) {
        constexpr int BW = decltype(bw_tag)::value;
        F_STAMP(0);   // whatever preceded (the update of the panel by the previous one)
        int pr_c[kPanel] = {0, 0, 0, 0};
        bool sing = false;
        unsigned long long alive0 = uniform_u64(alive0_v), alive1 = uniform_u64(alive1_v);
        int *const prow_par = prow_s + par * kPanel;
#pragma unroll
        for (int c = 0; c < kPanel; c++) { L0(c) = 0.0; L1(c) = 0.0; }
        // one pivot column.  The column index is a compile-time constant (generic lambda over integral_constant), so that every
        // A0(c) / LS(c, c2) / urow[c2] below is a fixed register whatever the unroller decides.
        auto factor_column = [&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value;
            if constexpr (c < BW) {
                const bool al0 = __builtin_amdgcn_inverse_ballot_w64(alive0), al1 = __builtin_amdgcn_inverse_ballot_w64(alive1);
                // Finite non-negative doubles order like their bit patterns, so the HIGH words (sign cleared) decide unless two
                // candidates agree in exponent and 20 mantissa bits.  Usual case: one lane attains the max high word and its two rows
                // differ there -> that row is the pivot, found with 32-bit integer work and ONE DPP reduction.  Everything else -- ties on
                // a high word, exact ties, an all-zero / denormal column, a NaN or Inf (high word >= 0x7ff00000) -- takes the full path
                // (64-bit compares, second reduction over the low words, smallest-logical-index rule).  Same pivot either way.
                const unsigned h0 = al0 ? ((unsigned)((unsigned long long)__double_as_longlong(A0(c)) >> 32) & 0x7fffffffu) : 0u;
                const unsigned h1 = al1 ? ((unsigned)((unsigned long long)__double_as_longlong(A1(c)) >> 32) & 0x7fffffffu) : 0u;
                const bool up = h1 > h0;                               // per lane: its second row is the larger one
                const unsigned vhi_i = up ? h1 : h0;
                const unsigned mhi_i = wave_umax(vhi_i);
                const unsigned long long cand = __builtin_amdgcn_ballot_w64(vhi_i == mhi_i);
                const unsigned long long eqm = __builtin_amdgcn_ballot_w64(h1 == h0);
                F_STAMP(1);   // candidates + wave max
                // olane / ohalf / pivot_ok are wave-uniform and are KEPT in SGPRs: the tie path computes them with vector instructions,
                // and merged with the usual path's scalar values they all lived in VGPRs (every later use went through
                // v_readfirstlane / 64-bit vector mask arithmetic)
                int olane, ohalf;
                bool pivot_ok;
                const int cl = __builtin_ctzll(cand);                  // cand != 0: the lane holding the max is in it
                // (one candidate lane, a finite non-zero maximum, its two rows differ in the high word -- as ONE scalar sum: fewer SALU
                //  instructions than three conditions and-ed as lane masks)
                if (__popcll(cand) + (mhi_i - 1u < 0x7fefffffu ? 0 : 2) + ((eqm & cand) != 0ull ? 2 : 0) == 1) {
                    olane = cl;
                    ohalf = (int)((__builtin_amdgcn_ballot_w64(up) >> cl) & 1ull);   // a scalar bit test of the compare's own lane mask
                    pivot_ok = true;
                } else {
                    const double v0 = al0 ? fabs(A0(c)) : -1.0, v1 = al1 ? fabs(A1(c)) : -1.0;
                    double vm = v0 > 0.0 ? v0 : 0.0;                   // NaN never wins, as in the reference scan
                    if (v1 > vm) vm = v1;
                    const double best = wave_max_nonneg_2x32(vm);
                    const bool w0 = al0 && v0 == best, w1 = al1 && v1 == best;
                    const unsigned long long t0 = __builtin_amdgcn_ballot_w64(w0), t1 = __builtin_amdgcn_ballot_w64(w1);
                    int ol, oh;
                    bool found = true;
                    if (__popcll(t0) + __popcll(t1) == 1) {           // one row attains the max
                        oh = t0 == 0ull;
                        ol = __builtin_ctzll(oh ? t1 : t0);
                    } else {
                        // exact ties: the reference's scan keeps the FIRST row of the maximum, i.e. the smallest LOGICAL index -- where
                        // its row swaps have put the rows by now: step j exchanges the row at logical position j with its pivot row
                        int lp0 = lane, lp1 = lane + 64;
                        auto displace = [&](int j, int prow) {         // both wave-uniform
                            const int a0 = __builtin_amdgcn_readlane(lp0, prow & 63), a1 = __builtin_amdgcn_readlane(lp1, prow & 63);
                            const int was = prow < 64 ? a0 : a1;
                            lp0 = lane == prow ? j : (lp0 == j ? was : lp0);
                            lp1 = lane + 64 == prow ? j : (lp1 == j ? was : lp1);
                        };
                        for (int j = 0; j < k; j++) {
                            const int h = __builtin_amdgcn_readlane(j < 64 ? hist0 : hist1, j & 63);
                            displace(j, h);
                        }
#pragma unroll
                        for (int c2 = 0; c2 < c; c2++) displace(k + c2, pr_c[c2]);
                        int pl = w0 ? lp0 : 0x7fffffff;
                        if (w1 && lp1 < pl) pl = lp1;
#pragma unroll
                        for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(pl, m, 64); pl = o < pl ? o : pl; }
                        found = pl != 0x7fffffff;
                        const unsigned long long o0 = __builtin_amdgcn_ballot_w64(al0 && lp0 == pl), o1 = __builtin_amdgcn_ballot_w64(al1 && lp1 == pl);
                        oh = o0 == 0ull;
                        ol = __builtin_ctzll(oh ? (o1 ? o1 : 1ull) : o0);
                    }
                    olane = __builtin_amdgcn_readfirstlane(ol);
                    ohalf = __builtin_amdgcn_readfirstlane(oh);
                    pivot_ok = __builtin_amdgcn_readfirstlane((best > 0.0 && found) ? 1 : 0) != 0;
                }
                F_STAMP(2);   // who attains it: compares, ballots, logical position
                if (!pivot_ok) sing = true;
                pr_c[c] = olane + (ohalf ? 64 : 0);
                // the owners publish that row at once.  A relaxed workgroup-scope atomic, NOT a volatile store: the volatile form
                // compiled to flat_store_dword + s_waitcnt vmcnt(0) (address-space inference skips volatile accesses), a
                // several-hundred-cycle stall on the critical chain of every column; this is one ds_write_b32, no wait
                // (stored by every lane: same address, same value -- no exec juggling on the chain)
                __hip_atomic_store(&prow_par[c], pr_c[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // the pivot row's entries / its earlier multipliers by readlane, and the pivot row leaves the candidates (what is left are
                // exactly the rows that get a multiplier): one wave-uniform branch on the half instead of a vector select per value
                double urow[kPanel];
                if (ohalf) {
#pragma unroll
                    for (int c2 = 0; c2 < kPanel; c2++) {
                        urow[c2] = lane_value_f64(A1(c2), olane);
                        if (c2 < c) LS(c, c2) = lane_value_f64(L1(c2), olane);
                    }
                    alive1 &= ~(1ull << olane);
                } else {
#pragma unroll
                    for (int c2 = 0; c2 < kPanel; c2++) {
                        urow[c2] = lane_value_f64(A0(c2), olane);
                        if (c2 < c) LS(c, c2) = lane_value_f64(L0(c2), olane);
                    }
                    alive0 &= ~(1ull << olane);
                }
                F_STAMP(3);   // pivot row entries / earlier multipliers by readlane
                const double piv = urow[c];
                const bool n0 = __builtin_amdgcn_inverse_ballot_w64(alive0), n1 = __builtin_amdgcn_inverse_ballot_w64(alive1);
                const double m0 = n0 ? A0(c) / piv : 0.0;
                const double m1 = n1 ? A1(c) / piv : 0.0;
                L0(c) = m0; L1(c) = m1;
                F_STAMP(4);   // the two IEEE divisions
#pragma unroll
                for (int c2 = 0; c2 < kPanel; c2++)
                    if (c2 > c) {
                        // the reference skips l == 0; e - 0 * u == e for the finite u of a block that is not rejected anyway
                        // (same argument as in P4), so no compare / select on the chain to the next pivot search
                        A0(c2) = A0(c2) - m0 * urow[c2];
                        A1(c2) = A1(c2) - m1 * urow[c2];
                    }
                F_STAMP(5);   // panel update
            }
        };
        static_assert(kPanel == 4 && kNR % kPanel == 1, "factor_panel names its four columns; the last panel has one");
        factor_column(std::integral_constant<int, 0>{});
        factor_column(std::integral_constant<int, 1>{});
        factor_column(std::integral_constant<int, 2>{});
        factor_column(std::integral_constant<int, 3>{});
        alive0_v = alive0; alive1_v = alive1;
        double *Lpp = Lp + par * 96 * kPanel;
#pragma unroll
        for (int c = 0; c < kPanel; c++) {
            Lpp[lane * kPanel + c] = L0(c);
            if (has1) Lpp[(lane + 64) * kPanel + c] = L1(c);
        }
        {   // which rows have ANY non-zero multiplier in this panel (round 5): the Macaulay rows are sparse, 47 % of the (row, panel) pairs
            // of the elimination have all four multipliers zero (instrumented oracle) -- the matrix waves skip those rows' updates, as the
            // reference algorithm does (if (l != 0) ...), by the scalar branch that already skips rows used as pivots
            const bool z0 = (L0(0) != 0.0) | (L0(1) != 0.0) | (L0(2) != 0.0) | (L0(3) != 0.0);
            const bool z1 = has1 && ((L1(0) != 0.0) | (L1(1) != 0.0) | (L1(2) != 0.0) | (L1(3) != 0.0));
            const unsigned long long q0 = __builtin_amdgcn_ballot_w64(z0), q1 = __builtin_amdgcn_ballot_w64(z1);
            if (lane == 0) { nzm[par * 2] = q0; nzm[par * 2 + 1] = q1; }
        }
#pragma unroll
        for (int c = 0; c < BW; c++) {                               // pivot history (k is a multiple of 4: a panel never straddles step 64)
            if (k < 64) hist0 = lane == k + c ? pr_c[c] : hist0;
            else hist1 = lane == k + c - 64 ? pr_c[c] : hist1;
        }
        if (lane == 0) {
            if (sing) flag[0] = 1;
            // (every pivot row index went out the moment it was decided.)  Of Lsub only the entries below the diagonal of a full panel are
            // ever multiplied by something that is not a literal zero: P4's uz[c2] is 0.0 for c2 >= bw, and what it computes for a pivot
            // row c >= bw is selected away (never used in arithmetic) -- so the last, one-column panel writes nothing here.
#pragma unroll
            for (int c = 1; c < BW; c++)
#pragma unroll
                for (int c2 = 0; c2 < c; c2++) Lsub[par * kPanel * kPanel + c * kPanel + c2] = LS(c, c2);
        }
        F_STAMP(6);   // publication of multipliers / pivot rows / sub-multipliers
    };
    auto load_panel = [&](int par) {
        const double *pp = panel + par * 96 * kPanel;
#pragma unroll
        for (int c = 0; c < kPanel; c++) {
            A0(c) = pp[lane * kPanel + c];
            A1(c) = has1 ? pp[(lane + 64) * kPanel + c] : 0.0;
        }
    };

    // The factor wave's chain is the LU's critical path and it shares its SIMD with matrix waves streaming fp64 FMAs: at equal
    // priority it gets one issue slot in ~4 (shader-clock split: ~20 cycles per dependent instruction).  Raised priority lets the
    // arbiter pick it whenever it is ready; the matrix waves fill the slots its latencies leave.  CHIP_PNP_PRIO (tuning knob).
    if (is_factor && a.factor_prio > 0) __builtin_amdgcn_s_setprio(3);
    if (is_factor) { load_panel(0); factor_panel(std::integral_constant<int, kPanel>{}, 0, 0); }
    __syncthreads();
    // tuning only (CHIP_PNP_STAMPS): per-phase shader-clock totals of the LU, wave 0 (a matrix wave) and wave 6 (the factor wave)
    unsigned long long lu_t = 0, lu_acc[4] = {0, 0, 0, 0};
    const bool lu_stamp = STAMP && lane == 0 && (wave == 0 || wave == 6);
#define LU_STAMP(i) do { if (lu_stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); lu_acc[i] += t_ - lu_t; lu_t = t_; } } while (0)
    if (lu_stamp) lu_t = __builtin_readcyclecounter();
#ifdef CHIP_PNP_FSTAMPS
    if (f_stamp) f_t = __builtin_readcyclecounter();
#endif
    // P3 as a function: the owners of panel q's pivot rows publish them (as they are NOW: after the owner's P4 of panel q-1) into
    // prow_raw[q & 1].  The pivot row indices are POLLED from LDS: the factor wave stores each one the moment it is decided, while the
    // matrix waves -- done with their trailing update ~2 k cycles before the factor wave (shader-clock split) -- would otherwise idle
    // at the barrier; so the publication of panel q runs underneath the factorisation of panel q instead of between two barriers
    // on the critical path (it was 1.5-2.1 k of ~10 k cycles per panel, plus a barrier).
    auto publish_pivot_rows = [&](int q) {
        const int kq = q * kPanel, parq = q & 1;
        const int bwq = (kNR - kq) < kPanel ? (kNR - kq) : kPanel;
#pragma unroll
        for (int c = 0; c < kPanel; c++) {
            if (c < bwq) {
                int prow = -1;
                for (int spin = 0; spin < (1 << 22); spin++) {       // bounded: a wedged factor wave must not hang the GPU
                    prow = __hip_atomic_load(&prow_s[parq * kPanel + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_read_b32
                    if (prow >= 0) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (prow < 0) { flag[0] = 1; prow = 0; }
                if (prow % 3 == ty) {
                    const int ps = __builtin_amdgcn_readfirstlane(prow / 3);
                    // er[ps] with a wave-uniform ps: a scalar switch (one jump, one register move) -- a run-time index would
                    // send er[] to scratch, and the 30-deep select chain used before cost ~500 cycles per pivot row
                    double u;
                    switch (ps) {
#define ER_CASE(n) case n: u = er[n]; break;
                        ER_CASE(0) ER_CASE(1) ER_CASE(2) ER_CASE(3) ER_CASE(4) ER_CASE(5) ER_CASE(6) ER_CASE(7) ER_CASE(8) ER_CASE(9)
                        ER_CASE(10) ER_CASE(11) ER_CASE(12) ER_CASE(13) ER_CASE(14) ER_CASE(15) ER_CASE(16) ER_CASE(17) ER_CASE(18)
                        ER_CASE(19) ER_CASE(20) ER_CASE(21) ER_CASE(22) ER_CASE(23) ER_CASE(24) ER_CASE(25) ER_CASE(26) ER_CASE(27)
                        ER_CASE(28) ER_CASE(29)
#undef ER_CASE
                        default: u = er[30]; break;
                    }
                    prow_raw[(parq * kPanel + c) * 128 + tx] = u;
                    live &= ~(1u << (ps & 31));   // (ps <= 30 for a real pivot row; a singular block's leftovers may name rows past 92)
                }
            }
        }
    };
    if (!is_factor) publish_pivot_rows(0);
    __syncthreads();
    for (int p = 0; p < kPanels; p++) {
        const int k = p * kPanel, par = p & 1;
        const int bw = (kNR - k) < kPanel ? (kNR - k) : kPanel;
        if (flag[0]) { singular = true; break; }
        // panel p's pivot row indices were consumed before the barrier above: mark the slots "undecided" for panel p + 2 (the factor
        // wave fills prow_s[par ^ 1] in this iteration and prow_s[par] in the next one, behind the barrier that ends this one)
        if (is_factor && lane < kPanel) prow_s[par * kPanel + lane] = -1;
        LU_STAMP(0);
        LU_STAMP(1);
        if (!is_factor) {
            // -- P4
            const double *Lss = Lsub + par * kPanel * kPanel;
            double u[kPanel], uz[kPanel];
#pragma unroll
            for (int c = 0; c < kPanel; c++) {
                double v = prow_raw[(par * kPanel + c) * 128 + tx];
#pragma unroll
                for (int c2 = 0; c2 < kPanel; c2++)
                    if (c2 < c) v = v - Lss[c * kPanel + c2] * uz[c2];   // select-free, as the row updates below (uz, see there)
                u[c] = v;
                uz[c] = (c < bw && tx > k + c) ? v : 0.0;
                if (ty == 0 && tx < kNC && c < bw && k + c >= 66) Urows[(k + c - 66) * kNC + tx] = v;
            }
            // The reference's elimination step is  if (l != 0) e -= l * u  on the columns right of the pivot.  The per-element form
            // of that (a compare and a 64-bit select per row and pivot: 3 of 5 VALU instructions) made P4 the LU's bottleneck -- 620
            // fp64 VALU instructions per wave and panel, 3.5 waves per SIMD = ~9 k cycles against the ~4-7 k of the factor wave's
            // chain (shader-clock split, profiles/r03_pnp_pmc.md: LU = 82 % of pnp_build_solve).  Equivalent form: the column
            // predicate is per THREAD, so it is folded into the pivot-row value once (uz = 0 left of the pivot: e - l*0 == e), and
            // l == 0 needs no test either (e - 0*u == e for finite u; a non-finite u only occurs in a block that is singular /
            // overflowed, whose hypothesis is rejected in both implementations).  2 instructions per row and pivot.
            // (the pivot rows' own triangular solve above uses the same uz: Lsub is 0 where the reference's l == 0 / c >= bw)
            const double *Lpp = Lp + par * 96 * kPanel;
            // `live` is wave-uniform (ty and the pivot rows are), but it is updated under a per-lane-looking condition and hipcc kept
            // it in a VGPR: every row paid v_and + v_cmp + s_and_saveexec + exec restore (62 VALU per wave and panel next to the 128
            // of the ~16 live rows' arithmetic).  From an SGPR the test is s_bitcmp1 + s_cbranch.
            // ... and rows whose four multipliers of this panel are all zero (nzm, from the factor wave): e - 0*u - 0*u - 0*u - 0*u == e
            unsigned nz_s;
            {
                const unsigned long long q0 = nzm[par * 2], q1 = nzm[par * 2 + 1];
                const int ridx = ty + 3 * lane;                        // lane sl < 31 looks up physical row ty + 3 sl
                const bool bit = lane < 31 && (((ridx < 64 ? q0 >> ridx : q1 >> (ridx - 64)) & 1ull) != 0ull);
                nz_s = (unsigned)__builtin_amdgcn_ballot_w64(bit);
            }
            const unsigned live_s = __builtin_amdgcn_readfirstlane(live) & nz_s;
#pragma unroll
            for (int sl = 0; sl < 31; sl++) {
                if (live_s >> sl & 1u) {   // wave-uniform: rows already used as pivots are skipped by a scalar branch
                    const double *lq = Lpp + (ty + 3 * sl) * kPanel;
                    double e = er[sl];
#pragma unroll
                    for (int c = 0; c < kPanel; c++) e = e - lq[c] * uz[c];
                    er[sl] = e;
                }
            }
            const int nc = tx - (k + 2 * kPanel);   // column of panel p+2 held by this thread -> panel[par]
            if (nc >= 0 && nc < kPanel) {
                double *dst = panel + par * 96 * kPanel + nc;
#pragma unroll
                for (int sl = 0; sl < 31; sl++) dst[(ty + 3 * sl) * kPanel] = er[sl];
            }
            if (p + 1 < kPanels) publish_pivot_rows(p + 1);   // underneath the factor wave's work on that very panel
        } else if (p + 1 < kPanels) {
            // -- F: panel p+1 (dumped before panel p was applied) -> apply panel p -> factorise
            const int kn = k + kPanel;
            const int bwn = (kNR - kn) < kPanel ? (kNR - kn) : kPanel;
            const double *pp = panel + (par ^ 1) * 96 * kPanel;
#pragma unroll
            for (int c2 = 0; c2 < kPanel; c2++) {
                // un[c]: pivot row c of panel p at column kn + c2, after the updates inside panel p (column > every step index)
                double un[kPanel];
#pragma unroll
                for (int c = 0; c < kPanel; c++) {
                    double v = prow_raw[(par * kPanel + c) * 128 + kn + c2];
#pragma unroll
                    for (int c3 = 0; c3 < kPanel; c3++)
                        if (c3 < c) v = v - LS(c, c3) * un[c3];   // no l != 0 test (v - 0*u == v)
                    un[c] = v;
                }
                double e0 = pp[lane * kPanel + c2];
                double e1 = has1 ? pp[(lane + 64) * kPanel + c2] : 0.0;
#pragma unroll
                for (int c = 0; c < kPanel; c++) {   // (a panel that is applied to a next one is a full panel: only the last has one column)
                    e0 = e0 - L0(c) * un[c];
                    e1 = e1 - L1(c) * un[c];
                }
                B0(c2) = e0; B1(c2) = e1;
            }
#pragma unroll
            for (int c2 = 0; c2 < kPanel; c2++) { A0(c2) = B0(c2); A1(c2) = B1(c2); }
            if (bwn == kPanel) factor_panel(std::integral_constant<int, kPanel>{}, kn, par ^ 1);
            else factor_panel(std::integral_constant<int, 1>{}, kn, par ^ 1);
        }
        LU_STAMP(2);          // P4 (matrix waves) / F (factor wave)
        __syncthreads();
        LU_STAMP(3);          // wait at the barrier behind P4 / F
    }
    if (lu_stamp) {
        unsigned long long *o = a.stamps + (size_t)slot * 24 + (wave == 0 ? 8 : 12);
        o[0] = lu_acc[0]; o[1] = lu_acc[1]; o[2] = lu_acc[2]; o[3] = lu_acc[3];
    }
#ifdef CHIP_PNP_FSTAMPS
    if (f_stamp && lane == 0) { unsigned long long *o = a.stamps + (size_t)slot * 24 + 16; for (int i = 0; i < 7; i++) o[i] = f_acc[i]; }
#endif
    if (STAMP && lane == 0) {   // which SIMD every wave of the workgroup sits on: HW_ID[5:4] (and the CU: [11:8], SE [14:13])
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11));   // hwreg(HW_ID, 0, 16)
        atomicOr(a.stamps + (size_t)slot * 24 + 23, (unsigned long long)((hw >> 4) & 3u) << (4 * wave) | (unsigned long long)((hw >> 8) & 0xfu) << 32);
    }
#undef F_STAMP
#undef LU_STAMP
    __builtin_amdgcn_s_setprio(0);
    if (singular) {
        if (tid == 0) a.ok[slot] = 0;
        return;
    }
    SOLVE_STAMP(4);   // LU done
    // ---- back-substitution: only the last 27 unknowns (boundary monomials) are referenced by B; Urows[i-66] = row i of U ----
    // One thread per right-hand side; the 27 x 26 / 2 terms of a column are one serial chain (the reference's order).  Fully unrolled
    // with the solved unknowns in REGISTERS: the rolled form read U and X from LDS term by term, ~70 cycles of exposed LDS latency
    // per term (25 k cycles per workgroup, 11 % of the kernel); here the U reads (wave-uniform broadcasts) carry no dependence on
    // the chain and are issued ahead of it, and a term costs its multiply and its dependent subtract.
    if (wave == 0) {
        const int c = lane < 27 ? lane : 0;   // lanes >= 27 repeat column 0 (no divergence, no store)
        double xs[27];
#pragma unroll
        for (int r = 26; r >= 0; r--) {       // row i = 66 + r of U
            const double *Ui = Urows + r * kNC;
            double s = Ui[93 + c];
#pragma unroll
            for (int q = r + 1; q < 27; q++) s = s - Ui[66 + q] * xs[q];
            xs[r] = s / Ui[66 + r];
        }
        if (lane < 27) {
#pragma unroll
            for (int r = 0; r < 27; r++) Xb[r * 27 + c] = xs[r];
        }
    }
    __syncthreads();
    SOLVE_STAMP(5);   // back-substitution done
    // ---- S = A - B X ----
    for (int e = tid; e < 729; e += kSolveThreads) {
        const int r = e / 27, j = e % 27;
        double s = 0.0;
        for (int t = 0; t < 4; t++)
            if (tb.s_col[r][t] == j) s = s + uu[t];
        for (int t = 1; t < 4; t++) {
            const int col = tb.s_col[r][t];
            if (col >= 27) s = s - uu[t] * Xb[(col - 93) * 27 + j];
        }
        a.Sg[(size_t)slot * 729 + e] = s;
    }
    if (tid == 0) a.ok[slot] = 1;
    SOLVE_STAMP(6);
#undef SOLVE_STAMP
}

// ------------------------------------------------------------------------------------------------ K5b + K6
constexpr int EN = 27;

// IEEE-754 double sqrt for an argument in the normal range that needs no pre-scaling (2^-767 <= t < inf, not NaN): the SAME
// instruction sequence hipcc emits for sqrt(double) -- v_rsq_f64 seed, one coupled Goldschmidt step, two residual corrections --
// minus its scaling of tiny arguments (v_cmp / v_ldexp in, v_ldexp out) and its zero / inf / NaN pass-through (v_cmp_class, two
// v_cndmask): three dependent instructions fewer on the critical chain of every double-shift QR step.  Bit-identical to sqrt()
// on that range (tests/test_pnp_gpu.py compares poses bit for bit with the CPU oracle, which calls libm's correctly rounded sqrt).
__device__ __forceinline__ double sqrt_normal_range(double t)
{
    const double y = __builtin_amdgcn_rsq(t);
    double g = t * y;
    double h = y * 0.5;
    const double r0 = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r0, g);
    h = __builtin_fma(h, r0, h);
    const double d0 = __builtin_fma(-g, g, t);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, t);
    return __builtin_fma(d1, h, g);
}



// ---- serial sums / updates of the Hessenberg reduction (orthes / ortran), one wave, operands in LDS -------------------------------
// A lone wave issues one instruction per ~8 cycles and pays ~100 cycles for every LDS read it waits for (profiles/r03_chain_latency.txt),
// so these loops (2800 terms per matrix, each a multiply and a dependent add) are written to (1) read the reflector u from LDS as
// wave-uniform broadcasts -- one DS read instead of two v_readlane per term --, and (2) keep the NEXT four terms' reads in flight
// underneath the current four terms' arithmetic (LDS returns in order: counted lgkmcnt waits).  Term order is the reference's.
//   dot_desc:  f = sum_{i = hi, hi-1, .., lo} a[i] * b[i * SB]     (f starts at +0.0, one rounding per multiply and per add)
// The n % 4 leading terms of a serial sum are straight-line code behind ONE wave-uniform switch (n is uniform: it derives from the
// reduction step m), the rest runs in whole groups of four.  Round 3 handled the 0..3 left-over terms at the END with three guarded
// blocks that hipcc if-converted into ~35-45 executed instructions per sum whatever the remainder was.  (Measured on the same box,
// round 4: 1000-hypothesis call 884 -> 866 us, reference mode 623 -> 611 us, bits unchanged.  Forcing the next group's LDS reads in
// front of the current group's arithmetic with sched_barrier + two alternating register sets was measured too: SLOWER by 0.5-1 %
// -- these loops are bound by their instruction count, ~8.5 cycles per instruction of a lone wave, not by LDS latency.)
template <int SB>
__device__ __forceinline__ double dot_desc(const double *a, const double *b, int hi, int lo)
{
    double f = 0.0;
    int i = hi;
    switch ((hi - lo + 1) & 3) {
        case 3: {
            const double a0 = a[i], a1 = a[i - 1], a2 = a[i - 2];
            const double b0 = b[i * SB], b1 = b[(i - 1) * SB], b2 = b[(i - 2) * SB];
            f = f + a0 * b0; f = f + a1 * b1; f = f + a2 * b2;
            i -= 3;
            break;
        }
        case 2: {
            const double a0 = a[i], a1 = a[i - 1];
            const double b0 = b[i * SB], b1 = b[(i - 1) * SB];
            f = f + a0 * b0; f = f + a1 * b1;
            i -= 2;
            break;
        }
        case 1: {
            const double a0 = a[i], b0 = b[i * SB];
            f = f + a0 * b0;
            i -= 1;
            break;
        }
        default: break;
    }
    if (i - 3 >= lo) {
        double a0 = a[i], a1 = a[i - 1], a2 = a[i - 2], a3 = a[i - 3];
        double b0 = b[i * SB], b1 = b[(i - 1) * SB], b2 = b[(i - 2) * SB], b3 = b[(i - 3) * SB];
        i -= 4;
        while (i - 3 >= lo) {
            const double na0 = a[i], na1 = a[i - 1], na2 = a[i - 2], na3 = a[i - 3];
            const double nb0 = b[i * SB], nb1 = b[(i - 1) * SB], nb2 = b[(i - 2) * SB], nb3 = b[(i - 3) * SB];
            f = f + a0 * b0; f = f + a1 * b1; f = f + a2 * b2; f = f + a3 * b3;
            a0 = na0; a1 = na1; a2 = na2; a3 = na3; b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
            i -= 4;
        }
        f = f + a0 * b0; f = f + a1 * b1; f = f + a2 * b2; f = f + a3 * b3;
    }
    return f;
}
//   dot_asc:  g = sum_{i = lo, lo+1, .., hi} a[i] * b[i * SB]
template <int SB>
__device__ __forceinline__ double dot_asc(const double *a, const double *b, int lo, int hi)
{
    double f = 0.0;
    int i = lo;
    switch ((hi - lo + 1) & 3) {
        case 3: {
            const double a0 = a[i], a1 = a[i + 1], a2 = a[i + 2];
            const double b0 = b[i * SB], b1 = b[(i + 1) * SB], b2 = b[(i + 2) * SB];
            f = f + a0 * b0; f = f + a1 * b1; f = f + a2 * b2;
            i += 3;
            break;
        }
        case 2: {
            const double a0 = a[i], a1 = a[i + 1];
            const double b0 = b[i * SB], b1 = b[(i + 1) * SB];
            f = f + a0 * b0; f = f + a1 * b1;
            i += 2;
            break;
        }
        case 1: {
            const double a0 = a[i], b0 = b[i * SB];
            f = f + a0 * b0;
            i += 1;
            break;
        }
        default: break;
    }
    if (i + 3 <= hi) {
        double a0 = a[i], a1 = a[i + 1], a2 = a[i + 2], a3 = a[i + 3];
        double b0 = b[i * SB], b1 = b[(i + 1) * SB], b2 = b[(i + 2) * SB], b3 = b[(i + 3) * SB];
        i += 4;
        while (i + 3 <= hi) {
            const double na0 = a[i], na1 = a[i + 1], na2 = a[i + 2], na3 = a[i + 3];
            const double nb0 = b[i * SB], nb1 = b[(i + 1) * SB], nb2 = b[(i + 2) * SB], nb3 = b[(i + 3) * SB];
            f = f + a0 * b0; f = f + a1 * b1; f = f + a2 * b2; f = f + a3 * b3;
            a0 = na0; a1 = na1; a2 = na2; a3 = na3; b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
            i += 4;
        }
        f = f + a0 * b0; f = f + a1 * b1; f = f + a2 * b2; f = f + a3 * b3;
    }
    return f;
}
//   abs_sum_asc:  acc + |b[lo * SB]| + |b[(lo+1) * SB]| + .. + |b[hi * SB]|, ascending, next four reads in flight under the current four adds
template <int SB>
__device__ __forceinline__ double abs_sum_asc(const double *b, int lo, int hi, double acc)
{
    int i = lo;
    switch ((hi - lo + 1) & 3) {
        case 3: { const double c0 = b[i * SB], c1 = b[(i + 1) * SB], c2 = b[(i + 2) * SB]; acc = acc + fabs(c0); acc = acc + fabs(c1); acc = acc + fabs(c2); i += 3; break; }
        case 2: { const double c0 = b[i * SB], c1 = b[(i + 1) * SB]; acc = acc + fabs(c0); acc = acc + fabs(c1); i += 2; break; }
        case 1: { const double c0 = b[i * SB]; acc = acc + fabs(c0); i += 1; break; }
        default: break;
    }
    if (i + 3 <= hi) {
        double b0 = b[i * SB], b1 = b[(i + 1) * SB], b2 = b[(i + 2) * SB], b3 = b[(i + 3) * SB];
        i += 4;
        while (i + 3 <= hi) {
            const double n0 = b[i * SB], n1 = b[(i + 1) * SB], n2 = b[(i + 2) * SB], n3 = b[(i + 3) * SB];
            acc = acc + fabs(b0); acc = acc + fabs(b1); acc = acc + fabs(b2); acc = acc + fabs(b3);
            b0 = n0; b1 = n1; b2 = n2; b3 = n3;
            i += 4;
        }
        acc = acc + fabs(b0); acc = acc + fabs(b1); acc = acc + fabs(b2); acc = acc + fabs(b3);
    }
    return acc;
}
//   axpy_rows:  b[i * SB] = b[i * SB] + c * a[i],  i = lo .. hi  (independent per i; c = -f gives the reference's  b - f * a:
//   x - y == x + (-y) and (-f) * a == -(f * a) exactly in IEEE arithmetic)
template <int SB>
__device__ __forceinline__ void axpy_rows(const double *a, double *b, double c, int lo, int hi)
{
    int i = lo;
    switch ((hi - lo + 1) & 3) {
        case 3: {
            const double a0 = a[i], a1 = a[i + 1], a2 = a[i + 2];
            const double b0 = b[i * SB], b1 = b[(i + 1) * SB], b2 = b[(i + 2) * SB];
            b[i * SB] = b0 + c * a0; b[(i + 1) * SB] = b1 + c * a1; b[(i + 2) * SB] = b2 + c * a2;
            i += 3;
            break;
        }
        case 2: {
            const double a0 = a[i], a1 = a[i + 1];
            const double b0 = b[i * SB], b1 = b[(i + 1) * SB];
            b[i * SB] = b0 + c * a0; b[(i + 1) * SB] = b1 + c * a1;
            i += 2;
            break;
        }
        case 1: {
            const double a0 = a[i], b0 = b[i * SB];
            b[i * SB] = b0 + c * a0;
            i += 1;
            break;
        }
        default: break;
    }
    for (; i + 3 <= hi; i += 4) {
        const double a0 = a[i], a1 = a[i + 1], a2 = a[i + 2], a3 = a[i + 3];
        const double b0 = b[i * SB], b1 = b[(i + 1) * SB], b2 = b[(i + 2) * SB], b3 = b[(i + 3) * SB];
        b[i * SB] = b0 + c * a0; b[(i + 1) * SB] = b1 + c * a1; b[(i + 2) * SB] = b2 + c * a2; b[(i + 3) * SB] = b3 + c * a3;
    }
}

//   sumsq_desc:  h = sum_{i = hi, hi-1, .., lo} u[i]^2  (wave-uniform broadcast reads, two elements per ds_read2)
__device__ __forceinline__ double sumsq_desc(const double *u, int hi, int lo)
{
    double h = 0.0;
    int i = hi;
    switch ((hi - lo + 1) & 3) {
        case 3: { const double o0 = u[i], o1 = u[i - 1], o2 = u[i - 2]; h = h + o0 * o0; h = h + o1 * o1; h = h + o2 * o2; i -= 3; break; }
        case 2: { const double o0 = u[i], o1 = u[i - 1]; h = h + o0 * o0; h = h + o1 * o1; i -= 2; break; }
        case 1: { const double o0 = u[i]; h = h + o0 * o0; i -= 1; break; }
        default: break;
    }
    for (; i - 3 >= lo; i -= 4) {
        const double o0 = u[i], o1 = u[i - 1], o2 = u[i - 2], o3 = u[i - 3];
        h = h + o0 * o0; h = h + o1 * o1; h = h + o2 * o2; h = h + o3 * o3;
    }
    return h;
}

// a wave-uniform double as an SGPR pair (asm "s" operands must not be handed a VGPR: the compiler does not insert the readfirstlane itself)
__device__ __forceinline__ double uniform_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ uint32_t lds_addr32(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }

// The three-row double-shift steps of one Francis sweep with a forwarded reflector, k = k0 .. k1 <= n-2, as ONE hand-scheduled
// instruction stream (round 4).  Same operations on the same values in the same order as qr_step<NOTLAST = true> with k != m -- so
// the same bits -- but:
//   * the activity masks of the row / column modification live in SGPR pairs advanced by SALU bit instructions (the compiled form
//     crossed from the vector to the scalar pipe three times per step: v_cmp -> s_and_saveexec -> s_cbranch_execz, ~30 cycles each
//     for a lone wave) and exec is simply written, not saved / restored;
//   * the test |p|+|q|+|r| == 0 is issued at the top and branched on five instructions later;
//   * the scalar bookkeeping of step k+1 (counters, masks) sits between the issue of the column modification's LDS reads and the
//     wait for them -- the one LDS write -> read round trip of a step that no layout removes;
//   * the wait states of the IEEE division (transcendental -> consumer, vcc -> v_div_fmas) are filled with address bumps and the
//     sub-diagonal value instead of s_nop; one v_mov_b64 instead of three v_mov at the top (the scaled p and r come out of
//     v_ldexp straight from the SGPR pairs the previous step's v_readlane left them in); addresses advance by one v_add each.
//   * H(k+3, k) of the step k = n-2 is 0: its v_readlane reads lane n+1, whose c0 register no step writes (lanes > n are masked
//     out of the column modification) and which is zeroed on entry.
// ~112 instructions per step against ~140 compiled.  Registers v72..v127 and s40..s69 belong to this block (clobber list: the
// compiler keeps nothing there across it); wait states follow the compiler's own code for the same sequences on gfx950 (one state
// between a transcendental and its consumer, four between a VALU write of vcc and v_div_fmas, one between a VALU write and
// v_readlane).
//   in : k0 <= k1, (p, q, r) = H(k0.., k0-1) as forwarded (or from the m search: first63 = 0 masks the first step's H(k, k-1) store), LDS byte addresses of HH(k0, lane), A(arow, k0), HH(k0, k0-1), lane masks
//   do_last (k1 == n-2 only): after step n-2 also run the sweep's LAST, two-row step k = n-1 (k comes back as n); zero = 2: that step
//   found |p|+|q|+|r| == 0 and was not executed (k comes back as n-1)
//   out: k = first step NOT executed (k1 + 1, or the step whose |p|+|q|+|r| is 0: zero = 1), (p, q, r) for that step
__device__ __forceinline__ void qr_steps_asm(int &k, int k1, double &p, double &q, double &r, uint32_t rowaddr, uint32_t coladdr,
                                             uint32_t subaddr, uint64_t rowmask, uint64_t colmask, uint64_t nmask, uint64_t first63, int do_last, int &zero)
{
    // The block hard-codes the LDS geometry of the 27 x 27 matrices: row stride 0xd8 = 27 * 8 bytes, diagonal stride
    // 0xe0 = 28 * 8, ds_read2_b64 offsets 27 / 54 / 81 (in 8-byte units: one / two / three rows down).  The row address is kept ONE
    // ROW BEHIND and read at +216 (the bump sits in a division's wait states): for k0 = 0 it starts 216 bytes below HH(0, lane),
    // which is inside the kernel's HsPad (28 doubles = 224 bytes in front of Hs), never below LDS address 0.
    static_assert(EN == 27, "qr_steps_asm: LDS strides 0xd8 / 0xe0 and the ds_read2 offsets 27 / 54 / 81 are those of EN = 27");
    static_assert(EN * 8 == 0xd8 && (EN + 1) * 8 == 0xe0, "qr_steps_asm: row / diagonal stride literals");
    const uint64_t m_odd = 0xAAAAAAAAAAAAAAAAull, m_l0 = 1ull, m_lt3 = 7ull, m_l63 = 1ull << 63;
    int kk = k, z = 0;
    double po = p, qo = q, ro = r;
    asm volatile(
        "s_mov_b64 s[66:67], exec\n\t"
        "s_mov_b64 s[40:41], %[p]\n\t"
        "s_mov_b64 s[42:43], %[q]\n\t"
        "s_mov_b64 s[44:45], %[r]\n\t"
        "s_mov_b32 s56, %[k]\n\t"
        "s_mov_b32 s57, %[k1]\n\t"
        "s_add_u32 s58, s56, 1\n\t"
        "s_add_u32 s59, s56, 2\n\t"
        "s_add_u32 s60, s56, 3\n\t"
        "s_mov_b64 s[62:63], %[rowmask]\n\t"
        "s_mov_b64 s[64:65], %[colmask]\n\t"
        "s_mov_b64 s[70:71], %[first63]\n\t"                                // lanes that store H(k, k-1) in the FIRST step run here (none: the caller does it)
        "v_subrev_u32 v72, 0xd8, %[rowaddr]\n\t"                           // one row behind: reads at +216.., bumped before the writes
        "v_mov_b32 v73, %[coladdr]\n\t"
        "s_sub_u32 s61, %[subaddr], 0xe0\n\t"
        "v_mov_b32 v74, s61\n\t"                                            // one step behind, bumped before its write
        "v_mov_b32 v118, 0\n\t"
        "v_mov_b32 v119, 0\n\t"
        "s_mov_b32 %[z], 0\n"
        "Lqr_step_%=:\n\t"
        // row-modification inputs H(k..k+2, lane): in flight underneath the reflector chain
        "ds_read_b64 v[80:81], v72 offset:216\n\t"
        "ds_read2_b64 v[82:85], v72 offset0:54 offset1:81\n\t"
        // reflector: x = |p| + |q| + |r|, exact power-of-two scaling, s = sqrt(p^2 + q^2 + r^2) with the sign of p
        "v_mov_b64 v[88:89], s[42:43]\n\t"
        "v_add_f64 v[92:93], |s[40:41]|, |v[88:89]|\n\t"
        "v_add_f64 v[92:93], v[92:93], |s[44:45]|\n\t"
        "v_cmp_eq_f64 vcc, 0, v[92:93]\n\t"
        "v_frexp_exp_i32_f64 v94, v[92:93]\n\t"
        "v_sub_u32 v95, 0, v94\n\t"
        "v_ldexp_f64 v[86:87], s[40:41], v95\n\t"
        "v_ldexp_f64 v[88:89], v[88:89], v95\n\t"
        "v_ldexp_f64 v[90:91], s[44:45], v95\n\t"
        "s_cbranch_vccnz Lqr_zero_%=\n\t"
        "v_mul_f64 v[96:97], v[86:87], v[86:87]\n\t"
        "v_mul_f64 v[98:99], v[88:89], v[88:89]\n\t"
        "v_add_f64 v[96:97], v[96:97], v[98:99]\n\t"
        "v_mul_f64 v[98:99], v[90:91], v[90:91]\n\t"
        "v_add_f64 v[100:101], v[96:97], v[98:99]\n\t"
        "v_rsq_f64 v[102:103], v[100:101]\n\t"
        "v_cmp_gt_f64 vcc, 0, v[86:87]\n\t"
        "v_mul_f64 v[104:105], v[100:101], v[102:103]\n\t"
        "v_mul_f64 v[106:107], v[102:103], 0.5\n\t"
        "v_fma_f64 v[108:109], -v[106:107], v[104:105], 0.5\n\t"
        "v_fma_f64 v[104:105], v[104:105], v[108:109], v[104:105]\n\t"
        "v_fma_f64 v[106:107], v[106:107], v[108:109], v[106:107]\n\t"
        "v_fma_f64 v[108:109], -v[104:105], v[104:105], v[100:101]\n\t"
        "v_fma_f64 v[104:105], v[108:109], v[106:107], v[104:105]\n\t"
        "v_fma_f64 v[108:109], -v[104:105], v[104:105], v[100:101]\n\t"
        "v_fma_f64 v[110:111], v[108:109], v[106:107], v[104:105]\n\t"
        "v_xor_b32 v75, 0x80000000, v111\n\t"
        "v_cndmask_b32 v111, v111, v75, vcc\n\t"
        "v_add_f64 v[86:87], v[86:87], v[110:111]\n\t"                       // p = p + s
        // x = p/s, y = q/s, z = r/s, q = q/p, r = r/p as one vector division over lanes 0..4
        "v_cndmask_b32 v114, v90, v88, %[modd]\n\t"
        "v_cndmask_b32 v115, v91, v89, %[modd]\n\t"
        "v_cndmask_b32 v114, v114, v86, %[ml0]\n\t"
        "v_cndmask_b32 v115, v115, v87, %[ml0]\n\t"
        "v_cndmask_b32 v116, v86, v110, %[mlt3]\n\t"
        "v_cndmask_b32 v117, v87, v111, %[mlt3]\n\t"
        "v_div_scale_f64 v[76:77], s[68:69], v[116:117], v[116:117], v[114:115]\n\t"
        "v_rcp_f64 v[120:121], v[76:77]\n\t"
        "v_add_u32 v72, 0xd8, v72\n\t"                                       // (wait state) row address -> HH(k, lane)
        "v_fma_f64 v[122:123], -v[76:77], v[120:121], 1.0\n\t"
        "v_fma_f64 v[120:121], v[120:121], v[122:123], v[120:121]\n\t"
        "v_fma_f64 v[122:123], -v[76:77], v[120:121], 1.0\n\t"
        "v_fma_f64 v[120:121], v[120:121], v[122:123], v[120:121]\n\t"
        "v_div_scale_f64 v[124:125], vcc, v[114:115], v[116:117], v[114:115]\n\t"
        "v_mul_f64 v[126:127], v[124:125], v[120:121]\n\t"
        "v_fma_f64 v[76:77], -v[76:77], v[126:127], v[124:125]\n\t"
        "v_ldexp_f64 v[112:113], -v[110:111], v94\n\t"                       // (wait state) H(k, k-1) = -s * 2^ex
        "v_add_u32 v74, 0xe0, v74\n\t"                                       // (wait state) its address
        "v_div_fmas_f64 v[76:77], v[76:77], v[120:121], v[126:127]\n\t"
        "v_div_fixup_f64 v[96:97], v[76:77], v[116:117], v[114:115]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 s46, v96, 0\n\t"
        "v_readlane_b32 s47, v97, 0\n\t"
        "v_readlane_b32 s48, v96, 1\n\t"
        "v_readlane_b32 s49, v97, 1\n\t"
        "v_readlane_b32 s50, v96, 2\n\t"
        "v_readlane_b32 s51, v97, 2\n\t"
        "v_readlane_b32 s52, v96, 3\n\t"
        "v_readlane_b32 s53, v97, 3\n\t"
        "v_readlane_b32 s54, v96, 4\n\t"
        "v_readlane_b32 s55, v97, 4\n\t"
        // H(k, k-1) by lane 63
        "s_mov_b64 exec, s[70:71]\n\t"
        "ds_write_b64 v74, v[112:113]\n\t"
        // row modification, column j = lane, lanes k..26
        "s_mov_b64 exec, s[62:63]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_mul_f64 v[98:99], s[52:53], v[82:83]\n\t"
        "v_add_f64 v[96:97], v[80:81], v[98:99]\n\t"
        "v_mul_f64 v[98:99], s[54:55], v[84:85]\n\t"
        "v_add_f64 v[96:97], v[96:97], v[98:99]\n\t"
        "v_mul_f64 v[98:99], v[96:97], s[50:51]\n\t"
        "v_add_f64 v[104:105], v[84:85], -v[98:99]\n\t"
        "v_mul_f64 v[98:99], v[96:97], s[46:47]\n\t"
        "v_add_f64 v[100:101], v[80:81], -v[98:99]\n\t"
        "v_mul_f64 v[98:99], v[96:97], s[48:49]\n\t"
        "v_add_f64 v[102:103], v[82:83], -v[98:99]\n\t"
        "ds_write_b64 v72, v[100:101]\n\t"
        "ds_write2_b64 v72, v[102:103], v[104:105] offset0:27 offset1:54\n\t"
        // column modification: H rows 0..min(n, k+3) (lanes 0..) and V rows 0..26 (lanes 32..58), one stream
        "s_mov_b64 exec, s[64:65]\n\t"
        "ds_read2_b64 v[106:109], v73 offset1:1\n\t"
        "ds_read_b64 v[110:111], v73 offset:16\n\t"
        // ... while those reads make their round trip: counters and masks of step k + 1
        "s_bitset0_b64 s[62:63], s56\n\t"
        "s_add_u32 s56, s56, 1\n\t"
        "s_add_u32 s58, s58, 1\n\t"
        "s_add_u32 s59, s59, 1\n\t"
        "s_add_u32 s60, s60, 1\n\t"
        "s_bitset1_b64 s[64:65], s60\n\t"
        "s_and_b64 s[64:65], s[64:65], %[nmask]\n\t"
        "s_mov_b64 s[70:71], %[ml63]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_mul_f64 v[114:115], s[46:47], v[106:107]\n\t"
        "v_mul_f64 v[116:117], s[48:49], v[108:109]\n\t"
        "v_add_f64 v[112:113], v[114:115], v[116:117]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f64 v[114:115], s[50:51], v[110:111]\n\t"
        "v_add_f64 v[112:113], v[112:113], v[114:115]\n\t"
        "v_add_f64 v[118:119], v[106:107], -v[112:113]\n\t"
        "v_mul_f64 v[114:115], v[112:113], s[54:55]\n\t"
        "v_add_f64 v[122:123], v[110:111], -v[114:115]\n\t"
        "v_mul_f64 v[116:117], v[112:113], s[52:53]\n\t"
        "v_add_f64 v[120:121], v[108:109], -v[116:117]\n\t"
        "ds_write_b64 v73, v[122:123] offset:16\n\t"
        "ds_write2_b64 v73, v[118:119], v[120:121] offset1:1\n\t"
        "s_mov_b64 exec, s[66:67]\n\t"
        // the next reflector comes from H(k+1..k+3, k) as just computed by lanes k+1..k+3 (counters are already those of step k + 1)
        "v_readlane_b32 s40, v118, s56\n\t"
        "v_readlane_b32 s41, v119, s56\n\t"
        "v_readlane_b32 s42, v118, s58\n\t"
        "v_readlane_b32 s43, v119, s58\n\t"
        "v_readlane_b32 s44, v118, s59\n\t"
        "v_readlane_b32 s45, v119, s59\n\t"
        "v_add_u32 v73, 8, v73\n\t"
        "s_cmp_le_u32 s56, s57\n\t"
        "s_cbranch_scc1 Lqr_step_%=\n\t"
        // ---- the LAST step of the sweep, k = n-1: the reflector spans two rows (r = 0 arrives as the forwarded H(k+3, k) of step n-2);
        //      same arithmetic as qr_step<NOTLAST = false>: no third row / column, nothing forwarded
        "s_cmp_eq_u32 %[dolast], 0\n\t"
        "s_cbranch_scc1 Lqr_done_%=\n\t"
        "ds_read_b64 v[80:81], v72 offset:216\n\t"
        "ds_read_b64 v[82:83], v72 offset:432\n\t"
        "v_mov_b64 v[88:89], s[42:43]\n\t"
        "v_add_f64 v[92:93], |s[40:41]|, |v[88:89]|\n\t"
        "v_add_f64 v[92:93], v[92:93], |s[44:45]|\n\t"
        "v_cmp_eq_f64 vcc, 0, v[92:93]\n\t"
        "v_frexp_exp_i32_f64 v94, v[92:93]\n\t"
        "v_sub_u32 v95, 0, v94\n\t"
        "v_ldexp_f64 v[86:87], s[40:41], v95\n\t"
        "v_ldexp_f64 v[88:89], v[88:89], v95\n\t"
        "v_ldexp_f64 v[90:91], s[44:45], v95\n\t"
        "s_cbranch_vccnz Lqr_zero_last_%=\n\t"
        "v_mul_f64 v[96:97], v[86:87], v[86:87]\n\t"
        "v_mul_f64 v[98:99], v[88:89], v[88:89]\n\t"
        "v_add_f64 v[96:97], v[96:97], v[98:99]\n\t"
        "v_mul_f64 v[98:99], v[90:91], v[90:91]\n\t"
        "v_add_f64 v[100:101], v[96:97], v[98:99]\n\t"
        "v_rsq_f64 v[102:103], v[100:101]\n\t"
        "v_cmp_gt_f64 vcc, 0, v[86:87]\n\t"
        "v_mul_f64 v[104:105], v[100:101], v[102:103]\n\t"
        "v_mul_f64 v[106:107], v[102:103], 0.5\n\t"
        "v_fma_f64 v[108:109], -v[106:107], v[104:105], 0.5\n\t"
        "v_fma_f64 v[104:105], v[104:105], v[108:109], v[104:105]\n\t"
        "v_fma_f64 v[106:107], v[106:107], v[108:109], v[106:107]\n\t"
        "v_fma_f64 v[108:109], -v[104:105], v[104:105], v[100:101]\n\t"
        "v_fma_f64 v[104:105], v[108:109], v[106:107], v[104:105]\n\t"
        "v_fma_f64 v[108:109], -v[104:105], v[104:105], v[100:101]\n\t"
        "v_fma_f64 v[110:111], v[108:109], v[106:107], v[104:105]\n\t"
        "v_xor_b32 v75, 0x80000000, v111\n\t"
        "v_cndmask_b32 v111, v111, v75, vcc\n\t"
        "v_add_f64 v[86:87], v[86:87], v[110:111]\n\t"
        "v_cndmask_b32 v114, v90, v88, %[modd]\n\t"
        "v_cndmask_b32 v115, v91, v89, %[modd]\n\t"
        "v_cndmask_b32 v114, v114, v86, %[ml0]\n\t"
        "v_cndmask_b32 v115, v115, v87, %[ml0]\n\t"
        "v_cndmask_b32 v116, v86, v110, %[mlt3]\n\t"
        "v_cndmask_b32 v117, v87, v111, %[mlt3]\n\t"
        "v_div_scale_f64 v[76:77], s[68:69], v[116:117], v[116:117], v[114:115]\n\t"
        "v_rcp_f64 v[120:121], v[76:77]\n\t"
        "v_add_u32 v72, 0xd8, v72\n\t"
        "v_fma_f64 v[122:123], -v[76:77], v[120:121], 1.0\n\t"
        "v_fma_f64 v[120:121], v[120:121], v[122:123], v[120:121]\n\t"
        "v_fma_f64 v[122:123], -v[76:77], v[120:121], 1.0\n\t"
        "v_fma_f64 v[120:121], v[120:121], v[122:123], v[120:121]\n\t"
        "v_div_scale_f64 v[124:125], vcc, v[114:115], v[116:117], v[114:115]\n\t"
        "v_mul_f64 v[126:127], v[124:125], v[120:121]\n\t"
        "v_fma_f64 v[76:77], -v[76:77], v[126:127], v[124:125]\n\t"
        "v_ldexp_f64 v[112:113], -v[110:111], v94\n\t"
        "v_add_u32 v74, 0xe0, v74\n\t"
        "v_div_fmas_f64 v[76:77], v[76:77], v[120:121], v[126:127]\n\t"
        "v_div_fixup_f64 v[96:97], v[76:77], v[116:117], v[114:115]\n\t"
        "s_nop 0\n\t"
        "v_readlane_b32 s46, v96, 0\n\t"
        "v_readlane_b32 s47, v97, 0\n\t"
        "v_readlane_b32 s48, v96, 1\n\t"
        "v_readlane_b32 s49, v97, 1\n\t"
        "v_readlane_b32 s52, v96, 3\n\t"
        "v_readlane_b32 s53, v97, 3\n\t"
        "s_mov_b64 exec, s[70:71]\n\t"
        "ds_write_b64 v74, v[112:113]\n\t"
        "s_mov_b64 exec, s[62:63]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t"
        "v_mul_f64 v[98:99], s[52:53], v[82:83]\n\t"
        "v_add_f64 v[96:97], v[80:81], v[98:99]\n\t"                       // pp = h0 + q * h1
        "v_mul_f64 v[98:99], v[96:97], s[46:47]\n\t"
        "v_add_f64 v[100:101], v[80:81], -v[98:99]\n\t"
        "v_mul_f64 v[98:99], v[96:97], s[48:49]\n\t"
        "v_add_f64 v[102:103], v[82:83], -v[98:99]\n\t"
        "ds_write2_b64 v72, v[100:101], v[102:103] offset1:27\n\t"
        "s_mov_b64 exec, s[64:65]\n\t"
        "ds_read2_b64 v[106:109], v73 offset1:1\n\t"
        "s_add_u32 s56, s56, 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f64 v[114:115], s[46:47], v[106:107]\n\t"
        "v_mul_f64 v[116:117], s[48:49], v[108:109]\n\t"
        "v_add_f64 v[112:113], v[114:115], v[116:117]\n\t"                 // pp = x * a0 + y * a1
        "v_add_f64 v[118:119], v[106:107], -v[112:113]\n\t"
        "v_mul_f64 v[116:117], v[112:113], s[52:53]\n\t"
        "v_add_f64 v[120:121], v[108:109], -v[116:117]\n\t"
        "ds_write2_b64 v73, v[118:119], v[120:121] offset1:1\n\t"
        "s_branch Lqr_done_%=\n"
        "Lqr_zero_last_%=:\n\t"
        "s_mov_b32 %[z], 2\n\t"
        "s_branch Lqr_done_%=\n"
        "Lqr_zero_%=:\n\t"
        "s_mov_b32 %[z], 1\n"
        "Lqr_done_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 exec, s[66:67]\n\t"
        "s_mov_b32 %[kout], s56\n\t"
        "s_mov_b64 %[pout], s[40:41]\n\t"
        "s_mov_b64 %[qout], s[42:43]\n\t"
        "s_mov_b64 %[rout], s[44:45]"
        : [z] "=&s"(z), [kout] "=&s"(kk), [pout] "=&s"(po), [qout] "=&s"(qo), [rout] "=&s"(ro)
        : [p] "s"(p), [q] "s"(q), [r] "s"(r), [k] "s"(k), [k1] "s"(k1), [subaddr] "s"(subaddr), [rowmask] "s"(rowmask),
          [colmask] "s"(colmask), [nmask] "s"(nmask), [rowaddr] "v"(rowaddr), [coladdr] "v"(coladdr), [modd] "s"(m_odd), [ml0] "s"(m_l0),
          [mlt3] "s"(m_lt3), [ml63] "s"(m_l63), [first63] "s"(first63), [dolast] "s"(do_last)
        : "memory", "vcc", "scc", "s70", "s71", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
          "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "v72", "v73", "v74", "v75", "v76",
          "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
          "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
          "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    k = kk; zero = z; p = po; q = qo; r = ro;
}

// The head of one Francis sweep on the common path (no deflation this sweep, no exceptional shift) as one hand-scheduled block:
// the deflation test (l search), the shifts x, y, w, the search for two consecutive small sub-diagonal elements (m search: p, q, r of
// the first reflector, scaled by an exact power of two) and the zeroing of H(i, i-2), H(i, i-3) below the bulge's start.  Same
// per-lane arithmetic and the same selections (largest passing lane) as the compiled code it replaces, which read its operands with
// ten LDS loads behind four exec-masked regions and two ballots (~2 k cycles per sweep): here every lane loads the seven entries of its
// 3 x 3 diagonal neighbourhood with four LDS instructions off one base address, together with the four entries the shifts need, all
// lanes compute, and the valid-lane ranges are applied to the compare masks by SALU.  Registers v72..v127 / s40..s71 as qr_steps_asm.
//   status 0: proceed with the double-shift steps from m (l, m, p, q, r valid, zeroing done)
//   status 1: l >= n - 1 (one or two roots to deflate): nothing was written, the caller's compiled path handles the sweep
__device__ __forceinline__ void qr_sweep_head_asm(int n, uint32_t hs0, int lane, double norm, int &status, int &l_out, int &m_out,
                                                  double &p, double &q, double &r)
{
    const uint32_t base = hs0 + (uint32_t)lane * 224u - 224u;     // &HH(lane - 1, lane - 1): every operand of lane `lane` is base + const
    const uint32_t nbase = hs0 + (uint32_t)(n - 1) * 224u;        // &HH(n - 1, n - 1)
    const uint64_t lvalid = ((1ull << (n + 1)) - 1) & ~1ull;      // lanes 1..n take part in the l search
    int st = 0, lo = 0, mo = 0;
    double po = 0.0, qo = 0.0, ro = 0.0;
    asm volatile(
        "s_mov_b64 s[66:67], exec\n\t"
        "v_mov_b32 v72, %[base]\n\t"
        "v_mov_b32 v73, %[nbase]\n\t"
        // per lane i: dm1 = H(i-1,i-1) [v80], sub = H(i,i-1) [v82], d0 = H(i,i) [v84], sup = H(i,i+1) [v86],
        //             subp1 = H(i+1,i) [v88], dp1 = H(i+1,i+1) [v90], subp2 = H(i+2,i+1) [v92];  uniform: y, H(n-1,n) [v94..97], H(n,n-1), x [v98..101]
        "ds_read2_b64 v[80:83], v72 offset1:27\n\t"
        "ds_read2_b64 v[84:87], v72 offset0:28 offset1:29\n\t"
        "ds_read2_b64 v[88:91], v72 offset0:55 offset1:56\n\t"
        "ds_read_b64 v[92:93], v72 offset:664\n\t"
        "ds_read2_b64 v[94:97], v73 offset1:1\n\t"
        "ds_read2_b64 v[98:101], v73 offset0:27 offset1:28\n\t"
        "s_mov_b32 s40, %[n]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        // ---- l search: small = |H(i,i-1)| < eps * (|H(i-1,i-1)| + |H(i,i)|  (or norm if that sum is 0)), lanes 1..n
        "v_add_f64 v[102:103], |v[80:81]|, |v[84:85]|\n\t"
        "v_cmp_eq_f64 vcc, 0, v[102:103]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 v103, v103, %[normhi], vcc\n\t"
        "v_cndmask_b32 v102, v102, %[normlo], vcc\n\t"
        "v_ldexp_f64 v[102:103], v[102:103], %[m52]\n\t"
        "v_cmp_lt_f64 s[42:43], |v[82:83]|, v[102:103]\n\t"
        "s_nop 1\n\t"
        "s_and_b64 s[42:43], s[42:43], %[lvalid]\n\t"
        "s_flbit_i32_b64 s44, s[42:43]\n\t"
        "s_sub_u32 s44, 63, s44\n\t"
        "s_cmp_eq_u64 s[42:43], 0\n\t"
        "s_cselect_b32 s44, 0, s44\n\t"                                      // s44 = l
        "s_sub_u32 s45, s40, 1\n\t"
        "s_cmp_ge_u32 s44, s45\n\t"
        "s_cbranch_scc1 Lsw_defl_%=\n\t"
        // ---- shifts: x = H(n,n), y = H(n-1,n-1), w = H(n,n-1) * H(n-1,n)
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_f64 v[104:105], v[98:99], v[96:97]\n\t"                       // w
        // ---- m search, lane mm = i in l..n-2
        "v_add_f64 v[106:107], v[100:101], -v[84:85]\n\t"                    // rr = x - zz
        "v_add_f64 v[108:109], v[94:95], -v[84:85]\n\t"                      // ss = y - zz
        "v_mul_f64 v[110:111], v[106:107], v[108:109]\n\t"
        "v_add_f64 v[110:111], v[110:111], -v[104:105]\n\t"                  // rr*ss - w
        "v_div_scale_f64 v[112:113], s[68:69], v[88:89], v[88:89], v[110:111]\n\t"
        "v_rcp_f64 v[114:115], v[112:113]\n\t"
        "v_add_f64 v[122:123], v[90:91], -v[84:85]\n\t"                      // (wait state) qm = H(i+1,i+1) - zz ...
        "v_fma_f64 v[116:117], -v[112:113], v[114:115], 1.0\n\t"
        "v_fma_f64 v[114:115], v[114:115], v[116:117], v[114:115]\n\t"
        "v_fma_f64 v[116:117], -v[112:113], v[114:115], 1.0\n\t"
        "v_fma_f64 v[114:115], v[114:115], v[116:117], v[114:115]\n\t"
        "v_div_scale_f64 v[118:119], vcc, v[110:111], v[88:89], v[110:111]\n\t"
        "v_mul_f64 v[120:121], v[118:119], v[114:115]\n\t"
        "v_fma_f64 v[112:113], -v[112:113], v[120:121], v[118:119]\n\t"
        "v_add_f64 v[122:123], v[122:123], -v[106:107]\n\t"                  // (wait state) ... - rr
        "v_add_f64 v[122:123], v[122:123], -v[108:109]\n\t"                  // (wait state) ... - ss
        "v_div_fmas_f64 v[112:113], v[112:113], v[114:115], v[120:121]\n\t"
        "v_div_fixup_f64 v[110:111], v[112:113], v[88:89], v[110:111]\n\t"
        "v_add_f64 v[110:111], v[110:111], v[86:87]\n\t"                     // pm = (rr*ss - w) / H(i+1,i) + H(i,i+1)
        "v_add_f64 v[124:125], |v[110:111]|, |v[122:123]|\n\t"
        "v_add_f64 v[124:125], v[124:125], |v[92:93]|\n\t"                   // |pm| + |qm| + |rm|
        "v_frexp_exp_i32_f64 v74, v[124:125]\n\t"
        "v_sub_u32 v74, 0, v74\n\t"
        "v_ldexp_f64 v[110:111], v[110:111], v74\n\t"                        // pm, qm, rm scaled by the exact power of two
        "v_ldexp_f64 v[122:123], v[122:123], v74\n\t"
        "v_ldexp_f64 v[126:127], v[92:93], v74\n\t"
        "v_add_f64 v[124:125], |v[122:123]|, |v[126:127]|\n\t"
        "v_mul_f64 v[124:125], v[124:125], |v[82:83]|\n\t"                   // |H(i,i-1)| * (|qm| + |rm|)
        "v_add_f64 v[116:117], |v[80:81]|, |v[84:85]|\n\t"
        "v_add_f64 v[116:117], v[116:117], |v[90:91]|\n\t"                   // |H(i-1,i-1)| + |zz| + |H(i+1,i+1)|
        "v_mul_f64 v[116:117], |v[110:111]|, v[116:117]\n\t"
        "v_ldexp_f64 v[116:117], v[116:117], %[m52]\n\t"
        "v_cmp_lt_f64 s[46:47], v[124:125], v[116:117]\n\t"
        // lanes l..n-2 take part; lane l always passes
        "s_lshl_b64 s[48:49], 1, s44\n\t"                                    // bit l
        "s_sub_u32 s50, s40, 1\n\t"
        "s_lshl_b64 s[50:51], 1, s50\n\t"
        "s_sub_u32 s50, s50, 1\n\t"
        "s_subb_u32 s51, s51, 0\n\t"                                         // bits 0..n-2
        "s_sub_u32 s52, s48, 1\n\t"
        "s_subb_u32 s53, s49, 0\n\t"                                         // bits 0..l-1
        "s_andn2_b64 s[50:51], s[50:51], s[52:53]\n\t"                       // bits l..n-2
        "s_or_b64 s[46:47], s[46:47], s[48:49]\n\t"
        "s_and_b64 s[46:47], s[46:47], s[50:51]\n\t"
        "s_flbit_i32_b64 s45, s[46:47]\n\t"
        "s_sub_u32 s45, 63, s45\n\t"                                         // s45 = m
        "v_readlane_b32 s54, v110, s45\n\t"
        "v_readlane_b32 s55, v111, s45\n\t"
        "v_readlane_b32 s56, v122, s45\n\t"
        "v_readlane_b32 s57, v123, s45\n\t"
        "v_readlane_b32 s58, v126, s45\n\t"
        "v_readlane_b32 s59, v127, s45\n\t"
        // H(i, i-2) = 0 for i in m+2..n, H(i, i-3) = 0 for i in m+3..n
        "s_add_u32 s60, s40, 1\n\t"
        "s_lshl_b64 s[60:61], 1, s60\n\t"
        "s_sub_u32 s60, s60, 1\n\t"
        "s_subb_u32 s61, s61, 0\n\t"                                         // bits 0..n
        "s_add_u32 s62, s45, 2\n\t"
        "s_lshl_b64 s[62:63], -1, s62\n\t"                                   // bits m+2..63
        "s_and_b64 s[62:63], s[62:63], s[60:61]\n\t"
        "v_mov_b32 v76, 0\n\t"
        "v_mov_b32 v77, 0\n\t"
        "s_mov_b64 exec, s[62:63]\n\t"
        "ds_write_b64 v72, v[76:77] offset:208\n\t"                          // &H(i, i-2) = base + 224 - 16
        "s_add_u32 s64, s45, 2\n\t"
        "s_bitset0_b64 s[62:63], s64\n\t"
        "s_mov_b64 exec, s[62:63]\n\t"
        "ds_write_b64 v72, v[76:77] offset:200\n\t"                          // &H(i, i-3)
        "s_mov_b64 exec, s[66:67]\n\t"
        "s_mov_b32 %[st], 0\n\t"
        "s_branch Lsw_done_%=\n"
        "Lsw_defl_%=:\n\t"
        "s_mov_b32 %[st], 1\n\t"
        "s_mov_b32 s45, 0\n\t"
        "s_mov_b64 s[54:55], 0\n\t"
        "s_mov_b64 s[56:57], 0\n\t"
        "s_mov_b64 s[58:59], 0\n"
        "Lsw_done_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 %[lo], s44\n\t"
        "s_mov_b32 %[mo], s45\n\t"
        "s_mov_b64 %[po], s[54:55]\n\t"
        "s_mov_b64 %[qo], s[56:57]\n\t"
        "s_mov_b64 %[ro], s[58:59]"
        : [st] "=&s"(st), [lo] "=&s"(lo), [mo] "=&s"(mo), [po] "=&s"(po), [qo] "=&s"(qo), [ro] "=&s"(ro)
        : [base] "v"(base), [nbase] "s"(nbase), [n] "s"(n), [lvalid] "s"(lvalid), [normlo] "v"(__double2loint(norm)),
          [normhi] "v"(__double2hiint(norm)), [m52] "s"(-52)
        : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55",
          "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "v72", "v73", "v74", "v75", "v76",
          "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95",
          "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
          "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    status = st; l_out = lo; m_out = mo; p = po; q = qo; r = ro;
}

struct EigArgs {
    PnpProblem prob[kPnpMaxBatch];
    int32_t H, S;
    int32_t slot0, pad2_;
    double thresh;
    int32_t use_mle;
    double *Sg;             // [H][729]  in: the action matrices; scratch of the loop-form back-substitution afterwards
    const double *Tg;       // [H][27]
    const int32_t *sample;  // [H][kSampleMax]
    const int32_t *ok;      // [H]
    int32_t mask_words;     // ceil(N/64)
    double *T_out;          // [H][16] column-major b_T_a
    double *cost;           // [H]
    int32_t *nin;           // [H]
    int32_t *valid;         // [H]  1 = exactly one DLS solution
    int32_t *nsol;          // [H]  number of cheirality-valid real solutions (diagnostics)
    unsigned long long *mask;  // [H][mask_words]
    int32_t debug_stop;     // tuning only (CHIP_PNP_DEBUG_STOP)
    int32_t backsub_loop;   // CHIP_PNP_BACKSUB=loop: the loop form of the back-substitution only (test knob: the two forms must agree bit for bit)
    unsigned long long *stamps;   // tuning only (CHIP_PNP_STAMPS): [H][8] shader-clock totals per wave, see pnp_eig_score<true>
};

#define HH(i, j) Hs[(i) * EN + (j)]
// Of the accumulated transformation V only the rows of the monomials {1, s3, s2, s1} are ever read back (the back-transform of the
// eigenvectors), the QR iteration updates V row by row (a column operation never mixes rows), and ortran -- which does mix rows --
// works on one column per lane: so V lives in REGISTERS during ortran (vcol[i] of lane j = V(i, j)) and only those four rows exist
// in LDS afterwards (round 5: 13.25 -> 8.4 KiB of LDS per wave, 12 -> 16 waves per CU when many problems are batched).
constexpr int kVRows = 4;
__device__ constexpr int kVRow[kVRows] = {0, 1, 3, 9};
#define V4(r, j) Vs4[(r) * EN + (j)]

// pnp_eig_score is ONE wave per workgroup: the LDS unit executes a wave's DS instructions in issue order, so a write by any lane
// is visible to a later read by any lane of the same wave without waiting for the write to complete.  What the code needs
// between its phases is therefore only that the COMPILER keeps the program order of the LDS accesses -- a wavefront-scope
// fence + scheduling barrier -- not __syncthreads(), whose workgroup-scope fence drains the LDS queue (s_waitcnt lgkmcnt(0)).
// Measured: 20 fewer full LDS drains in the kernel, no change in the call time (1.17 ms) -- the double-shift step is bound by
// its dependent fp64 chain (exact sqrt + IEEE division of the reflector, ~45 dependent operations), not by the LDS queue.
#define WAVE_SYNC()                                                   \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
    } while (0)

// ---- Hessenberg reduction with compile-time step index (round 4) ------------------------------------------------------------------
// orthes / ortran step M works on rows / columns M..26: L = 27 - M terms per serial sum.  With M a template parameter every sum is
// straight-line code -- no loop counters, no remainder handling, the operands of a sum loaded once and kept in registers for the
// update that follows it (the looped form re-read them) -- in the reference's term order.  25 + 25 instantiations, executed once each.
template <int M>
__device__ __forceinline__ void orthes_step(double *Hs, double *us, double *ortm, int lane)
{
    constexpr int L = EN - M, high = EN - 1;
    double scale = 0.0;                                   // scale = sum_{i = M..high} |H(i, M-1)|, ascending (wave-uniform chain)
    {
        double c[L];
#pragma unroll
        for (int t = 0; t < L; t++) c[t] = HH(M + t, M - 1);
#pragma unroll
        for (int t = 0; t < L; t++) scale = scale + fabs(c[t]);
    }
    if (scale != 0.0) {
        const bool mine = lane >= M && lane <= high;
        const double colv = mine ? HH(lane, M - 1) : 0.0;
        const double ov = colv / scale;                   // u_i = H(i, M-1) / scale on lanes M..high
        if (lane < EN + 5) us[lane] = mine ? ov : 0.0;
        WAVE_SYNC();
        double h = 0.0;                                   // h = sum_{i = high..M} u_i^2, descending
        {
            double u[L];
#pragma unroll
            for (int t = 0; t < L; t++) u[t] = us[M + t];
#pragma unroll
            for (int t = L - 1; t >= 0; t--) h = h + u[t] * u[t];
        }
        double g = sqrt(h);
        const double om = us[M];
        if (om > 0) g = -g;
        h = h - om * g;
        WAVE_SYNC();
        if (lane == M) us[M] = om - g;
        WAVE_SYNC();
        if (lane >= M && lane < EN) {                     // H = (I - u u^T/h) H, column j = lane
            double b[L];
#pragma unroll
            for (int t = 0; t < L; t++) b[t] = HH(M + t, lane);
            double f = 0.0;
#pragma unroll
            for (int t = L - 1; t >= 0; t--) f = f + us[M + t] * b[t];
            f = f / h;
            const double c = -f;
#pragma unroll
            for (int t = 0; t < L; t++) HH(M + t, lane) = b[t] + c * us[M + t];
        }
        WAVE_SYNC();
        if (lane <= high) {                               // H = H (I - u u^T/h), row i = lane
            double b[L];
#pragma unroll
            for (int t = 0; t < L; t++) b[t] = HH(lane, M + t);
            double f = 0.0;
#pragma unroll
            for (int t = L - 1; t >= 0; t--) f = f + us[M + t] * b[t];
            f = f / h;
            const double c = -f;
#pragma unroll
            for (int t = 0; t < L; t++) HH(lane, M + t) = b[t] + c * us[M + t];
        }
        WAVE_SYNC();
        if (lane == 0) {
            ortm[M] = scale * (om - g);
            HH(M, M - 1) = scale * g;
        }
        WAVE_SYNC();
    } else {
        if (lane == 0) ortm[M] = 0.0;
        WAVE_SYNC();
    }
}
template <int M>
__device__ __forceinline__ void ortran_step(double *Hs, double (&vcol)[EN], double *us, const double *ortm, int lane)
{
    constexpr int L = EN - M, high = EN - 1;
    const double hmm = HH(M, M - 1);
    if (hmm != 0.0) {
        const double om = ortm[M];
        // u = (ort[M], H(M+1, M-1), .., H(high, M-1)) staged contiguously for the broadcast reads
        if (lane >= M && lane <= high) us[lane] = (lane == M) ? om : HH(lane, M - 1);
        WAVE_SYNC();
        // column j = lane of V, rows M..26, in the lane's own registers: the arithmetic runs on every lane (lanes outside M..26 hold
        // columns the reflector does not touch: their result is discarded by the select), no LDS traffic but the broadcast reads of u
        const bool mine = lane >= M && lane <= high;
        double g = 0.0;
#pragma unroll
        for (int t = 0; t < L; t++) g = g + us[M + t] * vcol[M + t];
        g = (g / om) / hmm;
#pragma unroll
        for (int t = 0; t < L; t++) {
            const double nv = vcol[M + t] + g * us[M + t];
            vcol[M + t] = mine ? nv : vcol[M + t];
        }
        WAVE_SYNC();
    }
}
template <int M>
struct HessenbergSteps {
    static __device__ __forceinline__ void orthes(double *Hs, double *us, double *ortm, int lane)
    {
        orthes_step<M>(Hs, us, ortm, lane);
        if constexpr (M < EN - 2) HessenbergSteps<M + 1>::orthes(Hs, us, ortm, lane);
    }
    static __device__ __forceinline__ void ortran(double *Hs, double (&vcol)[EN], double *us, const double *ortm, int lane)
    {
        ortran_step<M>(Hs, vcol, us, ortm, lane);
        if constexpr (M > 1) HessenbergSteps<M - 1>::ortran(Hs, vcol, us, ortm, lane);
    }
};

// ---- back-substitution with compile-time row index (round 4) --------------------------------------------------------------------
// hqr2's real-eigenvector back-substitution, lane n = eigenvector n, as straight-line code over the rows i = 25 .. 0 with the vector x in
// REGISTERS (x[j], j a compile-time index).  The lane-per-eigenvector loops it replaces ran every lane's own trip counts one after the
// other (19 us of the reference-mode call; ~5 k instructions per wave, which is what counts when many problems are batched).  What makes
// the uniform form possible:
//   * which rows belong to a complex pair depends on wi[] only, so the branch structure is wave-uniform (bit masks of wi < 0 / wi == 0);
//   * the lower bound of a row's sum is uniform too: j starts at i + 1 unless row i + 1 is the pending second row of a pair (then i + 2);
//   * the upper bound (j <= n) is per lane -- but x[j] of lane n is 0.0 for j > n (never written), and rr + H(i,j) * 0.0 == rr bit for bit
//     for finite H (rr starts as +0.0 and a sum that starts there never becomes -0.0), so the terms beyond n are simply added.
// Same operations in the same order on the same values otherwise.  Two cases are left to the loop form (the caller falls back to it and
// redoes the whole back-substitution): a non-finite entry anywhere in H or V (0 * inf), and the overflow rescaling of hqr2 (x /= |x_i|
// when eps * x_i^2 > 1), neither of which a test scene has produced.
template <int I>
struct BackSubRow {
    // returns false when some lane needs the rescaling
    static __device__ __forceinline__ bool run(const double *Hs, const double *wr, const double *wi, double (&x)[EN], double pe, double &zz, double &ss,
                                               bool real_root, int lane, unsigned neg_mask, unsigned zero_mask, double eps, double eps_norm)
    {
        const double ww = HH(I, I) - pe;
        double rr = 0.0;
        if (!((neg_mask >> (I + 1)) & 1u)) rr = rr + HH(I, I + 1) * x[I + 1];   // row I + 1 solved in its own step
        // terms j = I + 2 .. 26 in the reference's order, the broadcast reads of H(I, j) software-pipelined in groups of kG: group g + 1
        // is in flight underneath the arithmetic of group g.  (Left to itself hipcc hoists ALL reads of a row in front of its chain --
        // up to 50 registers next to the 54 of x -- which no longer fits once the kernel is held to 128 VGPRs = 4 waves per SIMD.)
        {
            constexpr int kG = 6, J0 = I + 2, NT = EN - J0, NG = (NT + kG - 1) / kG;
            double hb[2][kG];
            if constexpr (NT > 0) {
#pragma unroll
                for (int t = 0; t < kG; t++) hb[0][t] = (t < NT) ? HH(I, J0 + t) : 0.0;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (g + 1 < NG) {
#pragma unroll
                        for (int t = 0; t < kG; t++) hb[(g + 1) & 1][t] = ((g + 1) * kG + t < NT) ? HH(I, J0 + (g + 1) * kG + t) : 0.0;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < kG; t++)
                        if (g * kG + t < NT) rr = rr + hb[g & 1][t] * x[J0 + g * kG + t];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        const bool act = real_root && lane > I;
        bool big = false;
        if ((neg_mask >> I) & 1u) { zz = ww; ss = rr; }
        else if ((zero_mask >> I) & 1u) {
            const double xi = (ww != 0.0) ? -rr / ww : -rr / eps_norm;
            x[I] = act ? xi : x[I];
            const double tt = fabs(xi);
            big = act && (eps * tt) * tt > 1;
        } else {  // 2x2 block of a complex pair above a real eigenvalue
            const double xx = HH(I, I + 1), yy = HH(I + 1, I);
            const double wri = wr[I], wii = wi[I];
            const double qq = (wri - pe) * (wri - pe) + wii * wii;
            const double tt = (xx * ss - zz * rr) / qq;
            const double x1 = (fabs(xx) > fabs(zz)) ? (-rr - ww * tt) / xx : (-ss - yy * tt) / zz;
            x[I] = act ? tt : x[I];
            x[I + 1] = act ? x1 : x[I + 1];
            const double at = fabs(tt);
            big = act && (eps * at) * at > 1;
        }
        if (__builtin_amdgcn_ballot_w64(big) != 0ull) return false;
        if constexpr (I > 0) return BackSubRow<I - 1>::run(Hs, wr, wi, x, pe, zz, ss, real_root, lane, neg_mask, zero_mask, eps, eps_norm);
        else return true;
    }
};

// STAMP = true (CHIP_PNP_STAMPS=1, tuning only): the wave accumulates s_memtime differences per segment of the QR iteration and
// leaves them in a.stamps[hyp][0..7]: 0 sweep overhead (deflation test, shifts, m search), 1 reflector (|p|+|q|+|r| .. quotients
// read back), 2 row modification, 3 column modification + forwarding, 4 number of double-shift steps, 5 number of sweeps,
// 6 Hessenberg reduction + accumulation, 7 whole kernel.  The stamps themselves cost ~10 % (s_memtime + lgkmcnt wait each).
template <bool STAMP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void pnp_eig_score(EigArgs a)
{
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0, t_start = 0;
    if constexpr (STAMP) { t_start = t_prev = __builtin_readcyclecounter(); }
#define PNP_STAMP(i)                                                        \
    do {                                                                    \
        if constexpr (STAMP) {                                              \
            const unsigned long long t_ = __builtin_readcyclecounter();     \
            clk[i] += t_ - t_prev;                                          \
            t_prev = t_;                                                    \
        }                                                                   \
    } while (0)
    // H sits 28 doubles (one row + one element) into its allocation: qr_sweep_head_asm addresses every operand of lane i relative to
    // &H(i-1, i-1), which for lane 0 lies 224 bytes in front of H -- inside this pad, not below LDS address 0 (no address wrap-around;
    // the pad's content is never used: lane 0's H(-1,-1) / H(0,-1) only feed a comparison whose result is overridden)
    __shared__ double HsPad[28 + EN * EN], Vs4[kVRows * EN];
    static_assert(28 * 8 >= 0xd8 + 8, "HsPad: qr_steps_asm's row address starts one row (0xd8 bytes) below HH(0, lane)");
    double *const Hs = HsPad + 28;
    __shared__ double ortm[EN], wr[EN], wi[EN], Tf[27], sxs[kSampleMax * 3], model[16];
    __shared__ __attribute__((aligned(16))) double us[EN + 5];   // the current Householder vector (orthes / ortran)
    const int lane = threadIdx.x;
    const int hyp = blockIdx.x + a.slot0;                          // slot: hypothesis (hyp % H) of problem (hyp / H)
    const PnpProblem pr = a.prob[hyp / a.H];
    const int low = 0, high = EN - 1, nn = EN;
    const double eps = DBL_EPSILON;

    if (a.ok[hyp] == 0) {
        if (lane == 0) { a.valid[hyp] = 0; a.nsol[hyp] = -1; a.nin[hyp] = 0; a.cost[hyp] = INFINITY; }
        return;
    }
    for (int e = lane; e < EN * EN; e += 64) Hs[e] = a.Sg[(size_t)hyp * 729 + e];
    if (lane < 27) Tf[lane] = a.Tg[hyp * 27 + lane];
    if (lane < a.S) {
        const int s = a.sample[hyp * kSampleMax + lane];
        sxs[3 * lane] = pr.X[3 * s]; sxs[3 * lane + 1] = pr.X[3 * s + 1]; sxs[3 * lane + 2] = pr.X[3 * s + 2];
    }
    WAVE_SYNC();

    if (a.debug_stop == 1) return;
    // ================= Householder reduction to Hessenberg form (orthes) =================
    // The reflector u (EISPACK's ort[]) lives in LDS (us[], zero outside m..high): the serial sums read it as wave-uniform
    // broadcasts, four terms ahead of the arithmetic (dot_desc / dot_asc / axpy_rows above).  Term order is the reference's.
#ifndef CHIP_PNP_HESS_LOOPED
    HessenbergSteps<1>::orthes(Hs, us, ortm, lane);
#else
    for (int m = low + 1; m <= high - 1; m++) {
        // scale = sum_{i = m..high} |H(i, m-1)|, ascending: every lane runs the same chain on broadcast reads of the column
        const double scale = abs_sum_asc<EN>(&HH(0, m - 1), m, high, 0.0);
        if (scale != 0.0) {
            const bool mine = lane >= m && lane <= high;
            const double colv = mine ? HH(lane, m - 1) : 0.0;
            double ov = colv / scale;                       // u_i = H(i, m-1) / scale on lanes m..high
            if (lane < EN + 5) us[lane] = mine ? ov : 0.0;
            WAVE_SYNC();
            double h = 0.0;                                  // h = sum_{i = high..m} u_i^2, descending
            h = sumsq_desc(us, high, m);
            double g = sqrt(h);
            const double om = us[m];
            if (om > 0) g = -g;
            h = h - om * g;
            WAVE_SYNC();
            if (lane == m) us[m] = om - g;
            WAVE_SYNC();
            if (lane >= m && lane < nn) {  // H = (I - u u^T/h) H, column j = lane
                double *colj = &HH(0, lane);
                double f = dot_desc<EN>(us, colj, high, m);
                f = f / h;
                axpy_rows<EN>(us, colj, -f, m, high);
            }
            WAVE_SYNC();
            if (lane <= high) {  // H = H (I - u u^T/h), row i = lane
                double *rowi = &HH(lane, 0);
                double f = dot_desc<1>(us, rowi, high, m);
                f = f / h;
                axpy_rows<1>(us, rowi, -f, m, high);
            }
            WAVE_SYNC();
            if (lane == 0) {
                ortm[m] = scale * (om - g);
                HH(m, m - 1) = scale * g;
            }
            WAVE_SYNC();
        } else {
            if (lane == 0) ortm[m] = 0.0;
            WAVE_SYNC();
        }
    }
#endif
    {   // accumulate the reflectors into V (ortran): column `lane` of V in registers, then the four rows that are read back -> LDS
        double vcol[EN];
#pragma unroll
        for (int i = 0; i < EN; i++) vcol[i] = (i == lane) ? 1.0 : 0.0;
        HessenbergSteps<EN - 2>::ortran(Hs, vcol, us, ortm, lane);
        if (lane < EN) {
#pragma unroll
            for (int rr = 0; rr < kVRows; rr++) V4(rr, lane) = vcol[kVRow[rr]];
        }
    }
    for (int e = lane; e < EN * EN; e += 64) {
        const int i = e / EN, j = e % EN;
        if (j < i - 1) Hs[e] = 0.0;
    }
    WAVE_SYNC();

    if (a.debug_stop == 2) return;
    PNP_STAMP(6);
    // ================= Francis double-shift QR with accumulation (hqr2) =================
    double norm = 0.0;   // sum over the Hessenberg part, row by row (one serial chain; broadcast reads four terms ahead)
    for (int i = 0; i < nn; i++) norm = abs_sum_asc<1>(&HH(i, 0), (i - 1 > 0 ? i - 1 : 0), nn - 1, norm);
    int n = nn - 1;
    double exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, w, x, y;
    int iter = 0;
    bool failed = false;
    while (n >= low) {
        // l = largest index in (low, n] whose sub-diagonal element is negligible (else low).  The reference scans
        // l = n, n-1, ... sequentially; every index is independent, so lane l tests its own and a ballot picks the
        // same l (exact reformulation: same arithmetic per index, same selection).
        int l = 0, m = 0;
        bool head = false;       // the sweep's head (l search, shifts, m search, zeroing) was done by qr_sweep_head_asm
        bool have_l = false;     // ... or at least its deflation test (l is valid: one or two roots to split off)
        if constexpr (!STAMP) {
            if (n >= 2 && iter != 10 && iter != 30 && iter < 60) {   // (exceptional shifts and the iteration limit stay compiled)
                int st = 1;
                qr_sweep_head_asm(n, lds_addr32(Hs), lane, norm, st, l, m, p, q, r);
                head = st == 0;
                have_l = true;
            }
        }
        if (!have_l) {
            bool small = false;
            if (lane > low && lane <= n) {
                double ss = fabs(HH(lane - 1, lane - 1)) + fabs(HH(lane, lane));
                if (ss == 0.0) ss = norm;
                small = fabs(HH(lane, lane - 1)) < eps * ss;
            }
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(small);
            l = bm ? 63 - __builtin_clzll(bm) : low;
        }
        if (!head && l == n) {  // one root
            const double v = HH(n, n) + exshift;
            WAVE_SYNC();
            if (lane == 0) { HH(n, n) = v; wr[n] = v; wi[n] = 0.0; }
            WAVE_SYNC();
            n--; iter = 0;
        } else if (!head && l == n - 1) {  // two roots
            w = HH(n, n - 1) * HH(n - 1, n);
            p = (HH(n - 1, n - 1) - HH(n, n)) / 2.0;
            q = p * p + w;
            z = sqrt(fabs(q));
            const double hnn = HH(n, n) + exshift, hn1 = HH(n - 1, n - 1) + exshift;
            x = hnn;
            WAVE_SYNC();
            if (lane == 0) { HH(n, n) = hnn; HH(n - 1, n - 1) = hn1; }
            WAVE_SYNC();
            if (q >= 0) {  // real pair
                z = (p >= 0) ? p + z : p - z;
                const double w0 = x + z;
                double w1 = w0;
                if (z != 0.0) w1 = x - w / z;
                if (lane == 0) { wr[n - 1] = w0; wr[n] = w1; wi[n - 1] = 0.0; wi[n] = 0.0; }
                x = HH(n, n - 1);
                s = fabs(x) + fabs(z);
                p = x / s; q = z / s;
                r = sqrt(p * p + q * q);
                p = p / r; q = q / r;
                WAVE_SYNC();
                if (lane >= n - 1 && lane < nn) {  // rows n-1, n
                    const int j = lane;
                    const double zz = HH(n - 1, j);
                    HH(n - 1, j) = q * zz + p * HH(n, j);
                    HH(n, j) = q * HH(n, j) - p * zz;
                }
                if (lane >= 32 && lane - 32 < kVRows) {  // accumulate (independent of H): the four rows of V that are kept
                    const int i = lane - 32;
                    const double zz = V4(i, n - 1);
                    V4(i, n - 1) = q * zz + p * V4(i, n);
                    V4(i, n) = q * V4(i, n) - p * zz;
                }
                WAVE_SYNC();
                if (lane <= n) {  // columns n-1, n
                    const int i = lane;
                    const double zz = HH(i, n - 1);
                    HH(i, n - 1) = q * zz + p * HH(i, n);
                    HH(i, n) = q * HH(i, n) - p * zz;
                }
                WAVE_SYNC();
            } else {  // complex pair
                if (lane == 0) { wr[n - 1] = x + p; wr[n] = x + p; wi[n - 1] = z; wi[n] = -z; }
                WAVE_SYNC();
            }
            n -= 2; iter = 0;
        } else {
          if (head) {
            iter++;
          } else {
            x = HH(n, n); y = 0.0; w = 0.0;
            if (l < n) { y = HH(n - 1, n - 1); w = HH(n, n - 1) * HH(n - 1, n); }
            if (iter == 10) {  // Wilkinson's exceptional shift
                exshift = exshift + x;
                WAVE_SYNC();
                if (lane >= low && lane <= n) HH(lane, lane) = HH(lane, lane) - x;
                WAVE_SYNC();
                s = fabs(HH(n, n - 1)) + fabs(HH(n - 1, n - 2));
                x = y = 0.75 * s;
                w = -0.4375 * s * s;
            }
            if (iter == 30) {  // second exceptional shift
                s = (y - x) / 2.0;
                s = s * s + w;
                if (s > 0) {
                    s = sqrt(s);
                    if (y < x) s = -s;
                    s = x - w / ((y - x) / 2.0 + s);
                    WAVE_SYNC();
                    if (lane >= low && lane <= n) HH(lane, lane) = HH(lane, lane) - s;
                    WAVE_SYNC();
                    exshift = exshift + s;
                    x = y = w = 0.964;
                }
            }
            iter++;
            if (iter > 60) { failed = true; break; }
            // two consecutive small sub-diagonal elements: the reference scans m = n-2, n-3, ..., l and stops at the first
            // m that passes the test (or at m == l).  Lane m evaluates its own candidate (p,q,r normalised as in the
            // reference); the ballot selects the largest passing m; p,q,r come from that lane.  Exact reformulation.
            {
                double pm = 0.0, qm = 0.0, rm = 0.0;
                bool pass = false;
                if (lane >= l && lane <= n - 2) {
                    const int mm = lane;
                    const double zz = HH(mm, mm);
                    double rr = x - zz, ss = y - zz;
                    pm = (rr * ss - w) / HH(mm + 1, mm) + HH(mm, mm + 1);
                    qm = HH(mm + 1, mm + 1) - zz - rr - ss;
                    rm = HH(mm + 2, mm + 1);
                    ss = fabs(pm) + fabs(qm) + fabs(rm);
                    {   // exact power-of-two scaling instead of EISPACK's division by |p|+|q|+|r| (see oracle/pnp_ransac.c)
                        int e2;
                        (void)frexp(ss, &e2);
                        pm = ldexp(pm, -e2); qm = ldexp(qm, -e2); rm = ldexp(rm, -e2);
                    }
                    if (mm == l) pass = true;
                    else pass = fabs(HH(mm, mm - 1)) * (fabs(qm) + fabs(rm)) <
                                eps * (fabs(pm) * (fabs(HH(mm - 1, mm - 1)) + fabs(zz) + fabs(HH(mm + 1, mm + 1))));
                }
                const unsigned long long bm = __builtin_amdgcn_ballot_w64(pass);
                m = 63 - __builtin_clzll(bm);   // lane l always passes and l <= n-2 in this branch
                p = lane_value_f64(pm, m); q = lane_value_f64(qm, m); r = lane_value_f64(rm, m);   // m is wave-uniform: v_readlane, not ds_bpermute
            }
            WAVE_SYNC();
            if (lane >= m + 2 && lane <= n) { HH(lane, lane - 2) = 0.0; if (lane > m + 2) HH(lane, lane - 3) = 0.0; }
            WAVE_SYNC();
          }
            // Double QR step on rows l..n, columns m..n.  Per step: (1) the reflector from column k-1, (2) row modification
            // (lane = column j of H), (3) column modification (lane = row i of H) together with the accumulation into V
            // (lanes 32..58 = row i of V): both have the form  pp = x*A(i,k) + y*A(i,k+1) [+ z*A(i,k+2)], so one
            // instruction stream serves them with a per-lane base pointer instead of two divergent blocks executed back
            // to back.  The first column of (3) is also what the next step's reflector is built from: it is forwarded
            // through v_readlane instead of an LDS write -> barrier -> read.
            PNP_STAMP(0);
            if constexpr (STAMP) clk[5] += 1;
            double *const Abase = (lane < 32) ? Hs : Vs4;    // lanes 0..26 work on H, lanes 32..35 on the four kept rows of V
            // (idle lanes -- 27..31, 36..63 -- read a row that exists, EN - 1 resp. 0, and store nothing: with V down to four rows the
            //  "row 27..31 of their matrix" they used to touch would lie past the workgroup's LDS allocation)
            const int arow = (lane < 32) ? (lane < EN ? lane : EN - 1) : (lane - 32 < kVRows ? lane - 32 : 0);
            bool fwd = false;
            double fp = 0.0, fq = 0.0, fr = 0.0;
            // One double-shift step.  NOTLAST (k != n-1: the reflector spans three rows) is a compile-time flag: with a run-time one
            // the third LDS read of each modification sat behind a branch, i.e. behind the first two reads' latency (shader-clock
            // split, profiles/r03_pnp_pmc.md: row / column modification ~480 / ~520 of ~1500 stamped cycles per step, the arithmetic in
            // them ~50).  The row modification's inputs H(k..k+2, lane) are final once step k-1 has written (the LDS executes a
            // wave's accesses in order), so they are read at the TOP of the step and land underneath the reflector's sqrt -> division
            // chain.  Same operations on the same values: bits unchanged.
            // Per-lane activity limit of the column modification, once per sweep: H lanes (< 32) work on rows arow <= min(n, k + 3), i.e.
            // "arow <= n and arow - 3 <= k"; V lanes 32..58 always; lanes 27..31 / 59..63 never.  One v_cmp per step instead of
            // s_min / v_mov / v_cndmask / v_cmp (round 4: the step is bound by its instruction COUNT, ~8.5 cycles each for a lone wave).
            const int col_lim = (lane < 32) ? ((lane <= n && lane < EN) ? lane - 3 : 0x7fffffff) : ((lane - 32 < kVRows) ? -0x7fffffff : 0x7fffffff);
            auto qr_step = [&](int k, auto notlast_tag) {
                constexpr bool NOTLAST = decltype(notlast_tag)::value;
                double pk = p, qk = q, rk = r;   // (k == m: from the m search)
                if (k != m) {   // (the rare LDS reads of p, q, r come BEFORE the prefetch: LDS returns in order, so the wait for them
                                //  then leaves the younger prefetch in flight; the other way round it drained the prefetch as well)
                    if (fwd) { pk = fp; qk = fq; rk = fr; }
                    else { pk = HH(k, k - 1); qk = HH(k + 1, k - 1); rk = NOTLAST ? HH(k + 2, k - 1) : 0.0; }
                }
                __builtin_amdgcn_sched_barrier(0);
                // row-modification inputs H(k..k+2, lane): every lane reads its own column (lanes outside k..26 read elements they
                // do not use -- inside the LDS allocation or past it, where reads return zero -- instead of a selected valid one)
                const bool rowact = lane >= k && lane < nn;
                const double h0 = Hs[k * EN + lane], h1 = Hs[(k + 1) * EN + lane];
                double h2 = 0.0;
                if constexpr (NOTLAST) h2 = Hs[(k + 2) * EN + lane];
                __builtin_amdgcn_sched_barrier(0);
                int ex = 0;
                if (k != m) {
                    fwd = false;
                    const double ax = fabs(pk) + fabs(qk) + fabs(rk);
                    if (ax == 0.0) return;
                    // overflow protection by an exact power-of-two scale 2^-ex, |p|+|q|+|r| = f * 2^ex (three v_ldexp instead of
                    // the division chain EISPACK has here; the oracle defines it the same way)
                    (void)frexp(ax, &ex);
                    pk = ldexp(pk, -ex); qk = ldexp(qk, -ex); rk = ldexp(rk, -ex);
                }
                fwd = false;
                // p, q, r are scaled so that |p| + |q| + |r| is in [0.5, 1) on both ways in (frexp above / in the m search), hence
                // p^2 + q^2 + r^2 is in [1/12, 1): sqrt_normal_range is IEEE sqrt there, without the range handling on the chain
                double sk = sqrt_normal_range(pk * pk + qk * qk + rk * rk);
                if (pk < 0) sk = -sk;
                if (k == m && sk == 0.0) return;   // k != m: |p|+|q|+|r| is in [0.5, 1) after the scaling above, s cannot be 0
                const double hkk1 = (k != m) ? ldexp(-sk, ex) : ((l != m) ? -HH(k, k - 1) : 0.0);
                const bool wr_sub = (k != m) || (l != m);
                pk = pk + sk;
                double xk, yk, zk, qq, rq;
                {   // x = p/s, y = q/s, z = r/s, q = q/p, r = r/p as one vector division over lanes 0..4
                    const double num = (lane == 0) ? pk : ((lane & 1) ? qk : rk);
                    const double den = (lane < 3) ? sk : pk;
                    const double quo = num / den;
                    xk = lane_value_f64(quo, 0); yk = lane_value_f64(quo, 1); zk = lane_value_f64(quo, 2);
                    qq = lane_value_f64(quo, 3); rq = lane_value_f64(quo, 4);
                }
                PNP_STAMP(1);
                WAVE_SYNC();
                if (wr_sub && lane == 63) HH(k, k - 1) = hkk1;
                {   // row modification, column j = lane: the arithmetic runs on every lane, only the stores are predicated
                    double pp = h0 + qq * h1;
                    if constexpr (NOTLAST) pp = pp + rq * h2;
                    const double n0 = h0 - pp * xk, n1 = h1 - pp * yk;
                    double n2 = 0.0;
                    if constexpr (NOTLAST) n2 = h2 - pp * zk;
                    if (rowact) {
                        if constexpr (NOTLAST) Hs[(k + 2) * EN + lane] = n2;
                        Hs[k * EN + lane] = n0;
                        Hs[(k + 1) * EN + lane] = n1;
                    }
                }
                WAVE_SYNC();
                PNP_STAMP(2);
                double c0;
                {   // column modification (H rows 0..min(n, k+3) on lanes 0..26 | V rows 0..26 on lanes 32..58): reads and arithmetic on
                    // every lane (idle lanes touch rows 27..31 of their matrix: inside / past the LDS allocation, unused), stores predicated
                    double *row = Abase + arow * EN + k;
                    const double a0 = row[0], a1 = row[1];
                    double a2 = 0.0;
                    if constexpr (NOTLAST) a2 = row[2];
                    double pp = xk * a0 + yk * a1;
                    if constexpr (NOTLAST) pp = pp + zk * a2;
                    c0 = a0 - pp;
                    const double c1 = a1 - pp * qq;
                    if (col_lim <= k) {
                        if constexpr (NOTLAST) row[2] = a2 - pp * rq;
                        row[0] = c0;
                        row[1] = c1;
                    }
                }
                // next reflector: H(k+1,k), H(k+2,k), H(k+3,k) as just computed by lanes k+1, k+2, k+3 (<= min(n, k+3): active lanes)
                if constexpr (NOTLAST) {
                    fp = lane_value_f64(c0, k + 1);
                    fq = lane_value_f64(c0, k + 2);
                    fr = (k + 1 != n - 1) ? lane_value_f64(c0, k + 3) : 0.0;
                    fwd = true;
                }
                WAVE_SYNC();
                PNP_STAMP(3);
                if constexpr (STAMP) clk[4] += 1;
            };
            int k = m;
            if constexpr (!STAMP) {
                // The three-row steps k = m .. n-2 run in ONE hand-scheduled asm loop (qr_steps_asm).  The first step of a sweep takes
                // the same path as a forwarded one: its (p, q, r) come out of the m search scaled to |p|+|q|+|r| in [0.5, 1), so the
                // loop's own power-of-two scaling is the identity (exponent 0) and the arithmetic is the reference's; only its
                // sub-diagonal differs -- H(m, m-1) is negated in place iff l != m -- which is done here, after the loop has run it
                // with the sub-diagonal store masked off.  The last (two-row) step and the rare step that finds |p|+|q|+|r| == 0
                // stay with qr_step.
                const double hm = (l != m) ? HH(m, m - 1) : 0.0;
                while (k <= n - 2) {
                    const bool first = k == m;
                    if (!first && !fwd) { qr_step(k, std::true_type()); k++; continue; }
                    double ap = uniform_f64(first ? p : fp), aq = uniform_f64(first ? q : fq), ar = uniform_f64(first ? r : fr);
                    const uint32_t hs0 = lds_addr32(Hs), vs0 = lds_addr32(Vs4);
                    const uint32_t rowaddr = hs0 + (uint32_t)(k * EN + lane) * 8u;
                    const uint32_t coladdr = ((lane < 32) ? hs0 : vs0) + (uint32_t)(arow * EN + k) * 8u;
                    const uint32_t subaddr = hs0 + (uint32_t)(k * EN + k - 1) * 8u;
                    const uint64_t vmask = ((1ull << kVRows) - 1) << 32;                                  // the kept V rows on lanes 32..35
                    const uint64_t nmask = ((1ull << (n + 1)) - 1) | vmask;                               // H rows 0..n
                    const uint64_t rowmask = ((1ull << EN) - 1) & (~0ull << k);                          // lanes k..26
                    const uint64_t colmask = (((1ull << (k + 4)) - 1) | vmask) & nmask;                   // H rows 0..min(n, k+3) | V rows
                    int zero = 0;
                    qr_steps_asm(k, n - 2, ap, aq, ar, rowaddr, coladdr, subaddr, rowmask, colmask, nmask, first ? 0ull : (1ull << 63), 1, zero);
                    if (first && k > m && l != m && lane == 63) HH(m, m - 1) = -hm;
                    if (k > m) { fp = ap; fq = aq; fr = ar; fwd = true; }
                    if (zero == 1) { qr_step(k, std::true_type()); k++; }   // that step sees the zero itself and returns with fwd = false
                }
            } else {
                qr_step(m, std::true_type());
                k = m + 1;
            }
            for (; k <= n - 2; k++) qr_step(k, std::true_type());
            if (k == n - 1) qr_step(n - 1, std::false_type());   // (k == n: the asm loop has run the last step as well)
        }
        PNP_STAMP(0);
    }
    WAVE_SYNC();
    if constexpr (STAMP) {
        clk[7] = __builtin_readcyclecounter() - t_start;
        if (lane < 8) {
            unsigned long long v = clk[0];
#pragma unroll
            for (int i = 1; i < 8; i++) v = (lane == i) ? clk[i] : v;
            a.stamps[(size_t)hyp * 8 + lane] = v;
        }
    }

    if (a.debug_stop == 3) return;
    // ================= back-substitution (real eigenvalues only), one lane per eigenvector =================
    int my_valid = 0;
    double R[9], t3[3];
    const bool real_root = !failed && norm != 0.0 && lane < nn && wi[lane] == 0.0;
    double v4[4] = {0.0, 0.0, 0.0, 0.0};
    bool have_v4 = false;
    if (!a.backsub_loop) {
        // every entry of H and V finite?  (lane-strided pass; one ballot)
        unsigned worst = 0;
        for (int e = lane; e < EN * EN; e += 64) {
            const unsigned hh = (unsigned)((unsigned long long)__double_as_longlong(Hs[e]) >> 32) & 0x7fffffffu;
            worst = hh > worst ? hh : worst;
        }
        for (int e = lane; e < kVRows * EN; e += 64) {
            const unsigned hv = (unsigned)((unsigned long long)__double_as_longlong(Vs4[e]) >> 32) & 0x7fffffffu;
            worst = hv > worst ? hv : worst;
        }
        const bool finite = __builtin_amdgcn_ballot_w64(worst >= 0x7ff00000u) == 0ull;
        if (finite && __builtin_amdgcn_ballot_w64(real_root) != 0ull) {
            const double wl = lane < nn ? wi[lane] : 0.0;
            const unsigned neg_mask = (unsigned)__builtin_amdgcn_ballot_w64(lane < nn && wl < 0.0);
            const unsigned zero_mask = (unsigned)__builtin_amdgcn_ballot_w64(lane < nn && wl == 0.0);
            const double pe = lane < nn ? wr[lane] : 0.0;
            double x[EN];
            // (an opaque copy of the lane id: the unit vectors below are the same values ortran's vcol[] started from, and hipcc would
            //  otherwise keep those 54 registers alive across the whole QR iteration to reuse them here -- 169 instead of 129 VGPRs)
            int lane_x = lane;
            asm volatile("" : "+v"(lane_x));
#pragma unroll
            for (int j = 0; j < EN; j++) x[j] = (j == lane_x) ? 1.0 : 0.0;
            double zz = 0.0, ss = 0.0;
            if (BackSubRow<EN - 2>::run(Hs, wr, wi, x, pe, zz, ss, real_root, lane, neg_mask, zero_mask, eps, eps * norm)) {
                // back-transform only the rows we need: v = V * x, rows {0,1,3,9} = monomials {1, s3, s2, s1}  (x[k] = 0.0 for k > n)
#pragma unroll
                for (int rrw = 0; rrw < kVRows; rrw++) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < EN; k++) acc = acc + V4(rrw, k) * x[k];
                    v4[rrw] = acc;
                }
                have_v4 = true;
            }
        } else if (finite) have_v4 = true;   // no real root at all: nothing to solve
    }
    if (!have_v4 && real_root) {   // the loop form: one lane per eigenvector, each with its own trip counts (see BackSubRow)
        const int nn_ = lane;  // eigenvalue index n
        const double pe = wr[nn_];
        int l = nn_;
        double zz = 0.0, ss = 0.0;
        // The eigenvector x of the quasi-triangular form (entries 0..nn_) of this rare path lives in GLOBAL memory: the hypothesis's
        // slot of the action-matrix buffer (729 doubles, dead since the kernel copied it into LDS) -- the LDS holds no spare matrix
        // any more (round 5).  Lane nn_ owns entries [27 nn_, 27 nn_ + nn_].
        double *const xcol = a.Sg + (size_t)hyp * 729 + (size_t)nn_ * EN;
#define XX(i, j) xcol[(i)]
        XX(nn_, nn_) = 1.0;
        for (int i = nn_ - 1; i >= 0; i--) {
            const double ww = HH(i, i) - pe;
            double rr = 0.0;
            for (int j = l; j <= nn_; j++) rr = rr + HH(i, j) * XX(j, nn_);
            if (wi[i] < 0.0) { zz = ww; ss = rr; }
            else {
                l = i;
                if (wi[i] == 0.0) {
                    if (ww != 0.0) XX(i, nn_) = -rr / ww; else XX(i, nn_) = -rr / (eps * norm);
                } else {  // 2x2 block of a complex pair above a real eigenvalue
                    const double xx = HH(i, i + 1), yy = HH(i + 1, i);
                    const double qq = (wr[i] - pe) * (wr[i] - pe) + wi[i] * wi[i];
                    const double tt = (xx * ss - zz * rr) / qq;
                    XX(i, nn_) = tt;
                    if (fabs(xx) > fabs(zz)) XX(i + 1, nn_) = (-rr - ww * tt) / xx; else XX(i + 1, nn_) = (-ss - yy * tt) / zz;
                }
                const double tt = fabs(XX(i, nn_));
                if ((eps * tt) * tt > 1) for (int j = i; j <= nn_; j++) XX(j, nn_) = XX(j, nn_) / tt;
            }
        }
        // back-transform only the rows we need: v = V * x, rows {0,1,3,9} = monomials {1, s3, s2, s1}
        for (int rrw = 0; rrw < kVRows; rrw++) {
            double acc = 0.0;
            for (int k = 0; k <= nn_; k++) acc = acc + V4(rrw, k) * XX(k, nn_);
            v4[rrw] = acc;
        }
#undef XX
    }
    if (real_root) {
        const double s1 = v4[3] / v4[0], s2 = v4[2] / v4[0], s3 = v4[1] / v4[0];
        if (fabs(s1) <= DBL_MAX && fabs(s2) <= DBL_MAX && fabs(s3) <= DBL_MAX) {
            const double nq = sqrt(((1.0 + s1 * s1) + s2 * s2) + s3 * s3);
            const double qw = 1.0 / nq, qx = s1 / nq, qy = s2 / nq, qz = s3 / nq;
            const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
            const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
            const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
            const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
            R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
            R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
            R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
            for (int aa = 0; aa < 3; aa++) {
                double acc = 0.0;
                for (int j = 0; j < 9; j++) acc = acc + Tf[9 * aa + j] * R[j];
                t3[aa] = acc;
            }
            my_valid = 1;
            for (int i = 0; i < a.S; i++) {  // cheirality over the SAMPLE points
                const double zc = ((R[6] * sxs[3 * i] + R[7] * sxs[3 * i + 1]) + R[8] * sxs[3 * i + 2]) + t3[2];
                if (zc < 0) { my_valid = 0; break; }
            }
        }
    }
    if (a.debug_stop == 4) return;
    const unsigned long long vb = __builtin_amdgcn_ballot_w64(my_valid != 0);
    const int nsol = __popcll(vb);
    if (lane == 0) a.nsol[hyp] = failed ? -2 : nsol;
    if (nsol == 1 && my_valid) {  // DlsPnpWithRansac.h:62  accept iff exactly one solution; b_T_a column-major
        model[0] = R[0]; model[1] = R[3]; model[2] = R[6]; model[3] = 0.0;
        model[4] = R[1]; model[5] = R[4]; model[6] = R[7]; model[7] = 0.0;
        model[8] = R[2]; model[9] = R[5]; model[10] = R[8]; model[11] = 0.0;
        model[12] = t3[0]; model[13] = t3[1]; model[14] = t3[2]; model[15] = 1.0;
    }
    WAVE_SYNC();
    if (nsol != 1) {
        if (lane == 0) { a.valid[hyp] = 0; a.nin[hyp] = 0; a.cost[hyp] = INFINITY; }
        return;
    }
    // ================= Error over all N correspondences (DlsPnpWithRansac.h:75-99) + MLE cost =================
    double T[16];
#pragma unroll
    for (int e = 0; e < 16; e++) T[e] = model[e];
    if (lane < 16) a.T_out[hyp * 16 + lane] = model[lane];
    double acc = 0.0;
    int cnt = 0;
    // One wave walks the N correspondences 64 at a time.  The loads of block b+1 are issued before the arithmetic of block b (two IEEE
    // divisions per point): as a plain loop every block waited a full global-memory latency for its five loads.
    unsigned long long my_word = 0ull;
    double nX0 = 0.0, nX1 = 0.0, nX2 = 0.0, nu = 0.0, nv = 0.0;
    if (lane < pr.N) { nX0 = pr.X[3 * lane]; nX1 = pr.X[3 * lane + 1]; nX2 = pr.X[3 * lane + 2]; nu = pr.uv[2 * lane]; nv = pr.uv[2 * lane + 1]; }
    for (int base = 0; base < pr.N; base += 64) {
        const int i = base + lane;
        const double X0 = nX0, X1 = nX1, X2 = nX2, pu = nu, pv = nv;
        const int in_ = i + 64;
        if (in_ < pr.N) { nX0 = pr.X[3 * in_]; nX1 = pr.X[3 * in_ + 1]; nX2 = pr.X[3 * in_ + 2]; nu = pr.uv[2 * in_]; nv = pr.uv[2 * in_ + 1]; }
        bool in = false;
        if (i < pr.N) {
            const double xx = ((T[0] * X0 + T[4] * X1) + T[8] * X2) + T[12];
            const double yy = ((T[1] * X0 + T[5] * X1) + T[9] * X2) + T[13];
            const double zz = ((T[2] * X0 + T[6] * X1) + T[10] * X2) + T[14];
            const double xn = xx / zz, yn = yy / zz;
            const double rr = fabs(xn - pu) + fabs(yn - pv);
            in = rr < a.thresh;
            acc = acc + (in ? rr : a.thresh);
        }
        const unsigned long long bw = __builtin_amdgcn_ballot_w64(in);
        cnt += __popcll(bw);
        // The mask lives in pinned HOST memory.  Stores count in vmcnt like loads, so a store per block made the next block's
        // `s_waitcnt vmcnt` wait for a PCIe write round trip (~4 us x 8 blocks at N = 512: the whole scoring phase).  Lane w keeps
        // word w; 64 words go out in one store instruction.
        const int word = base >> 6;
        if (lane == (word & 63)) my_word = bw;
        if ((word & 63) == 63 || base + 64 >= pr.N) {
            const int w0 = word & ~63;
            if (w0 + lane <= word) a.mask[(size_t)hyp * a.mask_words + w0 + lane] = my_word;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc = acc + __shfl_xor(acc, m, 64);
    if (lane == 0) {
        a.valid[hyp] = 1;
        a.nin[hyp] = cnt;
        a.cost[hyp] = a.use_mle ? acc : (double)(pr.N - cnt);
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct PnpState {
    PnpTables *tab_dev = nullptr;
    // device scratch, grown on demand
    double *X = nullptr, *uv = nullptr;   // device: one allocation, [X of all problems | uv of all problems]
    double *h_in = nullptr;               // pinned, device-mapped staging of the same layout: pnp_build_solve reads it in place (d_hin)
    double *d_hin = nullptr;              // device view of h_in
    int32_t cap_N = 0;
    double *Sg = nullptr, *Tg = nullptr;
    int32_t *sample = nullptr, *ok = nullptr;
    int32_t *h_sample_in = nullptr, *d_sample_in = nullptr;   // CHIP_SAMPLER_THEIA_PERSISTENT: the host-sequenced sample table (pinned) + its device view
    std::vector<int32_t> perm;                                // ... and the permutation it is sequenced on
    int32_t cap_H = 0, cap_words = 0;
    // Per-hypothesis results live in pinned, device-mapped HOST memory: pnp_eig_score stores them straight across PCIe (a few
    // hundred KB per call, posted while the kernel runs), so a call needs no D2H copy and a single stream synchronisation.
    double *h_cost = nullptr, *h_T = nullptr;            // host views
    int32_t *h_nin = nullptr, *h_valid = nullptr, *h_nsol = nullptr;
    unsigned long long *h_mask = nullptr;
    double *cost = nullptr, *T_out = nullptr;            // device views of the same allocations
    int32_t *nin = nullptr, *valid = nullptr, *nsol = nullptr;
    unsigned long long *mask = nullptr;
    unsigned long long *stamps = nullptr;                // tuning only (CHIP_PNP_STAMPS): [H][8] of pnp_eig_score, then [H][16] of pnp_build_solve
    hipStream_t s2 = nullptr;                            // second launch-pair stream of a split batch (CHIP_PNP_GROUPS)
    hipEvent_t ev_in = nullptr, ev_g2 = nullptr;
    int32_t stamps_n = 0;
};

int pnp_create(Ctx *c)
{
    PnpState *st = new (std::nothrow) PnpState();
    if (!st) return CHIP_ERR_OOM;
    c->pnp_state = st;
    PnpTables t;
    std::memset(&t, 0, sizeof t);
    build_tables(t);
    CHIP_HIP(c, hipMalloc(&st->tab_dev, sizeof(PnpTables)));
    CHIP_HIP(c, hipMemcpy(st->tab_dev, &t, sizeof t, hipMemcpyHostToDevice));
    CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(pnp_build_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 102 * 1024));
    if (const char *e = std::getenv("CHIP_PNP_OCCUPANCY"); e && e[0] == '1') {   // tuning only: resident workgroups per CU of the kernel pair
        int nb = -1, ne = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(pnp_build_solve<false>), kSolveThreads, kSolveLds);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&ne, reinterpret_cast<const void *>(pnp_eig_score<false>), 64, 0);
        std::fprintf(stderr, "pnp occupancy: pnp_build_solve %d workgroups per CU, pnp_eig_score %d waves per CU\n", nb, ne);
    }
    CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(pnp_build_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 102 * 1024));
    return CHIP_OK;
}

static void pnp_free_dev(PnpState *st)
{
    (void)hipFree(st->X); (void)hipHostFree(st->h_in);   // uv lives inside the X allocation
    (void)hipFree(st->Sg); (void)hipFree(st->Tg);
    (void)hipFree(st->sample); (void)hipFree(st->ok);
    (void)hipHostFree(st->h_cost); (void)hipHostFree(st->h_T); (void)hipHostFree(st->h_nin); (void)hipHostFree(st->h_valid);
    (void)hipHostFree(st->h_nsol); (void)hipHostFree(st->h_mask); (void)hipHostFree(st->h_sample_in);
    (void)hipFree(st->stamps);
    if (st->s2) (void)hipStreamDestroy(st->s2);
    if (st->ev_in) (void)hipEventDestroy(st->ev_in);
    if (st->ev_g2) (void)hipEventDestroy(st->ev_g2);
}

void pnp_destroy(Ctx *c)
{
    PnpState *st = static_cast<PnpState *>(c->pnp_state);
    if (!st) return;
    pnp_free_dev(st);
    (void)hipFree(st->tab_dev);
    delete st;
    c->pnp_state = nullptr;
}

static int pnp_reserve(Ctx *c, PnpState *st, int N, int H, int words, int P)
{
    // hipFree waits for the whole device: not with a resident scan instance on it, and none launched by the tick thread until the last
    // allocation below is done (at 10 Hz the lease never runs out, the frees would wait for as long as ticks keep coming)
    ResidentPause paused(c, c->tick_resident && (N > st->cap_N || H > st->cap_H || words > st->cap_words));
    if (N > st->cap_N) {
        (void)hipFree(st->X); (void)hipHostFree(st->h_in);
        st->X = st->uv = st->h_in = nullptr; st->cap_N = 0;
        CHIP_HIP(c, hipMalloc(&st->X, sizeof(double) * 5 * (size_t)N));
        CHIP_HIP(c, hipHostMalloc(&st->h_in, sizeof(double) * 5 * (size_t)N, hipHostMallocDefault));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->d_hin, st->h_in, 0));
        st->cap_N = N;
    }
    if (H > st->cap_H || words > st->cap_words) {
        const int nh = H > st->cap_H ? H : st->cap_H, nw = words > st->cap_words ? words : st->cap_words;
        (void)hipFree(st->Sg); (void)hipFree(st->Tg); (void)hipFree(st->sample); (void)hipFree(st->ok);
        (void)hipHostFree(st->h_cost); (void)hipHostFree(st->h_T); (void)hipHostFree(st->h_nin); (void)hipHostFree(st->h_valid);
        (void)hipHostFree(st->h_nsol); (void)hipHostFree(st->h_mask); (void)hipHostFree(st->h_sample_in);
        st->h_sample_in = nullptr;
        st->Sg = st->Tg = nullptr; st->sample = st->ok = nullptr;
        st->h_cost = st->h_T = nullptr; st->h_nin = st->h_valid = st->h_nsol = nullptr; st->h_mask = nullptr;
        st->cap_H = st->cap_words = 0;
        CHIP_HIP(c, hipMalloc(&st->Sg, sizeof(double) * 729 * (size_t)nh));
        CHIP_HIP(c, hipMalloc(&st->Tg, sizeof(double) * 27 * (size_t)nh));
        CHIP_HIP(c, hipMalloc(&st->sample, sizeof(int32_t) * kSampleMax * (size_t)nh));
        CHIP_HIP(c, hipMalloc(&st->ok, sizeof(int32_t) * (size_t)nh));
        CHIP_HIP(c, hipHostMalloc(&st->h_cost, sizeof(double) * (size_t)nh, hipHostMallocDefault));
        CHIP_HIP(c, hipHostMalloc(&st->h_T, sizeof(double) * 16 * (size_t)nh, hipHostMallocDefault));
        CHIP_HIP(c, hipHostMalloc(&st->h_nin, sizeof(int32_t) * (size_t)nh, hipHostMallocDefault));
        CHIP_HIP(c, hipHostMalloc(&st->h_valid, sizeof(int32_t) * (size_t)nh, hipHostMallocDefault));
        CHIP_HIP(c, hipHostMalloc(&st->h_nsol, sizeof(int32_t) * (size_t)nh, hipHostMallocDefault));
        CHIP_HIP(c, hipHostMalloc(&st->h_mask, sizeof(unsigned long long) * (size_t)nh * nw, hipHostMallocDefault));
        CHIP_HIP(c, hipHostMalloc(&st->h_sample_in, sizeof(int32_t) * kSampleMax * (size_t)nh, hipHostMallocDefault));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->d_sample_in, st->h_sample_in, 0));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->cost, st->h_cost, 0));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->T_out, st->h_T, 0));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->nin, st->h_nin, 0));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->valid, st->h_valid, 0));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->nsol, st->h_nsol, 0));
        CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->mask, st->h_mask, 0));
        st->cap_H = nh; st->cap_words = nw;
    }
    (void)P;
    return CHIP_OK;
}

// Up to kPnpMaxBatch problems in one pair of launches; problem i's outputs are exactly those of a single-problem call.
static int pnp_run(Ctx *c, PnpState *st, int P, const double *const *X, const double *const *uv, const int32_t *N,
                   const chip_ransac_params *p, const uint64_t *seeds, double *T_colmajor, float *confidence,
                   uint8_t *const *inlier_mask, chip_ransac_summary *summary)
{
    // tuning only (CHIP_PNP_HOST_TIMING=1): host-side phases of a call, averaged, printed at process exit
    struct HostTiming {
        double acc[5] = {0, 0, 0, 0, 0}; long n = 0; bool on = std::getenv("CHIP_PNP_HOST_TIMING") != nullptr;
        ~HostTiming() { if (on && n) std::fprintf(stderr, "pnp host timing over %ld calls (us): prepare+H2D enqueue %.1f, launches %.1f, wait %.1f, select+copy out %.1f, total %.1f\n",
                                                   n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n); }
    };
    static HostTiming ht;
    const auto ht_now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double ht0 = ht.on ? ht_now() : 0.0;
    double ht1 = 0.0, ht2 = 0.0, ht3 = 0.0;
    const int32_t S = p->sample_size;
    const int H = ransac_initial_iterations(p);
    int Ntot = 0, words = 0;
    for (int i = 0; i < P; i++) { Ntot += N[i]; const int w = (N[i] + 63) / 64; words = w > words ? w : words; }
    int rc = pnp_reserve(c, st, Ntot, P * H, words, P);
    if (rc != CHIP_OK) return rc;
    hipStream_t s = c->s_pnp;

    SolveArgs sa;
    EigArgs ea;
    std::memset(&sa, 0, sizeof sa);
    std::memset(&ea, 0, sizeof ea);
    st->uv = st->X + 3 * (size_t)Ntot;
    for (int i = 0, off = 0; i < P; off += N[i], i++) {
        std::memcpy(st->h_in + 3 * (size_t)off, X[i], sizeof(double) * 3 * (size_t)N[i]);
        std::memcpy(st->h_in + 3 * (size_t)Ntot + 2 * (size_t)off, uv[i], sizeof(double) * 2 * (size_t)N[i]);
        PnpProblem pr;
        pr.X = st->X + 3 * (size_t)off; pr.uv = st->uv + 2 * (size_t)off; pr.N = N[i]; pr.pad_ = 0;
        pr.seed = seeds ? seeds[i] : p->seed;
        ea.prob[i] = pr;                                   // pnp_eig_score: the device copy (made by pnp_build_solve)
        pr.X = st->d_hin + 3 * (size_t)off; pr.uv = st->d_hin + 3 * (size_t)Ntot + 2 * (size_t)off;
        sa.prob[i] = pr;                                   // pnp_build_solve: the pinned host buffer, in place
    }
    // no H2D copy: see SolveArgs.  (CHIP_PNP_H2D=1 restores it for A/B runs.)
    static const bool want_h2d = std::getenv("CHIP_PNP_H2D") != nullptr;
    if (want_h2d) {
        CHIP_HIP(c, hipMemcpyAsync(st->X, st->h_in, sizeof(double) * 5 * (size_t)Ntot, hipMemcpyHostToDevice, s));
        for (int i = 0; i < P; i++) sa.prob[i] = ea.prob[i];
    }
    if (ht.on) ht1 = ht_now();
    if (p->sampler == CHIP_SAMPLER_THEIA_PERSISTENT) {   // one persistent permutation per problem, hypotheses 0..H-1 in order
        for (int i = 0; i < P; i++) {
            st->perm.resize((size_t)N[i]);
            ransac_sample_table_persistent(seeds ? seeds[i] : p->seed, H, N[i], S, kSampleMax, st->perm.data(), st->h_sample_in + (size_t)i * H * kSampleMax);
        }
        sa.sample_in = st->d_sample_in;
    }
    sa.H = H; sa.S = S; sa.tab = st->tab_dev;
    sa.Sg = st->Sg; sa.Tg = st->Tg; sa.sample = st->sample; sa.ok = st->ok;
    sa.in_host = want_h2d ? nullptr : st->d_hin; sa.in_dev = st->X; sa.n_in = 5 * (int64_t)Ntot;
    const size_t lds = kSolveLds;
    static const bool want_stamps = std::getenv("CHIP_PNP_STAMPS") != nullptr;
    if (want_stamps) {
        if (st->stamps_n < P * H) {
            (void)hipFree(st->stamps);
            st->stamps = nullptr; st->stamps_n = 0;
            CHIP_HIP(c, hipMalloc(&st->stamps, sizeof(unsigned long long) * 32 * (size_t)P * H));
            st->stamps_n = P * H;
        }
        CHIP_HIP(c, hipMemsetAsync(st->stamps, 0, sizeof(unsigned long long) * 32 * (size_t)st->stamps_n, s));
        sa.stamps = st->stamps + 8 * (size_t)st->stamps_n;
    }
    ea.H = H; ea.S = S; ea.thresh = p->error_thresh; ea.use_mle = p->use_mle;
    ea.Sg = st->Sg; ea.Tg = st->Tg; ea.sample = st->sample; ea.ok = st->ok; ea.mask_words = words;
    ea.debug_stop = 0;
    ea.backsub_loop = 0;
#ifdef CHIP_TEST_HOOKS   // knobs that change what the kernels compute or which code path they take: test build only (lib/hooks/)
    { const char *dv = std::getenv("CHIP_PNP_DEBUG_STOP"); ea.debug_stop = dv ? std::atoi(dv) : 0; }
    {
        static const int loop_form = [] {
            const char *e = std::getenv("CHIP_PNP_BACKSUB");
            const int on = e && std::strcmp(e, "loop") == 0;
            if (on) std::fprintf(stderr, "[cerebro_hip] TEST KNOB ACTIVE: CHIP_PNP_BACKSUB=loop -- pnp_eig_score runs the loop form of the back-substitution\n");
            return on;
        }();
        ea.backsub_loop = loop_form;
    }
#endif
    ea.T_out = st->T_out; ea.cost = st->cost; ea.nin = st->nin; ea.valid = st->valid; ea.nsol = st->nsol; ea.mask = st->mask;
    if (want_stamps) ea.stamps = st->stamps;
    // A batch of several problems may go out as `groups` launch pairs on as many streams (CHIP_PNP_GROUPS, tuning): both kernels are
    // latency chains that leave issue slots free, so build(group g+1) can run underneath eig(group g).  Measured NEGATIVE twice: two
    // groups -3 % (round 5), and the fully pipelined form -- pnp_build_solve of consecutive slot groups back to back on one stream,
    // pnp_eig_score(g) on a second, lower-priority stream behind build(g)'s event -- 1.7-2.1 M hyp/s against 2.52 M for one launch pair
    // (round 6, profiles/r06_pnp.md): the two kernels do not share a CU's REGISTER FILE -- four resident eigen waves (4 x 128 rows)
    // leave room for ONE pnp_build_solve workgroup (784 rows) instead of two, whatever the stream priorities say.
    static const int want_groups = [] { const char *e = std::getenv("CHIP_PNP_GROUPS"); return e ? std::atoi(e) : 1; }();
    static const int want_prio = [] { const char *e = std::getenv("CHIP_PNP_PRIO"); return e ? std::atoi(e) : 0; }();   // measured: no effect (profiles/r03_pnp_pmc.md)
    sa.factor_prio = want_prio;
    int groups = want_groups < 1 ? 1 : (want_groups > 2 ? 2 : want_groups);
    if (groups > P) groups = P;
    if (groups > 1 && !st->s2) {
        CHIP_HIP(c, hipStreamCreateWithFlags(&st->s2, hipStreamNonBlocking));
        CHIP_HIP(c, hipEventCreateWithFlags(&st->ev_in, hipEventDisableTiming));
        CHIP_HIP(c, hipEventCreateWithFlags(&st->ev_g2, hipEventDisableTiming));
    }
    if (groups > 1) CHIP_HIP(c, hipEventRecord(st->ev_in, s));   // inputs (and the stamp memset) are on the device
    for (int g = 0; g < groups; g++) {
        const int p0 = (int)((long long)P * g / groups), p1 = (int)((long long)P * (g + 1) / groups);
        hipStream_t sg = g == 0 ? s : st->s2;
        if (g > 0) CHIP_HIP(c, hipStreamWaitEvent(sg, st->ev_in, 0));
        sa.slot0 = ea.slot0 = p0 * H;
        const int nslots = (p1 - p0) * H;
        if (want_stamps) hipLaunchKernelGGL(pnp_build_solve<true>, dim3(nslots), dim3(kSolveThreads), lds, sg, sa);
        else hipLaunchKernelGGL(pnp_build_solve<false>, dim3(nslots), dim3(kSolveThreads), lds, sg, sa);
        CHIP_HIP(c, hipGetLastError());
        if (want_stamps) hipLaunchKernelGGL(pnp_eig_score<true>, dim3(nslots), dim3(64), 0, sg, ea);
        else hipLaunchKernelGGL(pnp_eig_score<false>, dim3(nslots), dim3(64), 0, sg, ea);
        CHIP_HIP(c, hipGetLastError());
        if (g > 0) { CHIP_HIP(c, hipEventRecord(st->ev_g2, sg)); CHIP_HIP(c, hipStreamWaitEvent(s, st->ev_g2, 0)); }
    }
    if (ht.on) ht2 = ht_now();
    CHIP_HIP(c, hipStreamSynchronize(s));   // every per-hypothesis result is in host memory now
    if (ht.on) ht3 = ht_now();

    // K7: theia::Ransac::Estimate's sequential rule replayed over the per-hypothesis results
    int32_t best_h[kPnpMaxBatch], num_it[kPnpMaxBatch], n_models[kPnpMaxBatch];
    double best_cost[kPnpMaxBatch];
    for (int i = 0; i < P; i++) {
        best_cost[i] = DBL_MAX; n_models[i] = 0; num_it[i] = 0;
        const size_t o = (size_t)i * H;
        best_h[i] = ransac_select(p, N[i], H, st->h_valid + o, st->h_cost + o, st->h_nin + o, &num_it[i], &n_models[i], &best_cost[i]);
    }
    for (int i = 0; i < P; i++) {
        double *T = T_colmajor + 16 * (size_t)i;
        uint8_t *im = inlier_mask ? inlier_mask[i] : nullptr;
        int32_t nin = 0;
        if (best_h[i] >= 0) {
            std::memcpy(T, st->h_T + 16 * ((size_t)i * H + best_h[i]), sizeof(double) * 16);
            nin = st->h_nin[(size_t)i * H + best_h[i]];
            const unsigned long long *hm = st->h_mask + ((size_t)i * H + best_h[i]) * words;
            if (im)
                for (int k = 0; k < N[i]; k++) im[k] = (uint8_t)((hm[k >> 6] >> (k & 63)) & 1ull);
            const double ratio = (double)nin / (double)N[i];
            confidence[i] = (float)(1.0 - std::pow(1.0 - std::pow(ratio, (double)S), (double)num_it[i]));  // summary.confidence (:240)
        } else {
            for (int k = 0; k < 16; k++) T[k] = NAN;   // reference: uninitialised Matrix4d (:204); caller NaN-checks
            if (im) std::memset(im, 0, (size_t)N[i]);
            confidence[i] = 0.0f;
        }
        if (summary) {
            summary[i].n_iterations = num_it[i];
            summary[i].n_inliers = nin;
            summary[i].best_hypothesis = best_h[i];
            summary[i].n_models = n_models[i];
            summary[i].best_cost = best_h[i] >= 0 ? best_cost[i] : INFINITY;
        }
    }
    if (ht.on) { const double t4 = ht_now(); ht.acc[0] += ht1 - ht0; ht.acc[1] += ht2 - ht1; ht.acc[2] += ht3 - ht2; ht.acc[3] += t4 - ht3; ht.acc[4] += t4 - ht0; ht.n++; }
    return CHIP_OK;
}

}  // namespace chip

using namespace chip;

// Tuning aid, not part of the ABI: the per-wave shader-clock totals of the most recent pnp_eig_score<true> launch (CHIP_PNP_STAMPS=1).
extern "C" int chip_debug_pnp_stamps(chip_ctx *c, unsigned long long *out, int32_t n_hyp)
{
    if (!c || !out) return CHIP_ERR_INVALID_ARG;
    if (c->group) c = static_cast<chip_ctx *>(chip::group_root(c));
    std::lock_guard<std::mutex> lk(c->pnp_mu);
    PnpState *st = static_cast<PnpState *>(c->pnp_state);
    if (!st || !st->stamps || n_hyp > st->stamps_n) return CHIP_ERR_INVALID_ARG;
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(c->s_pnp));
    CHIP_HIP(c, hipMemcpy(out, st->stamps, sizeof(unsigned long long) * 8 * (size_t)n_hyp, hipMemcpyDeviceToHost));
    return CHIP_OK;
}

extern "C" int chip_debug_pnp_solve_stamps(chip_ctx *c, unsigned long long *out, int32_t n_hyp)
{
    if (!c || !out) return CHIP_ERR_INVALID_ARG;
    if (c->group) c = static_cast<chip_ctx *>(chip::group_root(c));
    std::lock_guard<std::mutex> lk(c->pnp_mu);
    PnpState *st = static_cast<PnpState *>(c->pnp_state);
    if (!st || !st->stamps || n_hyp > st->stamps_n) return CHIP_ERR_INVALID_ARG;
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(c->s_pnp));
    CHIP_HIP(c, hipMemcpy(out, st->stamps + 8 * (size_t)st->stamps_n, sizeof(unsigned long long) * 24 * (size_t)n_hyp, hipMemcpyDeviceToHost));
    return CHIP_OK;
}

extern "C" int chip_pnp_ransac_batch(chip_ctx *c, int32_t P, const double *const *X, const double *const *uv, const int32_t *N,
                                     const chip_ransac_params *p, const uint64_t *seeds, double *T_colmajor, float *confidence,
                                     uint8_t *const *inlier_mask, chip_ransac_summary *summary)
{
    if (!c || P < 0 || (P > 0 && (!X || !uv || !N)) || !p || !T_colmajor || !confidence) return CHIP_ERR_INVALID_ARG;
    if (c->group) c = static_cast<chip_ctx *>(chip::group_root(c));   // 1k x 512 is far too small to shard: devices[0] (SURVEY 8e "replicas only")
    const int32_t S = p->sample_size;
    for (int i = 0; i < P; i++) {
        if (!X[i] || !uv[i]) return CHIP_ERR_INVALID_ARG;
        if (N[i] < 20) return CHIP_ERR_TOO_FEW_POINTS;  // DlsPnpWithRansac.cpp:136-139
        if (S > N[i]) return CHIP_ERR_UNSUPPORTED;
    }
    if (S < 3 || S > kSampleMax || p->n_hypotheses < 0 || p->max_iterations < 1) return CHIP_ERR_UNSUPPORTED;
    if (p->sampler != CHIP_SAMPLER_FRESH && p->sampler != CHIP_SAMPLER_THEIA_PERSISTENT) return CHIP_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(c->pnp_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    PnpState *st = static_cast<PnpState *>(c->pnp_state);
    if (!st) return CHIP_ERR_INVALID_ARG;
    for (int i0 = 0; i0 < P; i0 += kPnpMaxBatch) {
        const int n = P - i0 < kPnpMaxBatch ? P - i0 : kPnpMaxBatch;
        const int rc = pnp_run(c, st, n, X + i0, uv + i0, N + i0, p, seeds ? seeds + i0 : nullptr, T_colmajor + 16 * (size_t)i0,
                               confidence + i0, inlier_mask ? inlier_mask + i0 : nullptr, summary ? summary + i0 : nullptr);
        if (rc != CHIP_OK) return rc;
    }
    return CHIP_OK;
}

extern "C" int chip_pnp_ransac(chip_ctx *c, const double *X, const double *uv, int32_t N, const chip_ransac_params *p,
                               double T_colmajor[16], float *confidence, uint8_t *inlier_mask, chip_ransac_summary *summary)
{
    if (!c || !X || !uv || !p || !T_colmajor || !confidence) return CHIP_ERR_INVALID_ARG;
    uint8_t *masks[1] = {inlier_mask};
    return chip_pnp_ransac_batch(c, 1, &X, &uv, &N, p, nullptr, T_colmajor, confidence, masks, summary);
}
