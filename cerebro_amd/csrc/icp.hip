// icp.hip -- batched Umeyama-ICP-in-RANSAC on gfx950: replaces the RANSAC branch of
// StaticTheiaPoseCompute::P3P_ICP (/root/reference/src/DlsPnpWithRansac.cpp:65-121), i.e. theia::Ransac over the
// AlignPointCloudsUmeyamaWithRansac estimator (src/DlsPnpWithRansac.h:104-166).
//
//   icp_models    : ONE LANE per hypothesis (64 hypotheses per wave; round 5 -- rounds 1-4 ran one wave per hypothesis with all 64
//                   lanes computing the same 3x3 solve).  The lane draws its 10-point sample (the same counter-based partial
//                   Fisher-Yates as PnP, here with the sparse permutation map in the lane's own registers: step i's entry is
//                   (j_i, value), a lookup takes the LAST earlier entry with that key), gathers the sample points, and runs
//                   the Umeyama solve (means, covariance, fixed-sweep Jacobi SVD, S22 sign, scale: ~600 flops + 24 rotations
//                   with a division and two square roots each); accept iff min(s, 1/s) > 0.9 (:137); writes b_T_a + valid.
//   icp_score     : one wave per hypothesis: the L2 error of all N correspondences under that hypothesis's model, MLE cost in
//                   the fixed lane-strided + butterfly order, inlier words by __ballot.
//                   Both kernels keep the operation order of oracle/icp_ransac.c, so poses, costs and masks are the bits of
//                   rounds 1-4 (tests/test_icp_gpu.py); the model work per call drops 64-fold (8000 hypotheses: 168 -> see
//                   profiles/r05_icp.txt), the reference-mode call (<= 50 hypotheses = one wave of icp_models) keeps its latency.
//   K7 on the host: ransac_common.h (shared with PnP).
// fp64, -ffp-contract=off, same operation order as oracle/icp_ransac.c => bit-identical poses and masks.
#include "ransac_common.h"
#include <cstring>
#include <new>
#include <vector>

namespace chip {

struct IcpArgs {
    const double *A;   // N x 3 (frame a)
    const double *B;   // N x 3 (frame b)
    int32_t N, S;
    uint64_t seed;
    double thresh;
    int32_t use_mle;
    int32_t mask_words;
    double *T_out;     // [H][16]
    double *cost;      // [H]
    int32_t *nin;      // [H]
    int32_t *valid;    // [H]
    unsigned long long *mask;  // [H][mask_words]
    int32_t H;
    uint64_t magic[kSampleMax];   // floor(2^64 / (N - i)): the sampler's 64-bit modulo as a multiply-high (the divisors depend on i only)
    const int32_t *sample_in;   // CHIP_SAMPLER_THEIA_PERSISTENT: [H][kSampleMax] sequenced by the host (pinned, device-mapped); else nullptr
    double *T_dev;     // [H][16] device copy of the models: icp_score reads it (T_out is pinned HOST memory)
    int32_t *valid_dev;// [H]
};

constexpr int kJacobiSweeps = 8;

#define JROT(p, q)                                                                             \
    do {                                                                                       \
        const double apq = Am[p][q];                                                           \
        if (apq != 0.0) {                                                                      \
            const double theta = (Am[q][q] - Am[p][p]) / (2.0 * apq);                          \
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
            const double cc = 1.0 / sqrt(tt * tt + 1.0), ss = tt * cc;                         \
            for (int r = 0; r < 3; r++) {                                                      \
                const double arp = Am[r][p], arq = Am[r][q];                                   \
                Am[r][p] = cc * arp - ss * arq;                                                \
                Am[r][q] = ss * arp + cc * arq;                                                \
            }                                                                                  \
            for (int r = 0; r < 3; r++) {                                                      \
                const double apr = Am[p][r], aqr = Am[q][r];                                   \
                Am[p][r] = cc * apr - ss * aqr;                                                \
                Am[q][r] = ss * apr + cc * aqr;                                                \
            }                                                                                  \
            for (int r = 0; r < 3; r++) {                                                      \
                const double vrp = V[r][p], vrq = V[r][q];                                     \
                V[r][p] = cc * vrp - ss * vrq;                                                 \
                V[r][q] = ss * vrp + cc * vrq;                                                 \
            }                                                                                  \
        }                                                                                      \
    } while (0)

#define COLSWAP(i, j)                                                                          \
    do {                                                                                       \
        const double tw = w[i]; w[i] = w[j]; w[j] = tw;                                        \
        for (int r = 0; r < 3; r++) { const double tv = V[r][i]; V[r][i] = V[r][j]; V[r][j] = tv; } \
    } while (0)

// Sample of hypothesis `hyp` by ONE lane: theia::RandomSampler's partial Fisher-Yates over a virtual identity permutation
// (oracle/pnp_ransac.c orc_ransac_sample; ransac_common.h has the wave-cooperative form).  Step i swaps positions i and
// j_i = i + draw_i % (N - i).  Position i is final after step i and never read again, so only the j-targets need remembering:
// entry e = (key j_e, the value step e wrote there); the current value of a position is that of the LAST earlier entry with
// that key, else the position itself.  Fully unrolled, so keys / values stay in registers.
// x % d by Barrett reduction with the host's m = floor(2^64 / d): q = mulhi(x, m) underestimates x / d by at most 2, so the
// remainder needs at most two corrections -- the exact x % d of the oracle (hipcc's generic 64-bit division is ~150 instructions
// per draw and lane; the divisors N - i are the same for every hypothesis).
__device__ __forceinline__ uint64_t mod_magic(uint64_t x, uint64_t d, uint64_t m)
{
    uint64_t r = x - __umul64hi(x, m) * d;
    if (r >= d) r -= d;
    if (r >= d) r -= d;
    return r;
}

__device__ __forceinline__ void ransac_sample_lane(uint64_t seed, int hyp, int N, int S, const uint64_t *magic, int smp[kSampleMax])
{
    int key[kSampleMax], val[kSampleMax];
#pragma unroll
    for (int i = 0; i < kSampleMax; i++) {
        key[i] = -1; val[i] = 0; smp[i] = 0;
        if (i < S) {
            const uint64_t x = rng_draw(seed, (uint32_t)hyp, (uint32_t)i);
            const int j = i + (int)mod_magic(x, (uint64_t)(N - i), magic[i]);
            int vi = i, vj = j;
#pragma unroll
            for (int e = 0; e < i; e++) {
                if (key[e] == i) vi = val[e];
                if (key[e] == j) vj = val[e];
            }
            smp[i] = vj;          // idx[i] <- old idx[j]
            key[i] = j;           // idx[j] <- old idx[i]   (j == i: vj == vi, a no-op as in the reference)
            val[i] = vi;
        }
    }
}

__global__ __launch_bounds__(64) void icp_models(IcpArgs a)
{
    const int hyp = blockIdx.x * 64 + threadIdx.x;
    if (hyp >= a.H) return;
    const int n = a.S;
    int smp[kSampleMax];
    if (a.sample_in) {
#pragma unroll
        for (int i = 0; i < kSampleMax; i++) smp[i] = i < n ? a.sample_in[(size_t)hyp * kSampleMax + i] : 0;
    } else {
        ransac_sample_lane(a.seed, hyp, a.N, n, a.magic, smp);
    }

    // ---- Umeyama on the sample ----
    double ma[3] = {0.0, 0.0, 0.0}, mb[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < kSampleMax; i++)
        if (i < n)
            for (int k = 0; k < 3; k++) { ma[k] = ma[k] + a.A[3 * smp[i] + k]; mb[k] = mb[k] + a.B[3 * smp[i] + k]; }
    for (int k = 0; k < 3; k++) { ma[k] = ma[k] / (double)n; mb[k] = mb[k] / (double)n; }
    double Sg[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}, var_a = 0.0;
#pragma unroll
    for (int i = 0; i < kSampleMax; i++)
        if (i < n) {
            double da[3], db[3];
            for (int k = 0; k < 3; k++) { da[k] = a.A[3 * smp[i] + k] - ma[k]; db[k] = a.B[3 * smp[i] + k] - mb[k]; }
            var_a = var_a + ((da[0] * da[0] + da[1] * da[1]) + da[2] * da[2]);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) Sg[r][c] = Sg[r][c] + db[r] * da[c];
        }
    var_a = var_a / (double)n;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Sg[r][c] = Sg[r][c] / (double)n;
    double Am[3][3], V[3][3], w[3], sig[3], U[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            Am[r][c] = (Sg[0][r] * Sg[0][c] + Sg[1][r] * Sg[1][c]) + Sg[2][r] * Sg[2][c];
            V[r][c] = (r == c) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < kJacobiSweeps; sweep++) { JROT(0, 1); JROT(0, 2); JROT(1, 2); }
    w[0] = Am[0][0]; w[1] = Am[1][1]; w[2] = Am[2][2];
    // descending selection sort (ties keep the lower index first), same comparisons as the oracle
    if (w[1] > w[0]) COLSWAP(0, 1);
    if (w[2] > w[0]) COLSWAP(0, 2);
    if (w[2] > w[1]) COLSWAP(1, 2);
    {
        const double det = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) - V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                           V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
        if (det < 0.0)
            for (int r = 0; r < 3; r++) V[r][2] = -V[r][2];
    }
    for (int k = 0; k < 3; k++) sig[k] = sqrt(w[k] > 0.0 ? w[k] : 0.0);
    bool ok = (sig[1] > 1e-6 * sig[0]) && (sig[0] > 0.0);
    double R[9], t[3], scale = 0.0;
    for (int e = 0; e < 9; e++) R[e] = 0.0;
    t[0] = t[1] = t[2] = 0.0;
    if (ok) {
        for (int k = 0; k < 2; k++)
            for (int r = 0; r < 3; r++) U[r][k] = ((Sg[r][0] * V[0][k] + Sg[r][1] * V[1][k]) + Sg[r][2] * V[2][k]) / sig[k];
        double S22 = 1.0;
        if (sig[2] > 1e-6 * sig[0]) {
            for (int r = 0; r < 3; r++) U[r][2] = ((Sg[r][0] * V[0][2] + Sg[r][1] * V[1][2]) + Sg[r][2] * V[2][2]) / sig[2];
            const double detU = U[0][0] * (U[1][1] * U[2][2] - U[1][2] * U[2][1]) - U[0][1] * (U[1][0] * U[2][2] - U[1][2] * U[2][0]) +
                                U[0][2] * (U[1][0] * U[2][1] - U[1][1] * U[2][0]);
            if (detU < 0.0) S22 = -1.0;
        } else {
            U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
            U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
            U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
        }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) R[3 * r + c] = (U[r][0] * V[c][0] + U[r][1] * V[c][1]) + (S22 * U[r][2]) * V[c][2];
        scale = ((sig[0] + sig[1]) + S22 * sig[2]) / var_a;
        for (int r = 0; r < 3; r++) t[r] = mb[r] - scale * ((R[3 * r] * ma[0] + R[3 * r + 1] * ma[1]) + R[3 * r + 2] * ma[2]);
        const double inv = 1.0 / scale;
        ok = (scale < inv ? scale : inv) > 0.9;   // DlsPnpWithRansac.h:137
    }
    a.valid_dev[hyp] = ok ? 1 : 0;
    if (!ok) { a.valid[hyp] = 0; a.nin[hyp] = 0; a.cost[hyp] = INFINITY; return; }
    double T[16];
    T[0] = R[0]; T[1] = R[3]; T[2] = R[6]; T[3] = 0.0;
    T[4] = R[1]; T[5] = R[4]; T[6] = R[7]; T[7] = 0.0;
    T[8] = R[2]; T[9] = R[5]; T[10] = R[8]; T[11] = 0.0;
    T[12] = t[0]; T[13] = t[1]; T[14] = t[2]; T[15] = 1.0;
    for (int e = 0; e < 16; e++) { a.T_dev[(size_t)hyp * 16 + e] = T[e]; a.T_out[(size_t)hyp * 16 + e] = T[e]; }
}

__global__ __launch_bounds__(64) void icp_score(IcpArgs a)
{
    const int lane = threadIdx.x;
    const int hyp = blockIdx.x;
    if (!a.valid_dev[hyp]) return;       // wave-uniform; icp_models has written valid / nin / cost of a rejected hypothesis
    double T[16];
    for (int e = 0; e < 16; e++) T[e] = a.T_dev[(size_t)hyp * 16 + e];
    // ---- Error (L2, DlsPnpWithRansac.h:152-164) over all N + MLE cost ----
    double acc = 0.0;
    int cnt = 0;
    for (int base = 0; base < a.N; base += 64) {
        const int i = base + lane;
        bool in = false;
        if (i < a.N) {
            const double a0 = a.A[3 * i], a1 = a.A[3 * i + 1], a2 = a.A[3 * i + 2];
            const double x = ((T[0] * a0 + T[4] * a1) + T[8] * a2) + T[12];
            const double y = ((T[1] * a0 + T[5] * a1) + T[9] * a2) + T[13];
            const double z = ((T[2] * a0 + T[6] * a1) + T[10] * a2) + T[14];
            const double dx = x - a.B[3 * i], dy = y - a.B[3 * i + 1], dz = z - a.B[3 * i + 2];
            const double rr = sqrt((dx * dx + dy * dy) + dz * dz);
            in = rr < a.thresh;
            acc = acc + (in ? rr : a.thresh);
        }
        const unsigned long long bw = __ballot(in);
        cnt += __popcll(bw);
        if (lane == 0) a.mask[(size_t)hyp * a.mask_words + (base >> 6)] = bw;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc = acc + __shfl_xor(acc, m, 64);
    if (lane == 0) {
        a.valid[hyp] = 1;
        a.nin[hyp] = cnt;
        a.cost[hyp] = a.use_mle ? acc : (double)(a.N - cnt);
    }
}

struct IcpState {
    double *A = nullptr, *B = nullptr, *T_out = nullptr, *cost = nullptr, *T_dev = nullptr;
    int32_t *nin = nullptr, *valid = nullptr, *valid_dev = nullptr;
    int32_t *h_sample_in = nullptr, *d_sample_in = nullptr;   // persistent-sampler table (pinned host + device view)
    std::vector<int32_t> perm;
    unsigned long long *mask = nullptr;
    int32_t cap_N = 0, cap_H = 0, cap_words = 0;
    double *h_cost = nullptr, *h_T = nullptr;
    int32_t *h_nin = nullptr, *h_valid = nullptr;
    unsigned long long *h_mask = nullptr;
    hipStream_t stream = nullptr;      // the ICP stream (not the PnP stream: the two estimations may overlap)
    bool pending = false;              // an enqueued estimation awaits chip_icp_ransac_collect
    int32_t pend_N = 0, pend_H = 0;
    chip_ransac_params pend_params{};
};

static void icp_free(IcpState *st)
{
    (void)hipFree(st->A); (void)hipFree(st->B); (void)hipFree(st->T_dev); (void)hipFree(st->valid_dev);   // T_out/cost/nin/valid/mask are device views of the pinned host buffers below
    (void)hipHostFree(st->h_cost); (void)hipHostFree(st->h_T); (void)hipHostFree(st->h_nin); (void)hipHostFree(st->h_valid); (void)hipHostFree(st->h_mask); (void)hipHostFree(st->h_sample_in);
    const hipStream_t keep = st->stream;   // buffers are regrown, the stream lives as long as the ctx
    *st = IcpState();
    st->stream = keep;
}

void icp_destroy(Ctx *c)
{
    IcpState *st = static_cast<IcpState *>(c->icp_state);
    if (!st) return;
    if (st->stream) { (void)hipStreamSynchronize(st->stream); (void)hipStreamDestroy(st->stream); st->stream = nullptr; }
    icp_free(st);
    delete st;
    c->icp_state = nullptr;
}

static int icp_reserve(Ctx *c, IcpState *st, int N, int H)
{
    const int words = (N + 63) / 64;
    if (N <= st->cap_N && H <= st->cap_H && words <= st->cap_words) return CHIP_OK;
    const int nN = N > st->cap_N ? N : st->cap_N, nH = H > st->cap_H ? H : st->cap_H, nW = words > st->cap_words ? words : st->cap_words;
    ResidentPause paused(c, c->tick_resident);   // hipFree waits for the whole device: no resident scan instance on it until the last allocation is done
    icp_free(st);
    CHIP_HIP(c, hipMalloc(&st->A, sizeof(double) * 3 * (size_t)nN));
    CHIP_HIP(c, hipMalloc(&st->B, sizeof(double) * 3 * (size_t)nN));
    CHIP_HIP(c, hipMalloc(&st->T_dev, sizeof(double) * 16 * (size_t)nH));
    CHIP_HIP(c, hipMalloc(&st->valid_dev, sizeof(int32_t) * (size_t)nH));
    // per-hypothesis results go straight to pinned, device-mapped host memory (as in pnp.hip): no D2H copies, one sync per call
    CHIP_HIP(c, hipHostMalloc(&st->h_cost, sizeof(double) * (size_t)nH, hipHostMallocDefault));
    CHIP_HIP(c, hipHostMalloc(&st->h_T, sizeof(double) * 16 * (size_t)nH, hipHostMallocDefault));
    CHIP_HIP(c, hipHostMalloc(&st->h_nin, sizeof(int32_t) * (size_t)nH, hipHostMallocDefault));
    CHIP_HIP(c, hipHostMalloc(&st->h_valid, sizeof(int32_t) * (size_t)nH, hipHostMallocDefault));
    CHIP_HIP(c, hipHostMalloc(&st->h_mask, sizeof(unsigned long long) * (size_t)nH * nW, hipHostMallocDefault));
    CHIP_HIP(c, hipHostMalloc(&st->h_sample_in, sizeof(int32_t) * kSampleMax * (size_t)nH, hipHostMallocDefault));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->d_sample_in, st->h_sample_in, 0));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->cost, st->h_cost, 0));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->T_out, st->h_T, 0));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->nin, st->h_nin, 0));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->valid, st->h_valid, 0));
    CHIP_HIP(c, hipHostGetDevicePointer((void **)&st->mask, st->h_mask, 0));
    st->cap_N = nN; st->cap_H = nH; st->cap_words = nW;
    return CHIP_OK;
}

}  // namespace chip

using namespace chip;

extern "C" void chip_icp_params_default(chip_ransac_params *p)
{
    if (!p) return;
    chip_ransac_params_default(p);
    p->error_thresh = 0.1;  // DlsPnpWithRansac.cpp:89
    p->sample_size = 10;    // DlsPnpWithRansac.h:118
}

// The estimation is split in two so that it can run underneath something else (its kernel is tiny and the PnP kernels leave
// the GPU 92-95 % idle): enqueue = H2D of the points + the kernel on the ICP stream, no synchronisation; collect = wait, replay
// theia's selection rule, fetch the winner.  chip_icp_ransac is enqueue + collect.
static int icp_enqueue_locked(chip_ctx *c, const double *A, const double *B, int32_t N, const chip_ransac_params *p)
{
    CHIP_HIP(c, hipSetDevice(c->device));
    if (!c->icp_state) {
        c->icp_state = new (std::nothrow) IcpState();
        if (!c->icp_state) return CHIP_ERR_OOM;
    }
    IcpState *st = static_cast<IcpState *>(c->icp_state);
    if (st->pending) return CHIP_ERR_BUSY;
    if (!st->stream) CHIP_HIP(c, hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
    const int32_t S = p->sample_size;
    const int H = ransac_initial_iterations(p);
    int rc = icp_reserve(c, st, N, H);
    if (rc != CHIP_OK) return rc;
    const int words = (N + 63) / 64;
    hipStream_t s = st->stream;
    CHIP_HIP(c, hipMemcpyAsync(st->A, A, sizeof(double) * 3 * (size_t)N, hipMemcpyHostToDevice, s));
    CHIP_HIP(c, hipMemcpyAsync(st->B, B, sizeof(double) * 3 * (size_t)N, hipMemcpyHostToDevice, s));
    IcpArgs a;
    a.A = st->A; a.B = st->B; a.N = N; a.S = S; a.seed = p->seed; a.thresh = p->error_thresh; a.use_mle = p->use_mle;
    a.mask_words = words; a.T_out = st->T_out; a.cost = st->cost; a.nin = st->nin; a.valid = st->valid; a.mask = st->mask;
    a.H = H; a.T_dev = st->T_dev; a.valid_dev = st->valid_dev;
    a.sample_in = nullptr;
    if (p->sampler == CHIP_SAMPLER_THEIA_PERSISTENT) {
        st->perm.resize((size_t)N);
        ransac_sample_table_persistent(p->seed, H, N, S, kSampleMax, st->perm.data(), st->h_sample_in);
        a.sample_in = st->d_sample_in;
    }
    for (int i = 0; i < kSampleMax; i++) {   // N - i >= 20 - 16 > 1 (icp_check_args), so the quotient fits 64 bits
        const uint64_t d = (uint64_t)(N - (i < S ? i : 0));
        a.magic[i] = (uint64_t)((((unsigned __int128)1) << 64) / d);
    }
    hipLaunchKernelGGL(icp_models, dim3((H + 63) / 64), dim3(64), 0, s, a);    // lane = hypothesis
    CHIP_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(icp_score, dim3(H), dim3(64), 0, s, a);                 // wave = hypothesis
    CHIP_HIP(c, hipGetLastError());
    st->pending = true;
    st->pend_N = N; st->pend_H = H; st->pend_params = *p;
    return CHIP_OK;
}

static int icp_collect_locked(chip_ctx *c, double T_colmajor[16], float *confidence, uint8_t *inlier_mask, chip_ransac_summary *summary)
{
    IcpState *st = static_cast<IcpState *>(c->icp_state);
    if (!st || !st->pending) return CHIP_ERR_BUSY;
    CHIP_HIP(c, hipSetDevice(c->device));
    CHIP_HIP(c, hipStreamSynchronize(st->stream));
    st->pending = false;
    const chip_ransac_params *p = &st->pend_params;
    const int32_t N = st->pend_N, S = p->sample_size;
    const int H = st->pend_H, words = (N + 63) / 64;
    double best_cost = DBL_MAX;
    int32_t n_models = 0, num_it = 0;
    const int32_t best_h = ransac_select(p, N, H, st->h_valid, st->h_cost, st->h_nin, &num_it, &n_models, &best_cost);
    int32_t nin = 0;
    if (best_h >= 0) {
        std::memcpy(T_colmajor, st->h_T + 16 * (size_t)best_h, sizeof(double) * 16);
        nin = st->h_nin[best_h];
        const unsigned long long *hm = st->h_mask + (size_t)best_h * words;
        if (inlier_mask)
            for (int i = 0; i < N; i++) inlier_mask[i] = (uint8_t)((hm[i >> 6] >> (i & 63)) & 1ull);
        const double ratio = (double)nin / (double)N;
        *confidence = (float)(1.0 - std::pow(1.0 - std::pow(ratio, (double)S), (double)num_it));  // summary.confidence (:121)
    } else {
        for (int i = 0; i < 16; i++) T_colmajor[i] = NAN;
        if (inlier_mask) std::memset(inlier_mask, 0, (size_t)N);
        *confidence = 0.0f;
    }
    if (summary) {
        summary->n_iterations = num_it;
        summary->n_inliers = nin;
        summary->best_hypothesis = best_h;
        summary->n_models = n_models;
        summary->best_cost = best_h >= 0 ? best_cost : INFINITY;
    }
    return CHIP_OK;
}

static int icp_check_args(const double *A, const double *B, int32_t N, const chip_ransac_params *p)
{
    if (!A || !B || !p) return CHIP_ERR_INVALID_ARG;
    if (N < 20) return CHIP_ERR_TOO_FEW_POINTS;  // DlsPnpWithRansac.cpp:19-22
    const int32_t S = p->sample_size;
    if (S < 3 || S > kSampleMax || S > N || p->n_hypotheses < 0 || p->max_iterations < 1) return CHIP_ERR_UNSUPPORTED;
    if (p->sampler != CHIP_SAMPLER_FRESH && p->sampler != CHIP_SAMPLER_THEIA_PERSISTENT) return CHIP_ERR_UNSUPPORTED;
    return CHIP_OK;
}

extern "C" int chip_icp_ransac_enqueue(chip_ctx *c, const double *A, const double *B, int32_t N, const chip_ransac_params *p)
{
    if (!c) return CHIP_ERR_INVALID_ARG;
    if (c->group) c = static_cast<chip_ctx *>(chip::group_root(c));
    const int rc = icp_check_args(A, B, N, p);
    if (rc != CHIP_OK) return rc;
    std::lock_guard<std::mutex> lk(c->icp_mu);
    return icp_enqueue_locked(c, A, B, N, p);
}

extern "C" int chip_icp_ransac_collect(chip_ctx *c, double T_colmajor[16], float *confidence, uint8_t *inlier_mask, chip_ransac_summary *summary)
{
    if (!c || !T_colmajor || !confidence) return CHIP_ERR_INVALID_ARG;
    if (c->group) c = static_cast<chip_ctx *>(chip::group_root(c));
    std::lock_guard<std::mutex> lk(c->icp_mu);
    return icp_collect_locked(c, T_colmajor, confidence, inlier_mask, summary);
}

extern "C" int chip_icp_ransac(chip_ctx *c, const double *A, const double *B, int32_t N, const chip_ransac_params *p,
                               double T_colmajor[16], float *confidence, uint8_t *inlier_mask, chip_ransac_summary *summary)
{
    if (!c || !T_colmajor || !confidence) return CHIP_ERR_INVALID_ARG;
    if (c->group) c = static_cast<chip_ctx *>(chip::group_root(c));
    int rc = icp_check_args(A, B, N, p);
    if (rc != CHIP_OK) return rc;
    std::lock_guard<std::mutex> lk(c->icp_mu);
    rc = icp_enqueue_locked(c, A, B, N, p);
    if (rc != CHIP_OK) return rc;
    return icp_collect_locked(c, T_colmajor, confidence, inlier_mask, summary);
}
