// batch.hip -- many-query batched mode (BASELINE north_star: "MFMA used only for the batched
// descriptor x database GEMM where it is genuinely a dense fp32 contraction").
//
// Semantics: the fp32 inner-product search of the reference's compiled-out faiss variants (faiss::IndexFlatIP on
// X.cast<float>(), top-k labels; /root/reference/src/Cerebro.cpp:390,422,455-472) for Q queries at once:
//   score(q, row) = one k-ordered fp32 fmaf chain over the D elements   (oracle: orc_dot_fmaf_f32)
//   top-k per query ordered (score desc, index desc).
// With Q >= ~40 queries per DB pass the arithmetic intensity (Q/2 flop/B) crosses the 19.7 flop/B ridge, so the work
// belongs on the matrix cores: v_mfma_f32_32x32x2_f32 is exact fp32 (bit-for-bit a k-ordered fmaf chain,
// cdna_hip_programming.md 3) at the 157 TFLOP/s vector rate.
//
// K_B db_gemm_topk : grid (P partitions of the DB prefix) x (Qpad/128 query tiles), 256 threads = 4 waves (2x2), each
//   wave owns a 64x64 block of the 128 (queries) x 128 (DB rows) tile as 2x2 MFMA 32x32 accumulators.  K is streamed in
//   chunks of 32 through LDS (row stride 33 floats: conflict-free ds_read_b32 fragment reads).  After the full K loop the
//   tile is transposed through the same LDS in two 64-query halves and 64 "owner" threads (one per query) scan the 128
//   new scores against their register-resident sorted top-K list.  Per partition and query one sorted list is written;
//   the lists are merged by topk_merge_batch (the shared merge_sorted_lists).
#include "chip_internal.h"
#include "topk_merge.h"
#include <cmath>
#include <new>

namespace chip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128;
constexpr int CT_LD = BN + 1;   // transposed score half-tile [64][129]

struct BatchArgs {
    const float *const *seg_table;
    int32_t seg_shift;
    int64_t seg_mask;
    int64_t n_rows;         // local rows [0, n_rows)
    int32_t D;
    const float *Q;         // [Qpad][D] device, rows >= Q are zero
    int32_t Qpad;
    int32_t K;
    int64_t rows_per_part;  // multiple of BN
    int64_t idx_mul, idx_add;
    chip_topk_entry *partial;  // [P][Qpad][K]
};

struct TopList {  // sorted (score desc, local row desc); empty slots (-inf, -1)
    float s[CHIP_MAX_TOPK];
    int32_t r[CHIP_MAX_TOPK];
};

__device__ __forceinline__ void list_init(TopList &L)
{
#pragma unroll
    for (int j = 0; j < CHIP_MAX_TOPK; j++) { L.s[j] = -INFINITY; L.r[j] = -1; }
}
__device__ __forceinline__ bool fkey_gt(float s, int32_t i, float s2, int32_t i2) { return s > s2 || (s == s2 && i > i2); }
__device__ __forceinline__ void list_push(TopList &L, int K, float s, int32_t row)
{
    // caller checked fkey_gt(s,row, L.s[K-1], L.r[K-1]); static-index insertion (no dynamic register indexing)
#pragma unroll
    for (int j = CHIP_MAX_TOPK - 1; j >= 1; j--) {
        if (j < K) {
            const bool above = fkey_gt(s, row, L.s[j - 1], L.r[j - 1]);       // new entry belongs above slot j-1
            const bool here = fkey_gt(s, row, L.s[j], L.r[j]) && !above;       // exactly at slot j
            if (above) { L.s[j] = L.s[j - 1]; L.r[j] = L.r[j - 1]; }
            else if (here) { L.s[j] = s; L.r[j] = row; }
        }
    }
    if (fkey_gt(s, row, L.s[0], L.r[0])) { L.s[0] = s; L.r[0] = row; }
}

template <int KC>
__global__ __launch_bounds__(256) void db_gemm_topk(BatchArgs a)
{
    constexpr int LDT = KC + 1;          // padded LDS row stride (floats): conflict-free ds_read_b32 fragment reads
    constexpr int LPR = KC / 4;          // lanes per row chunk (16 B each)
    constexpr int RPP = 256 / LPR;       // rows per staging pass
    constexpr int NPASS = BM / RPP;      // staging passes per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *As = reinterpret_cast<float *>(smem);        // [BM][LDT]
    float *Bs = As + BM * LDT;                           // [BN][LDT]
    float *Ct = reinterpret_cast<float *>(smem);         // [64][CT_LD] (aliases As/Bs after the K loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int D = a.D, K = a.K;
    const int q0 = blockIdx.y * BM;
    const int64_t part_lo = (int64_t)blockIdx.x * a.rows_per_part;
    int64_t part_hi = part_lo + a.rows_per_part;
    if (part_hi > a.n_rows) part_hi = a.n_rows;

    TopList L0, L1;   // owner thread t < 64: queries q0 + t and q0 + 64 + t
    list_init(L0);
    list_init(L1);

    // staging map: LPR consecutive lanes cover one row's K-chunk (KC*4 bytes: full-line, coalesced requests); NPASS passes
    const int ld_c = (tid % LPR) * 4, ld_r = tid / LPR;
    const float *arow[NPASS];
#pragma unroll
    for (int u = 0; u < NPASS; u++) arow[u] = a.Q + (int64_t)(q0 + ld_r + RPP * u) * D + ld_c;

    for (int64_t n0 = part_lo; n0 < part_hi; n0 += BN) {
        const float *brow[NPASS];
        bool bvalid[NPASS];
#pragma unroll
        for (int u = 0; u < NPASS; u++) {
            const int64_t br = n0 + ld_r + RPP * u;
            bvalid[u] = br < a.n_rows;
            const int64_t brc = bvalid[u] ? br : 0;
            brow[u] = a.seg_table[brc >> a.seg_shift] + (brc & a.seg_mask) * (int64_t)D + ld_c;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

        // software pipeline: the global loads of chunk kc+KC are issued before the MFMA loop of chunk kc
        f32x4 av[NPASS], bv[NPASS];
#pragma unroll
        for (int u = 0; u < NPASS; u++) {
            av[u] = *reinterpret_cast<const f32x4 *>(arow[u]);
            bv[u] = bvalid[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(brow[u])) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int kc = 0; kc < D; kc += KC) {
            __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
            for (int u = 0; u < NPASS; u++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    As[(ld_r + RPP * u) * LDT + ld_c + c] = av[u][c];
                    Bs[(ld_r + RPP * u) * LDT + ld_c + c] = bv[u][c];
                }
            if (kc + KC < D) {
#pragma unroll
                for (int u = 0; u < NPASS; u++) {
                    av[u] = *reinterpret_cast<const f32x4 *>(arow[u] + kc + KC);
                    bv[u] = bvalid[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(brow[u] + kc + KC)) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __syncthreads();
            const int fr = lane & 31, fk = lane >> 5;
#pragma unroll
            for (int kk = 0; kk < KC / 2; kk++) {
                const float a0 = As[(wm * 64 + fr) * LDT + 2 * kk + fk];
                const float a1 = As[(wm * 64 + 32 + fr) * LDT + 2 * kk + fk];
                const float b0 = Bs[(wn * 64 + fr) * LDT + 2 * kk + fk];
                const float b1 = Bs[(wn * 64 + 32 + fr) * LDT + 2 * kk + fk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        // ---- epilogue: two 64-query halves through LDS; C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int ncols = (part_hi - n0) < BN ? (int)(part_hi - n0) : BN;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            __syncthreads();   // As/Bs (or the previous half) no longer read
            if (wm == h) {
#pragma unroll
                for (int it = 0; it < 2; it++)
#pragma unroll
                    for (int jt = 0; jt < 2; jt++)
#pragma unroll
                        for (int e = 0; e < 16; e++) {
                            const int row = it * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);   // query within the half
                            const int col = wn * 64 + jt * 32 + (lane & 31);                        // DB row within the tile
                            Ct[row * CT_LD + col] = acc[it][jt][e];
                        }
            }
            __syncthreads();
            if (tid < 64) {
                TopList &L = h == 0 ? L0 : L1;
                float ts = L.s[0];
                int32_t tr = L.r[0];
#pragma unroll
                for (int j = 1; j < CHIP_MAX_TOPK; j++)
                    if (j < K) { ts = L.s[j]; tr = L.r[j]; }   // current K-th best
                for (int c = 0; c < ncols; c++) {
                    const float s = Ct[tid * CT_LD + c];
                    const int32_t row = (int32_t)(n0 + c);
                    if (fkey_gt(s, row, ts, tr)) {   // NaN never enters
                        list_push(L, K, s, row);
                        ts = L.s[0]; tr = L.r[0];
#pragma unroll
                        for (int j = 1; j < CHIP_MAX_TOPK; j++)
                            if (j < K) { ts = L.s[j]; tr = L.r[j]; }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid < 64) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const TopList &L = h == 0 ? L0 : L1;
            chip_topk_entry *dst = a.partial + ((int64_t)blockIdx.x * a.Qpad + q0 + h * 64 + tid) * K;
#pragma unroll
            for (int j = 0; j < CHIP_MAX_TOPK; j++)
                if (j < K) {
                    chip_topk_entry e;
                    e.score = (double)L.s[j];
                    e.idx = L.r[j] >= 0 ? (int64_t)L.r[j] * a.idx_mul + a.idx_add : -1;
                    dst[j] = e;
                }
        }
    }
}

// one workgroup per 4 queries: merge the P partition lists
struct BatchMergeArgs {
    const chip_topk_entry *in;  // [P][Qpad][K]
    int32_t n_lists, Qpad, K;
    chip_topk_entry *out;       // [Qpad][K]
};
__global__ __launch_bounds__(512) void topk_merge_batch(BatchMergeArgs a)
{
    __shared__ __attribute__((aligned(16))) char smem[kMergeSmem];
    merge_sorted_lists<4>(a.in, a.n_lists, a.Qpad, 4 * (int)blockIdx.x, a.K, a.out + (int64_t)4 * blockIdx.x * a.K, nullptr, 0, 0, 0.0, smem);
}

struct BatchState {
    float *Q = nullptr;
    chip_topk_entry *partial = nullptr, *out = nullptr, *h_out = nullptr;
    int64_t cap_q = 0, cap_partial = 0, cap_out = 0;
};

void batch_destroy(Ctx *c)
{
    BatchState *st = static_cast<BatchState *>(c->batch_state);
    if (!st) return;
    (void)hipFree(st->Q); (void)hipFree(st->partial); (void)hipFree(st->out); (void)hipHostFree(st->h_out);
    delete st;
    c->batch_state = nullptr;
}

}  // namespace chip

using namespace chip;

extern "C" int chip_query_batch_f32(chip_ctx *c, int64_t k, const float *queries, int32_t Q, int32_t topk, float *scores, int64_t *idx)
{
    if (!c || !queries || Q < 1) return CHIP_ERR_INVALID_ARG;
    if (topk < 1 || topk > CHIP_MAX_TOPK || c->D % 32 != 0) return CHIP_ERR_UNSUPPORTED;
    if (c->group || c->elem != 4) return CHIP_ERR_UNSUPPORTED;   // fp32 GEMM over float rows of one device (faiss casts to float, Cerebro.cpp:422)
    int64_t n_global;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        n_global = c->rows_global;
    }
    if (k < 0 || k > n_global) return CHIP_ERR_RANGE;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    if (!c->batch_state) {
        c->batch_state = new (std::nothrow) BatchState();
        if (!c->batch_state) return CHIP_ERR_OOM;
    }
    BatchState *st = static_cast<BatchState *>(c->batch_state);
    const int D = c->D;
    const int Qpad = (Q + BM - 1) / BM * BM;
    const int64_t n_rows = local_count(c, k);
    const int qtiles = Qpad / BM;
    // partitions of the prefix: ~2 workgroups per CU in total, at least one 128-row tile each, at most 512 lists
    int64_t tiles = (n_rows + BN - 1) / BN;
    if (tiles < 1) tiles = 1;
    int64_t P = (2 * (int64_t)c->n_cus + qtiles - 1) / qtiles;
    if (P > tiles) P = tiles;
    if (P > 512) P = 512;
    if (P < 1) P = 1;
    const int64_t rows_per_part = ((tiles + P - 1) / P) * BN;
    P = (n_rows + rows_per_part - 1) / rows_per_part;
    if (P < 1) P = 1;

    hipStream_t s = c->s_scan;
    if ((int64_t)Qpad * D > st->cap_q) {
        (void)hipFree(st->Q); st->Q = nullptr; st->cap_q = 0;
        CHIP_HIP(c, hipMalloc(&st->Q, sizeof(float) * (size_t)Qpad * D));
        st->cap_q = (int64_t)Qpad * D;
    }
    if (P * Qpad * topk > st->cap_partial) {
        (void)hipFree(st->partial); st->partial = nullptr; st->cap_partial = 0;
        CHIP_HIP(c, hipMalloc(&st->partial, sizeof(chip_topk_entry) * (size_t)(P * Qpad * topk)));
        st->cap_partial = P * Qpad * topk;
    }
    if ((int64_t)Qpad * topk > st->cap_out) {
        (void)hipFree(st->out); (void)hipHostFree(st->h_out); st->out = st->h_out = nullptr; st->cap_out = 0;
        CHIP_HIP(c, hipMalloc(&st->out, sizeof(chip_topk_entry) * (size_t)Qpad * topk));
        CHIP_HIP(c, hipHostMalloc(&st->h_out, sizeof(chip_topk_entry) * (size_t)Qpad * topk, hipHostMallocDefault));
        st->cap_out = (int64_t)Qpad * topk;
    }
    CHIP_HIP(c, hipMemsetAsync(st->Q, 0, sizeof(float) * (size_t)Qpad * D, s));
    CHIP_HIP(c, hipMemcpyAsync(st->Q, queries, sizeof(float) * (size_t)Q * D, hipMemcpyHostToDevice, s));

    BatchArgs a;
    a.seg_table = reinterpret_cast<const float *const *>(c->seg_table_dev); a.seg_shift = c->seg_shift; a.seg_mask = c->seg_rows - 1;
    a.n_rows = n_rows; a.D = D; a.Q = st->Q; a.Qpad = Qpad; a.K = topk; a.rows_per_part = rows_per_part;
    a.idx_mul = c->nranks; a.idx_add = c->nranks == 1 ? 0 : c->rank; a.partial = st->partial;
    const int KCsel = 32;   // measured: KC=64 (66 KiB LDS) halves residency and drops 111 -> 88 TFLOP/s at Q=256
    const size_t lds_gemm = sizeof(float) * 2 * BM * (KCsel + 1), lds_ct = sizeof(float) * 64 * CT_LD;
    const size_t lds = lds_gemm > lds_ct ? lds_gemm : lds_ct;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->prof_on) {
        if (c->prof_used + 2 > c->prof_ev.size())
            for (int i = 0; i < 2; i++) { hipEvent_t e; CHIP_HIP(c, hipEventCreate(&e)); c->prof_ev.push_back(e); }
        e0 = c->prof_ev[c->prof_used]; e1 = c->prof_ev[c->prof_used + 1];
        c->prof_used += 2;
        c->prof_bytes_last = (double)n_rows * D * 4.0 * qtiles;
        CHIP_HIP(c, hipEventRecord(e0, s));
    }
    if (KCsel == 64) {
        if (lds > 65536) CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_gemm_topk<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(db_gemm_topk<64>, dim3((unsigned)P, (unsigned)qtiles), dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL(db_gemm_topk<32>, dim3((unsigned)P, (unsigned)qtiles), dim3(256), lds, s, a);
    }
    CHIP_HIP(c, hipGetLastError());
    if (e1) CHIP_HIP(c, hipEventRecord(e1, s));
    BatchMergeArgs m;
    m.in = st->partial; m.n_lists = (int)P; m.Qpad = Qpad; m.K = topk; m.out = st->out;
    hipLaunchKernelGGL(topk_merge_batch, dim3((Q + 3) / 4), dim3(512), 0, s, m);
    CHIP_HIP(c, hipGetLastError());
    CHIP_HIP(c, hipMemcpyAsync(st->h_out, st->out, sizeof(chip_topk_entry) * (size_t)Q * topk, hipMemcpyDeviceToHost, s));
    CHIP_HIP(c, hipStreamSynchronize(s));
    for (int64_t i = 0; i < (int64_t)Q * topk; i++) {
        if (scores) scores[i] = (float)st->h_out[i].score;
        if (idx) idx[i] = st->h_out[i].idx;
    }
    return CHIP_OK;
}
