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
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
constexpr int CT_LD = BN + 1;   // score tile in LDS [128][129]
// DB tile of one stage in LDS: 16 blocks of 8 rows x 32 floats (one LDS-DMA instruction each: 1 KiB, lane-linear), block b
// placed at b * kBBlock + (b & 3) floats -- rotated by 0..3 floats, which the LDS-DMA honours (its destination base only needs
// 4-byte alignment: scripts/probes/glds_align_probe.hip) -- 4 floats of padding per block keep the rotated blocks apart.
constexpr int kBBlock = 8 * 32 + 4;
constexpr int kBTile = 16 * kBBlock;
__device__ __forceinline__ constexpr int b_row_off(int row) { return (row >> 3) * kBBlock + ((row >> 3) & 3) + (row & 7) * 32; }

struct BatchArgs {
    const float *const *seg_table;
    int32_t seg_shift;
    int64_t seg_mask;
    int64_t n_rows;         // local rows [0, n_rows)
    int32_t D;
    const float *Q;         // [Qpad][D] device, rows >= Q are zero
    const float *Qt;        // the same queries as K-chunk-major transposed tiles: [Qpad/128][D/32][32][128]
    int32_t Qpad;
    int32_t K;
    int64_t rows_per_part;  // multiple of BN
    int64_t idx_mul, idx_add;
    chip_topk_entry *partial;  // [P][Qpad][K]
};

__device__ __forceinline__ bool fkey_gt(float s, int32_t i, float s2, int32_t i2) { return s > s2 || (s == s2 && i > i2); }

struct TopList {  // sorted (score desc, local row desc); empty slots (-inf, -1)
    float s[CHIP_MAX_TOPK];
    int32_t r[CHIP_MAX_TOPK];
};

__device__ __forceinline__ void list_init(TopList &L)
{
#pragma unroll
    for (int j = 0; j < CHIP_MAX_TOPK; j++) { L.s[j] = -INFINITY; L.r[j] = -1; }
}
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

// LDS-DMA tile staging (cdna_hip_programming.md 3, rule 21): global_load_lds_dwordx4 copies 16 bytes per lane straight from
// global memory to LDS at (wave-uniform base + lane * 16) -- no staging VGPRs, no ds_write pass.  The MFMA loop must also be
// free of VALU work: measured on this chip (scripts/probes/mfma_probe.hip, one wave per SIMD) a v_mfma_f32_32x32x2_f32 stream
// runs at 0.96 of peak with its fragments read as plain ds_read_b32, but at 0.80 when every operand needs one v_cndmask (16-B
// fragment reads + select) -- VALU instructions do not hide under this MFMA.  So both LDS images are laid out such that a lane
// reads exactly the floats it feeds to the matrix core:
//   A (queries)  : the host side keeps a K-CHUNK-MAJOR, TRANSPOSED image of the query tile in global memory
//                  (Qt[qtile][chunk][k 0..31][row 0..127], written once per call by transpose_queries), so the DMA is a linear
//                  16 KiB copy per chunk and lane (fr, fk) reads A[R + fr][2 kk + fk] at float (2 kk + fk) * 128 + R + fr:
//                  32 consecutive floats per half-wave -- conflict-free ds_read_b32, two k-steps per ds_read2st64_b32.
//   B (DB rows)  : row-major in global memory (fixed): per stage 16 blocks of [8 rows][8 slots of 16 B], one LDS-DMA
//                  instruction each (the destination is lane-linear), slots XOR-swizzled on both sides (slot p of row r holds
//                  floats [4 k4, 4 k4 + 4) with k4 = p ^ (r & 7): the loader picks its SOURCE address accordingly) and block
//                  b ROTATED by b & 3 floats (b_row_off).  Lane (fr, fk) reads floats [fk] and [2 + fk] of its row's slot with
//                  ONE ds_read2_b32; the 32 rows of a half-wave are 4 blocks x 8 rows = 4 rotations x 8 swizzled slots = 32
//                  distinct banks.  (Without the rotation a column read over 128-B rows is 4-way bank-conflicted whatever the
//                  slot swizzle -- 32 rows, 8 slots, one float offset: round 2 ran that way until the probe showed the
//                  destination base of an LDS-DMA may be any multiple of 4 bytes.)
// The LDS-DMA is issued from inline asm ON PURPOSE: hipcc tracks the builtin form and, not knowing which stage a load fills,
// drains ALL of them (s_waitcnt vmcnt(0)) at every barrier.  Untracked, the loads of the next NST-1 chunks stay in flight
// across barriers and the loop waits with counted s_waitcnt vmcnt(8 (NST-2)): vmcnt retires in order, so "at most n
// outstanding" == "everything but the newest n loads has landed".
// M0 (the wave-uniform LDS destination base) is written in the same statement that uses it (cdna_hip_programming.md, asm notes).
__device__ __forceinline__ void glds16(const float *gsrc, uint32_t lds_byte_addr_wave_uniform)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr_wave_uniform)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// at most 8 * CHUNKS LDS-DMA loads still in flight, own LDS reads done, workgroup barrier
__device__ __forceinline__ void wait_loads_and_barrier(int chunks_in_flight)
{
    if (chunks_in_flight >= 2) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (chunks_in_flight == 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Qt[qtile][chunk][k][row] <- Q[qtile * 128 + row][chunk * KC + k]   (one thread per element; Q is a few MB at most)
__global__ __launch_bounds__(256) void transpose_queries(const float *__restrict__ Q, float *__restrict__ Qt, int Qpad, int D)
{
    constexpr int KC = 32;
    const int64_t n = (int64_t)Qpad * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i & 127);
        const int k = (int)((i >> 7) & (KC - 1));
        const int64_t t = i >> 12;                       // (qtile, chunk) flattened: chunk fastest
        const int chunk = (int)(t % (D / KC)), qt = (int)(t / (D / KC));
        Qt[i] = Q[(int64_t)(qt * 128 + row) * D + chunk * KC + k];
    }
}

template <int KC, int NST>
__global__ __launch_bounds__(256) void db_gemm_topk(BatchArgs a)
{
    static_assert(KC == 32, "one K-chunk = 8 slots of 4 floats per DB row");
    constexpr int TILE = BM * KC;        // floats of the A tile of one stage (16 KiB)
    constexpr int STAGE = TILE + kBTile; // A tile then B tile (16 blocks of 8 rows, kBBlock floats apart)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S0 = reinterpret_cast<float *>(smem);        // stage s: A = S0 + s*STAGE, B = A + TILE
    float *Ct = reinterpret_cast<float *>(smem);         // [128][CT_LD] (aliases the stages after the K loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int D = a.D, K = a.K;
    const int q0 = blockIdx.y * BM;
    const int64_t part_lo = (int64_t)blockIdx.x * a.rows_per_part;
    int64_t part_hi = part_lo + a.rows_per_part;
    if (part_hi > a.n_rows) part_hi = a.n_rows;

    // Top-k of the tile scores: ALL 256 threads take part -- thread t owns query q0 + (t & 127) over the column half (t >> 7)
    // of every tile, with its sorted list in registers; the two lists of a query are folded into one at the end of the kernel.
    // (Round 1 had 64 owner threads scan 2 x 128 columns while the other three waves -- and, in lockstep, the other workgroup
    // of the CU -- waited.)
    const int oq = tid & 127, och = tid >> 7;
    TopList L;
    list_init(L);

    const int n_chunks = D / KC;
    // A loader: the chunk image is contiguous (16 KiB): wave w, pass u copies bytes [(4 u + w) * 1024, + 1024)
    const float *const a_src = a.Qt + ((int64_t)blockIdx.y * n_chunks) * TILE + (wave * 64 + lane) * 4;
    // B loader: pass u, wave w, lane i -> row u*32 + w*8 + (i >> 3) (= row i >> 3 of 8-row block 4 u + w), slot p = i & 7,
    // source floats [4 k4, 4 k4 + 4), k4 = p ^ (row & 7)
    const int l_row = wave * 8 + (lane >> 3), l_p = lane & 7;
    const int l_k4 = (l_p ^ (lane >> 3)) * 4;
    // fragment map
    const int fr = lane & 31, fk = lane >> 5;
    const int a_off = fk * 128 + wm * 64 + fr;                        // + (2 kk) * 128 per k-step, + 32 for the second row block
    const int rb0 = wn * 64 + fr, rb1 = rb0 + 32;
    const int b_off0 = TILE + b_row_off(rb0) + fk, b_off1 = TILE + b_row_off(rb1) + fk, sw0 = rb0 & 7, sw1 = rb1 & 7;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(S0));

    for (int64_t n0 = part_lo; n0 < part_hi; n0 += BN) {
        const float *brow[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t br = n0 + u * 32 + l_row;
            const int64_t brc = br < a.n_rows ? br : 0;   // rows past the end: any valid row (their columns are never scanned)
            brow[u] = a.seg_table[brc >> a.seg_shift] + (brc & a.seg_mask) * (int64_t)D + l_k4;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

        // chunk c -> stage c % NST: 8 LDS-DMA instructions per wave, number j = 2 u + (0: A, 1: B)
        auto stage_load_one = [&](int c, int j) {
            const uint32_t st = lds_base + (uint32_t)((c % NST) * STAGE) * 4u;
            const int u = j >> 1;
            if (j & 1) glds16(brow[u] + c * KC, st + (uint32_t)(TILE + b_row_off(u * 32 + wave * 8)) * 4u);
            else glds16(a_src + (int64_t)c * TILE + u * 1024, st + (uint32_t)((4 * u + wave) * 256) * 4u);
        };
        auto stage_load = [&](int c) {
#pragma unroll
            for (int j = 0; j < 8; j++) stage_load_one(c, j);
        };
        __syncthreads();              // the previous tile's epilogue no longer reads Ct (aliases the stages)
#pragma unroll
        for (int c = 0; c < NST - 1; c++)
            if (c < n_chunks) stage_load(c);
        {
            const int issued = n_chunks < NST - 1 ? n_chunks : NST - 1;
            wait_loads_and_barrier(issued - 1);   // chunk 0 has landed
        }
        for (int c = 0; c < n_chunks; c++) {
            const float *St = S0 + (c % NST) * STAGE;
            // chunk c + NST - 1 goes to the stage last read in iteration c-1 (barrier since).  With 4 stages its 8 LDS-DMA
            // instructions are spread over the 8 MFMA groups below (issued in one burst they keep the lone wave of a SIMD from
            // issuing MFMAs for ~300 cycles per chunk); with 2 stages they must go out at once -- the chunk is needed at the end
            // of this iteration, and the other workgroup's wave fills the issue gap (measured: 119.4 burst vs 114.8 spread).
            const bool prefetch = c + NST - 1 < n_chunks;
            if (NST == 2 && prefetch) stage_load(c + NST - 1);
            // fragments of k-steps (2 k4, 2 k4 + 1) in f[k4 & 1]; the next pair is read before this pair's 8 MFMAs are issued
            float f[2][8];
            auto rd = [&](int k4, float *d) {
                d[0] = St[a_off + (2 * k4) * 256];      d[1] = St[a_off + (2 * k4 + 1) * 256];
                d[2] = St[a_off + 32 + (2 * k4) * 256]; d[3] = St[a_off + 32 + (2 * k4 + 1) * 256];
                const int p0 = (k4 ^ sw0) << 2, p1 = (k4 ^ sw1) << 2;
                d[4] = St[b_off0 + p0]; d[5] = St[b_off0 + p0 + 2];
                d[6] = St[b_off1 + p1]; d[7] = St[b_off1 + p1 + 2];
            };
            rd(0, f[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);       // the 4 reads of the first pair
#pragma unroll
            for (int k4 = 0; k4 < KC / 4; k4++) {
                if (NST > 2 && prefetch) stage_load_one(c + NST - 1, k4);
                if (k4 + 1 < KC / 4) rd(k4 + 1, f[(k4 + 1) & 1]);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const float a0 = f[k4 & 1][t], a1 = f[k4 & 1][2 + t], b0 = f[k4 & 1][4 + t], b1 = f[k4 & 1][6 + t];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
                // keep the order "reads of the next pair, then this pair's 8 MFMAs" through the scheduler
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // 8 MFMA
            }
            // chunk c+1 has landed (everything but the NST-2 newest chunks), every wave is done reading stage c % NST
            {
                int newer = n_chunks - (c + 2);   // chunks after c+1 that have been issued: min(NST - 2, n_chunks - c - 2)
                if (newer > NST - 2) newer = NST - 2;
                wait_loads_and_barrier(newer < 0 ? 0 : newer);
            }
        }
        // ---- epilogue: the 128 x 128 score tile through LDS (the two stages are free: the K loop ended with a barrier);
        // C/D layout of the MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int ncols = (part_hi - n0) < BN ? (int)(part_hi - n0) : BN;
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = wm * 64 + it * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);   // query within the tile
                    const int col = wn * 64 + jt * 32 + (lane & 31);                                 // DB row within the tile
                    Ct[row * CT_LD + col] = acc[it][jt][e];
                }
        __syncthreads();
        {
            float ts = L.s[0];
            int32_t tr = L.r[0];
#pragma unroll
            for (int j = 1; j < CHIP_MAX_TOPK; j++)
                if (j < K) { ts = L.s[j]; tr = L.r[j]; }   // current K-th best
            const int c_hi = ncols < (och + 1) * 64 ? ncols : (och + 1) * 64;
            for (int c = och * 64; c < c_hi; c++) {
                const float s = Ct[oq * CT_LD + c];
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
    // fold the two column-half lists of a query into one (through LDS, once per workgroup): one list per (partition, query)
    __syncthreads();
    float *const Ls = reinterpret_cast<float *>(smem);                 // [128][CHIP_MAX_TOPK]
    int32_t *const Lr = reinterpret_cast<int32_t *>(Ls + 128 * CHIP_MAX_TOPK);
    if (och == 1) {
#pragma unroll
        for (int j = 0; j < CHIP_MAX_TOPK; j++) { Ls[oq * CHIP_MAX_TOPK + j] = L.s[j]; Lr[oq * CHIP_MAX_TOPK + j] = L.r[j]; }
    }
    __syncthreads();
    if (och == 0) {
        for (int j = 0; j < K; j++) {
            const float s = Ls[oq * CHIP_MAX_TOPK + j];
            const int32_t row = Lr[oq * CHIP_MAX_TOPK + j];
            float ts = L.s[0];
            int32_t tr = L.r[0];
#pragma unroll
            for (int jj = 1; jj < CHIP_MAX_TOPK; jj++)
                if (jj < K) { ts = L.s[jj]; tr = L.r[jj]; }
            if (row >= 0 && fkey_gt(s, row, ts, tr)) list_push(L, K, s, row);
        }
        chip_topk_entry *dst = a.partial + ((int64_t)blockIdx.x * a.Qpad + q0 + oq) * K;
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
    float *Q = nullptr, *Qt = nullptr;
    chip_topk_entry *partial = nullptr, *out = nullptr, *h_out = nullptr;
    int64_t cap_q = 0, cap_partial = 0, cap_out = 0;
};

void batch_destroy(Ctx *c)
{
    BatchState *st = static_cast<BatchState *>(c->batch_state);
    if (!st) return;
    (void)hipFree(st->Q); (void)hipFree(st->Qt); (void)hipFree(st->partial); (void)hipFree(st->out); (void)hipHostFree(st->h_out);
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
    // partitions of the prefix: `wgs` workgroups per CU in total, at least one 128-row tile each, at most 512 lists.
    // Stages of the LDS-DMA pipeline: 2 (64 KiB per workgroup, two workgroups per CU) or 4 (128 KiB, one per CU); CHIP_BATCH_STAGES
    // is a tuning knob, the default is the measured best.
    const int nst = env_int("CHIP_BATCH_STAGES", 2) >= 4 ? 4 : 2;
    const int wgs = nst == 4 ? 1 : 2;
    int64_t tiles = (n_rows + BN - 1) / BN;
    if (tiles < 1) tiles = 1;
    int64_t P = (wgs * (int64_t)c->n_cus + qtiles - 1) / qtiles;
    if (P > tiles) P = tiles;
    if (P > 512) P = 512;
    if (P < 1) P = 1;
    const int64_t rows_per_part = ((tiles + P - 1) / P) * BN;
    P = (n_rows + rows_per_part - 1) / rows_per_part;
    if (P < 1) P = 1;

    hipStream_t s = c->s_scan;
    if ((int64_t)Qpad * D > st->cap_q) {
        (void)hipFree(st->Q); st->Q = nullptr; st->cap_q = 0;
        (void)hipFree(st->Qt); st->Qt = nullptr;
        CHIP_HIP(c, hipMalloc(&st->Q, sizeof(float) * (size_t)Qpad * D));
        CHIP_HIP(c, hipMalloc(&st->Qt, sizeof(float) * (size_t)Qpad * D));
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
    {   // K-chunk-major transposed image of the query tiles (what the A-side LDS-DMA copies linearly)
        int64_t g = ((int64_t)Qpad * D + 255) / 256;
        if (g > (int64_t)c->n_cus * 8) g = (int64_t)c->n_cus * 8;
        hipLaunchKernelGGL(transpose_queries, dim3((unsigned)g), dim3(256), 0, s, st->Q, st->Qt, Qpad, D);
        CHIP_HIP(c, hipGetLastError());
    }

    BatchArgs a;
    a.seg_table = reinterpret_cast<const float *const *>(c->seg_table_dev); a.seg_shift = c->seg_shift; a.seg_mask = c->seg_rows - 1;
    a.n_rows = n_rows; a.D = D; a.Q = st->Q; a.Qt = st->Qt; a.Qpad = Qpad; a.K = topk; a.rows_per_part = rows_per_part;
    a.idx_mul = c->nranks; a.idx_add = c->nranks == 1 ? 0 : c->rank; a.partial = st->partial;
    constexpr int KCsel = 32;
    const size_t lds_gemm = sizeof(float) * (size_t)nst * (BM * KCsel + kBTile) /* stages x (A + B) */, lds_ct = sizeof(float) * BM * CT_LD;
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
    if (nst == 4) {
        CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_gemm_topk<KCsel, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((db_gemm_topk<KCsel, 4>), dim3((unsigned)P, (unsigned)qtiles), dim3(256), lds, s, a);
    } else {
        CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_gemm_topk<KCsel, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((db_gemm_topk<KCsel, 2>), dim3((unsigned)P, (unsigned)qtiles), dim3(256), lds, s, a);
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
