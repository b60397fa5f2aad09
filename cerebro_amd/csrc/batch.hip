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
// K_B db_gemm_topk<KC, NST, WN, KL> : grid (P workgroups) x (query tiles).  Tile = 64 WN queries x 64 WN DB rows on 2 x WN waves,
//   each wave owning WN x 2 MFMA 32x32 accumulators (64 WN queries x 64 DB rows):
//     WN = 4: 256 x 256 on 8 waves, one workgroup per CU -- 16 MFMAs per 6 fragment reads, one workgroup barrier per 128 MFMAs,
//             the DB streamed once per 256 queries; used when the padded query count is a multiple of 256 (0.84 of the fp32
//             matrix peak at Q = 256 x 1M rows);
//     WN = 2: 128 x 128 on 4 waves, two workgroups per CU (0.78), for everything else.
//   K is streamed in chunks of 32 through a two-stage LDS ring filled by LDS-DMA (see "LDS-DMA tile staging" below).  DB tiles
//   are claimed from a counter (one atomic per ~1 ms tile).  After the K loop of a tile its scores go through the same LDS, 128
//   DB rows per pass, and ALL threads scan them against register-resident sorted top-K lists (thread = query x 64-column half;
//   KL = list capacity, 8 or 16).  One sorted list per workgroup and query is written; topk_merge_batch (the shared
//   merge_sorted_lists) merges them.
#include "chip_internal.h"
#include "topk_merge.h"
#include <cmath>
#include <new>
#include <type_traits>

namespace chip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;              // queries are padded to a multiple of the small tile
constexpr int CT_LD = 128 + 1;      // score tile in LDS [tile queries][129]: 128 DB rows per epilogue pass
// DB tile of one stage in LDS: 16 blocks of 8 rows x 32 floats (one LDS-DMA instruction each: 1 KiB, lane-linear), block b
// placed at b * kBBlock + (b & 3) floats -- rotated by 0..3 floats, which the LDS-DMA honours (its destination base only needs
// 4-byte alignment: scripts/probes/glds_align_probe.hip) -- 4 floats of padding per block keep the rotated blocks apart.
constexpr int kBBlock = 8 * 32 + 4;
constexpr int kMaxQTiles = 4096;    // query tiles per call (grid.y); 4096 x 128 queries
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
    uint32_t *tile_ctr;     // [query tiles] next unclaimed DB tile (zeroed before the launch)
    int64_t idx_mul, idx_add;
    chip_topk_entry *partial;  // [P][Qpad][K]
};

__device__ __forceinline__ bool fkey_gt(float s, int32_t i, float s2, int32_t i2) { return s > s2 || (s == s2 && i > i2); }

template <int KL>
struct TopList {  // sorted (score desc, local row desc); empty slots (-inf, -1); KL = capacity (8 or CHIP_MAX_TOPK): the
    float s[KL];  // lists live in registers across the whole K loop, next to 64..128 accumulator registers
    int32_t r[KL];
};

template <int KL>
__device__ __forceinline__ void list_init(TopList<KL> &L)
{
#pragma unroll
    for (int j = 0; j < KL; j++) { L.s[j] = -INFINITY; L.r[j] = -1; }
}
template <int KL>
__device__ __forceinline__ void list_push(TopList<KL> &L, int K, float s, int32_t row)
{
    // caller checked fkey_gt(s,row, L.s[K-1], L.r[K-1]); static-index insertion (no dynamic register indexing)
#pragma unroll
    for (int j = KL - 1; j >= 1; j--) {
        if (j < K) {
            const bool above = fkey_gt(s, row, L.s[j - 1], L.r[j - 1]);       // new entry belongs above slot j-1
            const bool here = fkey_gt(s, row, L.s[j], L.r[j]) && !above;       // exactly at slot j
            if (above) { L.s[j] = L.s[j - 1]; L.r[j] = L.r[j - 1]; }
            else if (here) { L.s[j] = s; L.r[j] = row; }
        }
    }
    if (fkey_gt(s, row, L.s[0], L.r[0])) { L.s[0] = s; L.r[0] = row; }
}
// the current K-th best of a list (K <= KL, run-time)
template <int KL>
__device__ __forceinline__ void list_kth(const TopList<KL> &L, int K, float &ts, int32_t &tr)
{
    ts = L.s[0]; tr = L.r[0];
#pragma unroll
    for (int j = 1; j < KL; j++)
        if (j < K) { ts = L.s[j]; tr = L.r[j]; }
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

// Qt[qtile][chunk][k][row] <- Q[qtile * TM + row][chunk * KC + k], TM = rows of a query tile (128 or 256)
// (one thread per element; Q is a few MB at most)
__global__ __launch_bounds__(256) void transpose_queries(const float *__restrict__ Q, float *__restrict__ Qt, int Qpad, int D, int tm_shift)
{
    constexpr int KC = 32;
    const int64_t n = (int64_t)Qpad * D;
    const int TM = 1 << tm_shift;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i & (TM - 1));
        const int k = (int)((i >> tm_shift) & (KC - 1));
        const int64_t t = i >> (tm_shift + 5);           // (qtile, chunk) flattened: chunk fastest
        const int chunk = (int)(t % (D / KC)), qt = (int)(t / (D / KC));
        Qt[i] = Q[(int64_t)(qt * TM + row) * D + chunk * KC + k];
    }
}

// WN = waves along the DB rows of a tile (2 or 4); a wave owns WN x 2 MFMA 32x32 blocks (64 WN queries x 64 DB rows), the
// workgroup tile is 64 WN x 64 WN (128 x 128 with 4 waves, 256 x 256 with 8), 2 x WN waves, 8 LDS-DMA per wave per chunk either way.
template <int KC, int NST, int WN, int KL>
__global__ __launch_bounds__(128 * WN) void db_gemm_topk(BatchArgs a)
{
    static_assert(KC == 32, "one K-chunk = 8 slots of 4 floats per DB row");
    static_assert(WN == 2 || WN == 4, "tile 128 x 128 or 256 x 256");
    constexpr int TM = 64 * WN, TN = 64 * WN;            // tile: queries x DB rows
    constexpr int NT = 128 * WN, WAVES = 2 * WN;          // threads, waves
    constexpr int TILE = TM * KC;                         // floats of the A tile of one stage
    constexpr int BTILE = (TN / 8) * kBBlock;             // DB tile: TN / 8 blocks of 8 rows, kBBlock floats apart
    constexpr int STAGE = TILE + BTILE;                   // A tile then B tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S0 = reinterpret_cast<float *>(smem);        // stage s: A = S0 + s*STAGE, B = A + TILE
    float *Ct = reinterpret_cast<float *>(smem);         // [TM][CT_LD] (aliases the stages after the K loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int D = a.D, K = a.K;
    const int q0 = blockIdx.y * TM;
    // DB tiles are handed out dynamically: workgroup x of a query tile starts on DB tile x and then takes the next unclaimed one
    // from a counter (one atomic per tile, ~1 ms of work; fetched at the start of the tile, read at its end).  With a static
    // split 3907 tiles over 256 workgroups cost 16 tile times on every CU; claimed tiles cost 15.26.  The lists a workgroup
    // keeps do not care which tiles fed them: the merged top-K is the same set whatever the assignment.
    __shared__ unsigned next_tile_slot;
    const int64_t n_tiles = (a.n_rows + TN - 1) / TN;

    // Top-k of the tile scores: ALL threads take part -- thread t owns query q0 + (t % TM) over the 64-column half (t / TM)
    // of every 128-column epilogue pass, with its sorted list in registers; the two lists of a query are folded at the end.
    // (Round 1 had 64 owner threads scan 2 x 128 columns while the other three waves -- and, in lockstep, the other workgroup
    // of the CU -- waited.)
    const int oq = tid & (TM - 1), och = tid / TM;
    TopList<KL> L;
    list_init(L);

    const int n_chunks = D / KC;
    // A loader: the chunk image is contiguous (TM * 128 B): pass u, wave w copies bytes [(u * WAVES + w) * 1024, + 1024)
    const float *const a_src = a.Qt + ((int64_t)blockIdx.y * n_chunks) * TILE + (wave * 64 + lane) * 4;
    // B loader: pass u, wave w, lane i -> 8-row block u * WAVES + w, row i >> 3 of it, slot p = i & 7,
    // source floats [4 k4, 4 k4 + 4), k4 = p ^ (row & 7)
    const int l_row = wave * 8 + (lane >> 3), l_p = lane & 7;
    const int l_k4 = (l_p ^ (lane >> 3)) * 4;
    // fragment map
    const int fr = lane & 31, fk = lane >> 5;
    const int rb0 = wn * 64 + fr, rb1 = rb0 + 32;
    const int b_off0 = TILE + b_row_off(rb0) + fk, b_off1 = TILE + b_row_off(rb1) + fk, sw0 = rb0 & 7, sw1 = rb1 & 7;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(S0));

    // Units of work: the first n_full = floor(n_tiles / P) * P tiles are whole tiles (P = workgroups of this query tile: full
    // rounds in which every CU is busy); what remains -- r < P tiles, the round that would leave P - r CUs idle -- is cut
    // into query halves when 2 r <= P (each wave then owns WN / 2 x 2 blocks and the unit takes half a tile time: 15.5 instead of
    // 16 tile times for 3907 tiles on 256 CUs).
    const int64_t n_full = n_tiles / gridDim.x * gridDim.x;
    const bool halves = 2 * (n_tiles - n_full) <= (int64_t)gridDim.x;
    const int64_t n_units = halves ? n_full + 2 * (n_tiles - n_full) : n_tiles;
    for (int64_t unit = blockIdx.x; unit < n_units;) {
        const bool half_unit = halves && unit >= n_full;
        const int64_t tile = half_unit ? n_full + ((unit - n_full) >> 1) : unit;
        const int qhalf = half_unit ? (int)((unit - n_full) & 1) : 0;
        const int nrb = half_unit ? WN / 2 : WN;                                      // 32-query row blocks per wave in this unit
        const int qrow0 = half_unit ? qhalf * (TM / 2) + wm * (16 * WN) : wm * (32 * WN);   // first query row of this wave
        const int a_off = fk * TM + qrow0 + fr;                        // + kk * 2 TM per k-step, + 32 i for row block i
        const int64_t n0 = tile * TN;
        unsigned claimed = 0;
        if (tid == 0) claimed = __hip_atomic_fetch_add(a.tile_ctr + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float *brow[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t br = n0 + u * (8 * WAVES) + l_row;
            const int64_t brc = br < a.n_rows ? br : 0;   // rows past the end: any valid row (their columns are never scanned)
            brow[u] = a.seg_table[brc >> a.seg_shift] + (brc & a.seg_mask) * (int64_t)D + l_k4;
        }
        f32x16 acc[WN][2];
#pragma unroll
        for (int i = 0; i < WN; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

        // chunk c -> stage c % NST: 8 LDS-DMA instructions per wave, number j = 2 u + (0: A, 1: B)
        auto stage_load_one = [&](int c, int j) {
            const uint32_t st = lds_base + (uint32_t)((c % NST) * STAGE) * 4u;
            const int u = j >> 1;
            if (j & 1) glds16(brow[u] + c * KC, st + (uint32_t)(TILE + b_row_off((u * WAVES + wave) * 8)) * 4u);
            else glds16(a_src + (int64_t)c * TILE + u * (WAVES * 256), st + (uint32_t)((u * WAVES + wave) * 256) * 4u);
        };
        auto stage_load = [&](int c) {
#pragma unroll
            for (int j = 0; j < 8; j++) stage_load_one(c, j);
        };
#pragma unroll
        for (int c = 0; c < NST - 1; c++)
            if (c < n_chunks) stage_load(c);
        {
            const int issued = n_chunks < NST - 1 ? n_chunks : NST - 1;
            wait_loads_and_barrier(issued - 1);   // chunk 0 has landed
        }
        // the K loop, instantiated for whole tiles (NRB = WN row blocks per wave) and for query halves (NRB = WN / 2)
        auto k_loop = [&](auto nrb_tag) {
            constexpr int NRB = decltype(nrb_tag)::value;
            for (int c = 0; c < n_chunks; c++) {
                const float *St = S0 + (c % NST) * STAGE;
                // chunk c + NST - 1 goes to the stage last read in iteration c-1 (barrier since).  When its 8 LDS-DMA instructions are
                // issued matters (a burst keeps a wave from issuing MFMAs for ~300 cycles; a late load is waited for at the barrier):
                //   4 stages            one in front of each of the 8 MFMA groups of this iteration (the chunk is needed 3 iterations on);
                //   2 stages, 128 tile  all 8 at once at the top (the other workgroup's wave fills the gap: 119.4 TF vs 114.8 spread);
                //   2 stages, 256 tile  two in front of each of MFMA groups 1..4 (both waves of a SIMD belong to this workgroup and burst
                //                       together: 0.814 of peak vs 0.792 for the burst, 0.805 for groups 0..3, 0.76 for one per group).
                const bool prefetch = c + NST - 1 < n_chunks;
                if (NST == 2 && WN == 2 && prefetch) stage_load(c + NST - 1);
                // fragments of k-steps (2 k4, 2 k4 + 1) in f[k4 & 1]: [2 i + t] = A row block i, [2 WN + 2 j + t] = B block j;
                // the next pair is read before this pair's 4 WN MFMAs are issued
                float f[2][2 * WN + 4];
                auto rd = [&](int k4, float *d) {
#pragma unroll
                    for (int i = 0; i < NRB; i++) {
                        d[2 * i] = St[a_off + 32 * i + (2 * k4) * (2 * TM)];
                        d[2 * i + 1] = St[a_off + 32 * i + (2 * k4 + 1) * (2 * TM)];
                    }
                    const int p0 = (k4 ^ sw0) << 2, p1 = (k4 ^ sw1) << 2;
                    d[2 * WN] = St[b_off0 + p0];     d[2 * WN + 1] = St[b_off0 + p0 + 2];
                    d[2 * WN + 2] = St[b_off1 + p1]; d[2 * WN + 3] = St[b_off1 + p1 + 2];
                };
                rd(0, f[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, NRB + 2, 0);  // the reads of the first pair
#pragma unroll
                for (int k4 = 0; k4 < KC / 4; k4++) {
                    if (NST > 2 && prefetch) stage_load_one(c + NST - 1, k4);
                    if (NST == 2 && WN == 4 && prefetch && k4 >= 1 && k4 <= 4) { stage_load_one(c + 1, 2 * k4 - 2); stage_load_one(c + 1, 2 * k4 - 1); }
                    if (k4 + 1 < KC / 4) rd(k4 + 1, f[(k4 + 1) & 1]);
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const float b0 = f[k4 & 1][2 * WN + t], b1 = f[k4 & 1][2 * WN + 2 + t];
#pragma unroll
                        for (int i = 0; i < NRB; i++) {
                            const float av = f[k4 & 1][2 * i + t];
                            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[i][0], 0, 0, 0);
                            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[i][1], 0, 0, 0);
                        }
                    }
                    // keep the order "reads of the next pair, then this pair's MFMAs" through the scheduler
                    __builtin_amdgcn_sched_group_barrier(0x100, NRB + 2, 0);   // DS reads (ds_read2)
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NRB, 0);   // MFMA
                }
                // chunk c+1 has landed (everything but the NST-2 newest chunks), every wave is done reading stage c % NST
                {
                    int newer = n_chunks - (c + 2);   // chunks after c+1 that have been issued: min(NST - 2, n_chunks - c - 2)
                    if (newer > NST - 2) newer = NST - 2;
                    wait_loads_and_barrier(newer < 0 ? 0 : newer);
                }
            }
        };
        if (half_unit) k_loop(std::integral_constant<int, WN / 2>{});
        else k_loop(std::integral_constant<int, WN>{});
        // ---- epilogue: the score tile through LDS, 128 DB rows (two of the WN wave columns) per pass (the stages are free:
        // the K loop ended with a barrier); C/D layout of the MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int ncols = (a.n_rows - n0) < TN ? (int)(a.n_rows - n0) : TN;
#pragma unroll
        for (int h = 0; h < WN / 2; h++) {
            if (h > 0) __syncthreads();   // the previous pass's scan is done with Ct
            if ((wn >> 1) == h) {
#pragma unroll
                for (int it = 0; it < WN; it++) {
                    if (it >= nrb) continue;   // a query half: the upper row blocks hold nothing
#pragma unroll
                    for (int jt = 0; jt < 2; jt++)
#pragma unroll
                        for (int e = 0; e < 16; e++) {
                            const int row = qrow0 + it * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);   // query within the tile
                            const int col = (wn & 1) * 64 + jt * 32 + (lane & 31);                                  // DB row within the pass
                            Ct[row * CT_LD + col] = acc[it][jt][e];
                        }
                }
            }
            __syncthreads();
            float ts;
            int32_t tr;
            list_kth(L, K, ts, tr);
            const int pass_cols = ncols - h * 128;         // columns of this pass that exist
            const bool mine = !half_unit || (oq >= qhalf * (TM / 2) && oq < (qhalf + 1) * (TM / 2));   // a query half: only its queries have scores
            const int c_hi = !mine ? 0 : pass_cols < (och + 1) * 64 ? pass_cols : (och + 1) * 64;
            for (int c = och * 64; c < c_hi; c++) {
                const float s = Ct[oq * CT_LD + c];
                const int32_t row = (int32_t)(n0 + h * 128 + c);
                if (fkey_gt(s, row, ts, tr)) {   // NaN never enters
                    list_push(L, K, s, row);
                    list_kth(L, K, ts, tr);
                }
            }
        }
        // next tile: published by thread 0, read by everyone after the barrier that also ends this tile's use of Ct
        if (tid == 0) next_tile_slot = gridDim.x + claimed;
        __syncthreads();
        unit = next_tile_slot;
    }
    // fold the two column-half lists of a query into one (through LDS, once per workgroup): one list per (partition, query)
    __syncthreads();
    float *const Ls = reinterpret_cast<float *>(smem);                 // [TM][KL]
    int32_t *const Lr = reinterpret_cast<int32_t *>(Ls + TM * KL);
    if (och == 1) {
#pragma unroll
        for (int j = 0; j < KL; j++) { Ls[oq * KL + j] = L.s[j]; Lr[oq * KL + j] = L.r[j]; }
    }
    __syncthreads();
    if (och == 0) {
        for (int j = 0; j < K; j++) {
            const float s = Ls[oq * KL + j];
            const int32_t row = Lr[oq * KL + j];
            float ts;
            int32_t tr;
            list_kth(L, K, ts, tr);
            if (row >= 0 && fkey_gt(s, row, ts, tr)) list_push(L, K, s, row);
        }
        chip_topk_entry *dst = a.partial + ((int64_t)blockIdx.x * a.Qpad + q0 + oq) * K;
#pragma unroll
        for (int j = 0; j < KL; j++)
            if (j < K) {
                chip_topk_entry e;
                e.score = (double)L.s[j];
                e.idx = L.r[j] >= 0 ? (int64_t)L.r[j] * a.idx_mul + a.idx_add : -1;
                dst[j] = e;
            }
    }
    (void)NT;
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
    uint32_t *tile_ctr = nullptr;     // one counter per query tile (kMaxQTiles)
    chip_topk_entry *partial = nullptr, *out = nullptr, *h_out = nullptr;
    int64_t cap_q = 0, cap_partial = 0, cap_out = 0;
    // sharded DBs (chip_multi.hip): every shard's [Qpad][K] list side by side, and the merged result of the whole DB
    chip_topk_entry *gathered = nullptr, *merged = nullptr;
    int64_t cap_gathered = 0, cap_merged = 0;
    hipEvent_t ev_done = nullptr;     // this shard's list is complete (recorded on its scan stream)
};

void batch_destroy(Ctx *c)
{
    BatchState *st = static_cast<BatchState *>(c->batch_state);
    if (!st) return;
    (void)hipFree(st->Q); (void)hipFree(st->Qt); (void)hipFree(st->tile_ctr); (void)hipFree(st->partial); (void)hipFree(st->out); (void)hipHostFree(st->h_out);
    (void)hipFree(st->gathered); (void)hipFree(st->merged);
    if (st->ev_done) (void)hipEventDestroy(st->ev_done);
    delete st;
    c->batch_state = nullptr;
}

int32_t batch_qpad(int32_t Q) { return (Q + BM - 1) / BM * BM; }

// This ctx's share of a many-query call, enqueued on its scan stream: Q queries (host, Q x D fp32) against its LOCAL rows of the global
// prefix [0, k) -- the whole prefix on a plain ctx, rows i % G == rank of it on a shard -- leaving one sorted [Qpad][topk] list of
// (score, GLOBAL index) entries in device memory (*out_dev).  No host synchronisation.  Caller: query lock held, device current.
int batch_local_enqueue(Ctx *c, int64_t k, const float *queries, int32_t Q, int32_t topk, chip_topk_entry **out_dev, int32_t *Qpad_out)
{
    if (!c->batch_state) {
        c->batch_state = new (std::nothrow) BatchState();
        if (!c->batch_state) return CHIP_ERR_OOM;
    }
    BatchState *st = static_cast<BatchState *>(c->batch_state);
    ResidentPause paused(c, c->tick_resident);   // the many-query scan fills every CU (and may free / allocate): no resident scan instance until it returns
    const int D = c->D;
    const int Qpad = batch_qpad(Q);
    const int64_t n_rows = local_count(c, k);
    // Tile shape: 256 x 256 with 8 waves (one workgroup per CU: 16 MFMAs per 6 fragment reads, one barrier per 128 MFMAs, the DB
    // streamed once per 256 queries) when the padded query count is a multiple of 256, else 128 x 128 with 4 waves.
    // CHIP_BATCH_TILE=128 forces the small tile; CHIP_BATCH_STAGES=4 selects its 4-stage / one-workgroup-per-CU variant.
    const bool wide = Qpad % 256 == 0 && env_int("CHIP_BATCH_TILE", 256) >= 256;
    const int TM = wide ? 256 : BM, TN = TM;
    const int qtiles = Qpad / TM;
    const int nst = !wide && env_int("CHIP_BATCH_STAGES", 2) >= 4 ? 4 : 2;
    const int wgs = wide || nst == 4 ? 1 : 2;            // workgroups per CU
    int64_t tiles = (n_rows + TN - 1) / TN;
    if (tiles < 1) tiles = 1;
    int64_t P = (wgs * (int64_t)c->n_cus + qtiles - 1) / qtiles;   // workgroups per query tile (each ends with one list per query): at most 512
    if (P > tiles) P = tiles;
    if (P > 512) P = 512;
    const int cap_wgs = env_int("CHIP_BATCH_WGS", 0);   // diagnosis / tests: fewer workgroups, i.e. many claimed tiles each
    if (cap_wgs > 0 && P > cap_wgs) P = cap_wgs;
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
    if (!st->tile_ctr) CHIP_HIP(c, hipMalloc(&st->tile_ctr, sizeof(uint32_t) * kMaxQTiles));
    if (qtiles > kMaxQTiles) return CHIP_ERR_UNSUPPORTED;
    CHIP_HIP(c, hipMemsetAsync(st->tile_ctr, 0, sizeof(uint32_t) * kMaxQTiles, s));
    CHIP_HIP(c, hipMemsetAsync(st->Q, 0, sizeof(float) * (size_t)Qpad * D, s));
    CHIP_HIP(c, hipMemcpyAsync(st->Q, queries, sizeof(float) * (size_t)Q * D, hipMemcpyHostToDevice, s));
    {   // K-chunk-major transposed image of the query tiles (what the A-side LDS-DMA copies linearly)
        int64_t g = ((int64_t)Qpad * D + 255) / 256;
        if (g > (int64_t)c->n_cus * 8) g = (int64_t)c->n_cus * 8;
        hipLaunchKernelGGL(transpose_queries, dim3((unsigned)g), dim3(256), 0, s, st->Q, st->Qt, Qpad, D, wide ? 8 : 7);
        CHIP_HIP(c, hipGetLastError());
    }

    BatchArgs a;
    a.seg_table = reinterpret_cast<const float *const *>(c->seg_table_dev); a.seg_shift = c->seg_shift; a.seg_mask = c->seg_rows - 1;
    a.n_rows = n_rows; a.D = D; a.Q = st->Q; a.Qt = st->Qt; a.Qpad = Qpad; a.K = topk; a.tile_ctr = st->tile_ctr;
    a.idx_mul = c->nranks; a.idx_add = c->nranks == 1 ? 0 : c->rank; a.partial = st->partial;
    constexpr int KCsel = 32;
    const size_t lds_gemm = sizeof(float) * (size_t)nst * ((size_t)TM * KCsel + (size_t)(TN / 8) * kBBlock) /* stages x (A + B) */;
    const size_t lds_ct = sizeof(float) * (size_t)TM * CT_LD;
    const size_t lds = lds_gemm > lds_ct ? lds_gemm : lds_ct;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->prof_on) {
        if (c->prof_used + 2 > c->prof_ev.size())
            for (int i = 0; i < 2; i++) { hipEvent_t e; CHIP_HIP(c, hipEventCreate(&e)); c->prof_ev.push_back(e); }
        e0 = c->prof_ev[c->prof_used]; e1 = c->prof_ev[c->prof_used + 1];
        c->prof_used += 2;
        c->prof_bytes_last = (double)n_rows * D * 4.0 * qtiles;   // the DB is streamed once per query tile
        CHIP_HIP(c, hipEventRecord(e0, s));
    }
    auto launch = [&](auto kernel, int threads) -> int {
        CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)P, (unsigned)qtiles), dim3(threads), lds, s, a);
        return CHIP_OK;
    };
    int lrc;
    if (topk <= 8) lrc = wide ? launch(db_gemm_topk<KCsel, 2, 4, 8>, 512) : nst == 4 ? launch(db_gemm_topk<KCsel, 4, 2, 8>, 256) : launch(db_gemm_topk<KCsel, 2, 2, 8>, 256);
    else lrc = wide ? launch(db_gemm_topk<KCsel, 2, 4, CHIP_MAX_TOPK>, 512) : nst == 4 ? launch(db_gemm_topk<KCsel, 4, 2, CHIP_MAX_TOPK>, 256) : launch(db_gemm_topk<KCsel, 2, 2, CHIP_MAX_TOPK>, 256);
    if (lrc != CHIP_OK) return lrc;
    CHIP_HIP(c, hipGetLastError());
    if (e1) CHIP_HIP(c, hipEventRecord(e1, s));
    BatchMergeArgs m;
    m.in = st->partial; m.n_lists = (int)P; m.Qpad = Qpad; m.K = topk; m.out = st->out;
    hipLaunchKernelGGL(topk_merge_batch, dim3((Q + 3) / 4), dim3(512), 0, s, m);
    CHIP_HIP(c, hipGetLastError());
    *out_dev = st->out;
    *Qpad_out = Qpad;
    return CHIP_OK;
}

// D2H of a finished [Q][topk] list on the ctx's scan stream + conversion to the caller's arrays (synchronises that stream)
int batch_deliver(Ctx *c, const chip_topk_entry *list_dev, int32_t Q, int32_t topk, float *scores, int64_t *idx)
{
    BatchState *st = static_cast<BatchState *>(c->batch_state);
    hipStream_t s = c->s_scan;
    CHIP_HIP(c, hipMemcpyAsync(st->h_out, list_dev, sizeof(chip_topk_entry) * (size_t)Q * topk, hipMemcpyDeviceToHost, s));
    CHIP_HIP(c, hipStreamSynchronize(s));
    if (st->h_out[0].idx == kFailedShardIdx) return CHIP_ERR_SHARD_FAILED;   // a shard took part with the failure mark (chip_multi.hip)
    for (int64_t i = 0; i < (int64_t)Q * topk; i++) {
        if (scores) scores[i] = (float)st->h_out[i].score;
        if (idx) idx[i] = st->h_out[i].idx;
    }
    return CHIP_OK;
}

// buffers of the cross-shard merge on the ctx that merges: gathered [n_lists][Qpad][topk], merged [Qpad][topk]
int batch_exchange_buffers(Ctx *c, int n_lists, int32_t Qpad, int32_t topk, chip_topk_entry **gathered, chip_topk_entry **merged, hipEvent_t *ev_done)
{
    if (!c->batch_state) {
        c->batch_state = new (std::nothrow) BatchState();
        if (!c->batch_state) return CHIP_ERR_OOM;
    }
    BatchState *st = static_cast<BatchState *>(c->batch_state);
    const int64_t ng = (int64_t)n_lists * Qpad * topk, nm = (int64_t)Qpad * topk;
    if (ng > st->cap_gathered) {
        (void)hipFree(st->gathered); st->gathered = nullptr; st->cap_gathered = 0;
        CHIP_HIP(c, hipMalloc(&st->gathered, sizeof(chip_topk_entry) * (size_t)ng));
        st->cap_gathered = ng;
    }
    if (nm > st->cap_merged) {
        (void)hipFree(st->merged); st->merged = nullptr; st->cap_merged = 0;
        CHIP_HIP(c, hipMalloc(&st->merged, sizeof(chip_topk_entry) * (size_t)nm));
        st->cap_merged = nm;
    }
    if (!st->ev_done) CHIP_HIP(c, hipEventCreateWithFlags(&st->ev_done, hipEventDisableTiming));
    if (gathered) *gathered = st->gathered;
    if (merged) *merged = st->merged;
    if (ev_done) *ev_done = st->ev_done;
    return CHIP_OK;
}

// merge of n_lists per-shard lists [n_lists][Qpad][topk] -> out [Qpad][topk] on stream s (the same exact selection as within a device)
int batch_merge_lists(Ctx *c, hipStream_t s, const chip_topk_entry *in, int n_lists, int32_t Qpad, int32_t Q, int32_t topk, chip_topk_entry *out)
{
    BatchMergeArgs m;
    m.in = in; m.n_lists = n_lists; m.Qpad = Qpad; m.K = topk; m.out = out;
    hipLaunchKernelGGL(topk_merge_batch, dim3((Q + 3) / 4), dim3(512), 0, s, m);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

}  // namespace chip

using namespace chip;

extern "C" int chip_query_batch_f32(chip_ctx *c, int64_t k, const float *queries, int32_t Q, int32_t topk, float *scores, int64_t *idx)
{
    if (!c || !queries || Q < 1) return CHIP_ERR_INVALID_ARG;
    if (topk < 1 || topk > CHIP_MAX_TOPK || c->D % 32 != 0) return CHIP_ERR_UNSUPPORTED;
    if ((int64_t)(Q + BM - 1) / BM > kMaxQTiles) return CHIP_ERR_UNSUPPORTED;
    if (c->group) return group_query_batch(c, k, queries, Q, topk, scores, idx);   // G per-device passes -> one merge on devices[0]
    if (c->elem != 4) return CHIP_ERR_UNSUPPORTED;   // fp32 GEMM over float rows (faiss casts to float, Cerebro.cpp:422)
    int64_t n_global;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        n_global = c->rows_global;
    }
    if (k < 0) return CHIP_ERR_RANGE;
    const bool collective = c->xchg != nullptr;      // sharded ctx with an in-library exchange: every rank makes the same call
    if (k > n_global && !collective) return CHIP_ERR_RANGE;
    std::lock_guard<std::mutex> qlk(c->query_mu);
    CHIP_HIP(c, hipSetDevice(c->device));
    if (collective) return xchg_query_batch(c, k, queries, Q, topk, scores, idx, k > n_global);
    chip_topk_entry *out = nullptr;
    int32_t Qpad = 0;
    const int rc = batch_local_enqueue(c, k, queries, Q, topk, &out, &Qpad);
    if (rc != CHIP_OK) return rc;
    return batch_deliver(c, out, Q, topk, scores, idx);
}
