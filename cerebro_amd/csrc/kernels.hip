// kernels.hip -- hand-written gfx950 kernels for the descriptor dot-product scan path.
//
//   K1 db_scan_topk   : replaces  u = v^T * M.leftCols(k) (x3) + maxCoeff + last-index argmax
//                       (/root/reference/src/Cerebro.cpp:1026-1043).  HBM-bound streaming kernel:
//                       one DB pass for all queries, fp64 accumulation of exact fp32 products in the
//                       fixed order of DESIGN.md 3, per-wave register top-K, per-block LDS merge.
//   K2 topk_merge     : merges per-block (or per-GPU) lists and applies the accept rule Cerebro.cpp:1056.
//   K3 narrow/copy    : M.col(_s) = desc (Cerebro.cpp:1005-1006): f64 -> f32 with a losslessness check.
//   synth             : on-device synthetic DB (spec = oracle/dot_scan.c).
//
// CDNA4 notes: wave = 64; queries live in LDS (NQ*D*4 B) and are read with conflict-free ds_read_b128;
// DB rows are streamed with 16-B-per-lane non-temporal global loads (1 KiB per wave-instruction, U in flight
// per lane); the wave id is readfirstlane'd so row bases stay in SGPRs.  No MFMA: AI = NQ/2 flop/B.
#include "chip_internal.h"
#include "topk_merge.h"
#include <cmath>

namespace chip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------ K1
// scan -> per-wave register top-K -> per-block LDS merge -> one sorted partial list per workgroup in global memory.
// The cross-workgroup merge (+ accept decision) is K2 on the ctx stream, behind an event, so that it overlaps the
// NEXT tick's scan (a fused last-workgroup merge was measured to serialise ~28 us per tick).
// One 16-byte streaming load of a DB row chunk.  POLICY 1 (production) = the non-temporal hint; 0 = plain; 2..5 = other
// gfx950 cache-policy bit combinations, reachable only through CHIP_SCAN_VARIANT in tuning builds.
template <int POLICY>
__device__ __forceinline__ f32x4 stream_load(const f32x4 *p)
{
    if constexpr (POLICY == 0) return *p;
    else if constexpr (POLICY == 1) return __builtin_nontemporal_load(p);
    else {
        f32x4 v;
        if constexpr (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
        else if constexpr (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
        else if constexpr (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
}

template <int NQ, int U, bool FULL, int NT, int R>
__global__ __launch_bounds__(1024) void db_scan_topk(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *qs = reinterpret_cast<float *>(smem);  // [NQ][D]
    const int D = a.D;
    const int K = a.K;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpb = blockDim.x >> 6;

    // stage the query descriptors once per block (L2-resident after the first block)
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const float *src = a.q[q];
        for (int e = tid * 4; e < D; e += blockDim.x * 4)
            *reinterpret_cast<f32x4 *>(qs + q * D + e) = *reinterpret_cast<const f32x4 *>(src + e);
    }
    __syncthreads();

    // per-wave running top-K: lane j < K holds the j-th best (score desc, index desc)
    double my_s[NQ], thr_s[NQ];
    int64_t my_i[NQ], thr_i[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { my_s[q] = -INFINITY; my_i[q] = -1; thr_s[q] = -INFINITY; thr_i[q] = -1; }

    const int64_t tw = (int64_t)gridDim.x * wpb;
    const int e0 = lane * 4;
    // each wave owns rows w, w+tw, ... ; R of them are in flight together (R independent accumulator sets)
    for (int64_t r0 = (int64_t)blockIdx.x * wpb + wave; r0 < a.n_rows; r0 += tw * R) {
        const float *row[R];
        double acc[R][NQ];
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const int64_t r = r0 + rr * tw;
            const int64_t rc = r < a.n_rows ? r : r0;   // clamp (result of a clamped row is discarded)
            row[rr] = a.seg_table[rc >> a.seg_shift] + (rc & a.seg_mask) * (int64_t)D;
#pragma unroll
            for (int q = 0; q < NQ; q++) acc[rr][q] = 0.0;
        }
        for (int base = 0; base < D; base += 256 * U) {
            f32x4 v[R][U];
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int e = base + u * 256 + e0;
                    if (FULL || e < D) {
                        const f32x4 *p = reinterpret_cast<const f32x4 *>(row[rr] + e);
                        v[rr][u] = stream_load<NT>(p);
                    }
                }
            }
            if constexpr (NT >= 2) {   // inline-asm loads: the compiler does not track them
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int rr = 0; rr < R; rr++)
#pragma unroll
                    for (int u = 0; u < U; u++) asm volatile("" : "+v"(v[rr][u]));
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = base + u * 256 + e0;
                if (FULL || e < D) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(qs + q * D + e);
                        const double w0 = (double)w.x, w1 = (double)w.y, w2 = (double)w.z, w3 = (double)w.w;
#pragma unroll
                        for (int rr = 0; rr < R; rr++) {
                            // exact products, one rounding per add: fma == mul-then-add here
                            acc[rr][q] = __builtin_fma(w0, (double)v[rr][u].x, acc[rr][q]);
                            acc[rr][q] = __builtin_fma(w1, (double)v[rr][u].y, acc[rr][q]);
                            acc[rr][q] = __builtin_fma(w2, (double)v[rr][u].z, acc[rr][q]);
                            acc[rr][q] = __builtin_fma(w3, (double)v[rr][u].w, acc[rr][q]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const int64_t r = r0 + rr * tw;
            if (r >= a.n_rows) break;   // wave-uniform
            // fixed butterfly: acc[L] += acc[L ^ m], m = 32..1  (every lane ends with the same bits)
#pragma unroll
            for (int q = 0; q < NQ; q++) {
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) acc[rr][q] = acc[rr][q] + __shfl_xor(acc[rr][q], m, 64);
            }
            const int64_t gi = r * a.idx_mul + a.idx_add;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const double s = acc[rr][q];
                if (key_gt(s, gi, thr_s[q], thr_i[q])) {  // wave-uniform, rare after warm-up; NaN never enters
                    const bool worse = key_gt(s, gi, my_s[q], my_i[q]);
                    const unsigned long long m = __ballot(worse) & ((1ull << K) - 1ull);
                    const int pos = __builtin_ctzll(m);
                    const double up_s = __shfl_up(my_s[q], 1, 64);
                    const int64_t up_i = __shfl_up(my_i[q], 1, 64);
                    if (lane < K) {
                        if (lane > pos) { my_s[q] = up_s; my_i[q] = up_i; }
                        else if (lane == pos) { my_s[q] = s; my_i[q] = gi; }
                    }
                    thr_s[q] = __shfl(my_s[q], K - 1, 64);
                    thr_i[q] = __shfl(my_i[q], K - 1, 64);
                }
            }
        }
    }

    // ---- block merge: wpb sorted lists of K per query -> one sorted list of K, by waves 0..NQ-1 ----
    __syncthreads();  // all waves done with qs; reuse LDS
    chip_topk_entry *cand = reinterpret_cast<chip_topk_entry *>(smem);  // [wpb][NQ][K]
    if (lane < K) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            chip_topk_entry t;
            t.score = my_s[q];
            t.idx = my_i[q];
            cand[(wave * NQ + q) * K + lane] = t;
        }
    }
    __syncthreads();
    for (int q = wave; q < NQ; q += wpb) {
        const int ncand = wpb * K;  // <= 16 waves * 16 = 256 -> at most 4 per lane
        double cs[4];
        int64_t ci[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int c = lane + 64 * h;
            if (c < ncand) {
                const chip_topk_entry t = cand[((c / K) * NQ + q) * K + (c % K)];
                cs[h] = t.score;
                ci[h] = t.idx;
            } else { cs[h] = -INFINITY; ci[h] = -1; }
        }
        chip_topk_entry *outp = a.partial + ((int64_t)blockIdx.x * NQ + q) * K;
        for (int j = 0; j < K; j++) {
            double bs = cs[0];
            int64_t bi = ci[0];
#pragma unroll
            for (int h = 1; h < 4; h++)
                if (key_gt(cs[h], ci[h], bs, bi)) { bs = cs[h]; bi = ci[h]; }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const double os = __shfl_xor(bs, m, 64);
                const int64_t oi = __shfl_xor(bi, m, 64);
                if (key_gt(os, oi, bs, bi)) { bs = os; bi = oi; }
            }
#pragma unroll
            for (int h = 0; h < 4; h++)
                if (ci[h] == bi && cs[h] == bs) { cs[h] = -INFINITY; ci[h] = -1; }
            if (lane == 0) { chip_topk_entry t; t.score = bs; t.idx = bi; outp[j] = t; }
        }
    }
}

template <int NQ, int U, bool FULL, int NT, int R>
static int launch_scan_k(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
    if (lds > 65536) CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_topk<NQ, U, FULL, NT, R>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((db_scan_topk<NQ, U, FULL, NT, R>), dim3(grid), dim3(block), lds, s, a);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

template <int NQ, int U, int NT, int R>
static int launch_scan_t(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
    return a.D % (256 * U) == 0 ? launch_scan_k<NQ, U, true, NT, R>(c, s, a, grid, lds, block)
                                : launch_scan_k<NQ, U, false, NT, R>(c, s, a, grid, lds, block);
}

// scan_variant (CHIP_SCAN_VARIANT, tuning/A-B only): 0 = production (U=8, non-temporal loads, 1 row in flight per wave)
template <int NQ>
static int launch_scan_q(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
#ifdef CHIP_SCAN_TUNING_VARIANTS
    switch (c->scan_variant) {
        case 1: return launch_scan_t<NQ, 4, 1, 1>(c, s, a, grid, lds, block);
        case 2: return launch_scan_t<NQ, 16, 1, 1>(c, s, a, grid, lds, block);
        case 3: return launch_scan_t<NQ, 8, 0, 1>(c, s, a, grid, lds, block);
        case 4: return launch_scan_t<NQ, 4, 1, 2>(c, s, a, grid, lds, block);
        case 5: return launch_scan_t<NQ, 8, 1, 2>(c, s, a, grid, lds, block);
        case 6: return launch_scan_t<NQ, 4, 0, 2>(c, s, a, grid, lds, block);
        case 7: return launch_scan_t<NQ, 8, 2, 1>(c, s, a, grid, lds, block);
        case 8: return launch_scan_t<NQ, 8, 3, 1>(c, s, a, grid, lds, block);
        case 9: return launch_scan_t<NQ, 8, 4, 1>(c, s, a, grid, lds, block);
        case 10: return launch_scan_t<NQ, 8, 5, 1>(c, s, a, grid, lds, block);
        default: break;
    }
#endif
    return launch_scan_t<NQ, 8, 1, 1>(c, s, a, grid, lds, block);
}

// Workgroup shape of K1.  The nq query descriptors sit in LDS (nq*D*4 bytes per workgroup), so the shape follows D:
// 2 workgroups x 512 threads per CU while two copies fit in the 160 KiB (D = 4096: 48 KiB each), else 1 x 1024 threads
// (D = 8192, the reference's default model: 96 KiB) -- the same 16 waves per CU either way (measured: 6.7 TB/s vs 5.5
// with 512 x 1).  CHIP_SCAN_BLOCK / CHIP_SCAN_BPC override (tuning only).
static size_t scan_lds_bytes(const Ctx *c, int nq, int K, int block)
{
    const size_t lds_q = (size_t)nq * c->D * sizeof(float);
    const size_t lds_m = (size_t)(block / 64) * nq * K * sizeof(chip_topk_entry);
    return lds_q > lds_m ? lds_q : lds_m;
}

static void scan_shape(const Ctx *c, int nq, int *block, int *bpc)
{
    if (c->scan_block > 0) { *block = c->scan_block; *bpc = c->scan_blocks_per_cu; return; }
    const bool two_fit = 2 * (scan_lds_bytes(c, nq, CHIP_MAX_TOPK, 512) + 1024) <= 160 * 1024;
    *block = two_fit ? 512 : 1024;
    *bpc = two_fit ? 2 : 1;
}

int scan_grid_for(const Ctx *c, int64_t n_rows, int nq)
{
    int block, bpc;
    scan_shape(c, nq, &block, &bpc);
    const int wpb = block / 64;
    int64_t want = (n_rows + wpb - 1) / wpb;
    // CHIP_SCAN_RESERVE (default 0) leaves workgroup slots free for the small kernels: a full grid holds every CU's registers
    // (2 x 8 waves x 110 VGPRs), so a merge launched underneath a running scan waits for a scan workgroup to retire.  Reserving
    // 4 slots cuts the merge's wait from ~1.7 ms to ~30 us at 1M rows for -0.15 % throughput; throughput is the headline, and a
    // lone synchronous tick has nothing running underneath it, so the default keeps the full grid.
    int64_t cap = (int64_t)c->n_cus * bpc - c->scan_reserve;
    if (cap > c->max_grid) cap = c->max_grid;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

int launch_scan(Ctx *c, hipStream_t s, const ScanArgs &a, int nq, int grid)
{
    int block, bpc;
    scan_shape(c, nq, &block, &bpc);
    const size_t lds = scan_lds_bytes(c, nq, a.K, block);
    if (lds > 160 * 1024 || grid > 512) return CHIP_ERR_UNSUPPORTED;  // K2 holds one partial list per thread
    switch (nq) {
        case 1: return launch_scan_q<1>(c, s, a, grid, lds, block);
        case 2: return launch_scan_q<2>(c, s, a, grid, lds, block);
        case 3: return launch_scan_q<3>(c, s, a, grid, lds, block);
        case 4: return launch_scan_q<4>(c, s, a, grid, lds, block);
    }
    return CHIP_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------ K2
// Merge (+ decision) of the per-workgroup lists of a scan, or of the per-GPU lists after the RCCL all-gather: one
// workgroup of 512 threads.
template <int NQ>
__global__ __launch_bounds__(512) void topk_merge(MergeArgs a)
{
    __shared__ __attribute__((aligned(16))) char smem[kMergeSmem];
    merge_sorted_lists<NQ>(a.in, a.n_lists, NQ, 0, a.K, a.out, a.result, a.l, a.locality, a.thresh, smem);
}

int launch_merge(Ctx *c, hipStream_t s, const MergeArgs &a, int nq)
{
    if (a.n_lists > 512) return CHIP_ERR_UNSUPPORTED;
    switch (nq) {
        case 1: hipLaunchKernelGGL(topk_merge<1>, dim3(1), dim3(512), 0, s, a); break;
        case 2: hipLaunchKernelGGL(topk_merge<2>, dim3(1), dim3(512), 0, s, a); break;
        case 3: hipLaunchKernelGGL(topk_merge<3>, dim3(1), dim3(512), 0, s, a); break;
        case 4: hipLaunchKernelGGL(topk_merge<4>, dim3(1), dim3(512), 0, s, a); break;
        default: return CHIP_ERR_UNSUPPORTED;
    }
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

// ------------------------------------------------------------------------------------------------ K3
struct StoreArgs {
    float *const *seg_table;
    int32_t seg_shift;
    int64_t seg_mask;
    float *ring;
    int32_t D;
    int32_t rank, nranks;
    int64_t first_global;
    int64_t n;
    int64_t ring_from;   // only global rows >= ring_from are mirrored into the ring (the newest CHIP_RING_ROWS)
    uint32_t *flags;
};

__device__ __forceinline__ void store_row4(const StoreArgs &a, int64_t g, int e, f32x4 v)
{
    if (a.nranks == 1 || (g % a.nranks) == a.rank) {
        const int64_t loc = a.nranks == 1 ? g : g / a.nranks;
        float *dst = a.seg_table[loc >> a.seg_shift] + (loc & a.seg_mask) * (int64_t)a.D + e;
        *reinterpret_cast<f32x4 *>(dst) = v;
    }
    if (a.ring && g >= a.ring_from) *reinterpret_cast<f32x4 *>(a.ring + (g % CHIP_RING_ROWS) * (int64_t)a.D + e) = v;
}

__global__ __launch_bounds__(256) void narrow_f64_rows(StoreArgs a, const double *__restrict__ src)
{
    const int64_t per_row = a.D / 4;
    const int64_t total = a.n * per_row;
    uint32_t bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int e = (int)(i - r * per_row) * 4;
        const f64x2 lo = *reinterpret_cast<const f64x2 *>(src + r * a.D + e);
        const f64x2 hi = *reinterpret_cast<const f64x2 *>(src + r * a.D + e + 2);
        const double x[4] = {lo.x, lo.y, hi.x, hi.y};
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float f = (float)x[c];  // round-to-nearest-even
            if (!(fabs(x[c]) <= 1.79769313486231570e308)) bad |= 2u;         // NaN / Inf
            else if ((double)f != x[c]) bad |= 1u;                             // not fp32-representable (incl. overflow)
            v[c] = f;
        }
        store_row4(a, a.first_global + r, e, v);
    }
    if (bad) atomicOr(a.flags, bad);
}

__global__ __launch_bounds__(256) void copy_f32_rows(StoreArgs a, const float *__restrict__ src)
{
    const int64_t per_row = a.D / 4;
    const int64_t total = a.n * per_row;
    uint32_t bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int e = (int)(i - r * per_row) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + r * a.D + e);
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (!(fabsf(v[c]) <= 3.402823466e38f)) bad |= 2u;
        store_row4(a, a.first_global + r, e, v);
    }
    if (bad) atomicOr(a.flags, bad);
}

static StoreArgs make_store_args(Ctx *c, int64_t first_global, int64_t n, uint32_t *flags, bool write_ring = true)
{
    StoreArgs a;
    a.seg_table = c->seg_table_dev;
    a.seg_shift = c->seg_shift;
    a.seg_mask = c->seg_rows - 1;
    a.ring = write_ring ? c->ring_dev : nullptr;
    a.D = c->D;
    a.rank = c->rank;
    a.nranks = c->nranks;
    a.first_global = first_global;
    a.n = n;
    a.ring_from = first_global + n - CHIP_RING_ROWS;
    a.flags = flags;
    return a;
}

static int grid_for_elems(const Ctx *c, int64_t total_threads)
{
    int64_t g = (total_threads + 255) / 256;
    int64_t cap = (int64_t)c->n_cus * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int launch_narrow_f64(Ctx *c, hipStream_t s, const double *src, int64_t n, int64_t first_global, uint32_t *flags_dev, bool write_ring)
{
    StoreArgs a = make_store_args(c, first_global, n, flags_dev, write_ring);
    hipLaunchKernelGGL(narrow_f64_rows, dim3(grid_for_elems(c, n * (c->D / 4))), dim3(256), 0, s, a, src);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

int launch_copy_f32(Ctx *c, hipStream_t s, const float *src, int64_t n, int64_t first_global, uint32_t *flags_dev, bool write_ring)
{
    StoreArgs a = make_store_args(c, first_global, n, flags_dev, write_ring);
    hipLaunchKernelGGL(copy_f32_rows, dim3(grid_for_elems(c, n * (c->D / 4))), dim3(256), 0, s, a, src);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

// ------------------------------------------------------------------------------------------------ synth
// Spec: oracle/dot_scan.c (orc_splitmix64 / orc_synth_row_f32).  Integer domain + one exact int->float
// conversion + one float multiply => bit-identical to the CPU generator.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t synth_rowkey(uint64_t seed, int64_t row)
{
    return splitmix64(seed + 0x632BE59BD9B4E019ULL * (uint64_t)row);
}
__device__ __forceinline__ int32_t synth_from_key(uint64_t key, int32_t e)
{
    const uint64_t h = splitmix64(key + (uint64_t)(uint32_t)e);
    const int32_t s = (int32_t)(h & 0xFFFF) + (int32_t)((h >> 16) & 0xFFFF) + (int32_t)((h >> 32) & 0xFFFF) + (int32_t)(h >> 48);
    return s - 131070;
}

struct SynthArgs {
    StoreArgs st;
    uint64_t seed;
    float scale, scale_planted;
    const int64_t *plant_dst, *plant_src;
    const int32_t *plant_kind;
    int64_t n_plant;
};

__global__ __launch_bounds__(256) void synth_rows(SynthArgs a)
{
    const int64_t per_row = a.st.D / 4;
    const int64_t total = a.st.n * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int e = (int)(i - r * per_row) * 4;
        const int64_t g = a.st.first_global + r;
        // planted? (binary search over the sorted dst list)
        int32_t kind = 0;
        int64_t src = -1;
        int64_t lo = 0, hi = a.n_plant;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.plant_dst[mid] < g) lo = mid + 1; else hi = mid;
        }
        if (lo < a.n_plant && a.plant_dst[lo] == g) { kind = a.plant_kind[lo]; src = a.plant_src[lo]; }
        const uint64_t key = synth_rowkey(a.seed, g);
        const uint64_t skey = kind ? synth_rowkey(a.seed, src) : 0;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (kind == 0) v[c] = (float)synth_from_key(key, e + c) * a.scale;
            else if (kind == 2) v[c] = (float)synth_from_key(skey, e + c) * a.scale;
            else v[c] = (float)(5 * synth_from_key(skey, e + c) + synth_from_key(key, e + c)) * a.scale_planted;
        }
        store_row4(a.st, g, e, v);
    }
}

int launch_synth(Ctx *c, hipStream_t s, int64_t first_global, int64_t n, uint64_t seed,
                 const int64_t *plant_dst_dev, const int64_t *plant_src_dev, const int32_t *plant_kind_dev, int64_t n_plant)
{
    SynthArgs a;
    a.st = make_store_args(c, first_global, n, nullptr);
    a.seed = seed;
    const double var = 1431655765.0;  // 4 * (65536^2 - 1) / 12
    a.scale = (float)(1.0 / std::sqrt((double)c->D * var));
    a.scale_planted = (float)(1.0 / std::sqrt((double)c->D * var * 26.0));
    a.plant_dst = plant_dst_dev;
    a.plant_src = plant_src_dev;
    a.plant_kind = plant_kind_dev;
    a.n_plant = n_plant;
    hipLaunchKernelGGL(synth_rows, dim3(grid_for_elems(c, n * (c->D / 4))), dim3(256), 0, s, a);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

}  // namespace chip
