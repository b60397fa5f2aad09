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
// Storage type T is float (default: NetVLAD descriptors are float32 values on a float64 wire) or double (descriptors that
// are genuinely float64, e.g. ReljaNetVLAD's numpy WPCA output, whole_image_desc_compute_server.py:148-149).
// CDNA4 notes: wave = 64; queries live in LDS (NQ*D*sizeof(T) B) and are read with conflict-free ds_read_b128;
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
template <typename T> struct Vec16;                       // 16 bytes of storage elements = one lane's share of a wave load
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };
template <> struct Vec16<double> { typedef f64x2 type; static constexpr int N = 2; };

// DB rows are reached through a pointer loaded from the segment table, which the compiler can only treat as a GENERIC pointer:
// it would emit flat_load, which is tracked by BOTH vmcnt and lgkmcnt -- every wait for an LDS read (the staged queries) then
// also waits for the row loads in flight.  The explicit address-space cast makes them global_load (vmcnt only).
template <typename V>
__device__ __forceinline__ const __attribute__((address_space(1))) V *as_global(const V *p)
{
    return (const __attribute__((address_space(1))) V *)p;
}

template <int POLICY, typename V>
__device__ __forceinline__ V stream_load(const V *p)
{
    if constexpr (POLICY == 0) return *as_global(p);
    else if constexpr (POLICY == 1) return __builtin_nontemporal_load(as_global(p));
    else {
        V v;
        if constexpr (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
        else if constexpr (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
        else if constexpr (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
}

// Dot products of one DB row with NQ staged queries in the fixed order of DESIGN.md 3: lane L accumulates elements
// j*CH + N*L + c (j ascending, c = 0..N-1; N = 4 / CH = 256 for float storage, N = 2 / CH = 128 for double) into one fp64
// accumulator per query by fma -- for float storage the product of two fp32 values is exact in fp64, so fma == mul-then-add;
// for double storage the single rounding of the fma IS the definition (oracle: orc_dot_tree_f64) -- then the xor butterfly
// acc[L] += acc[L ^ m], m = 32..1, after which every lane holds the same bits.
template <typename T, int NQ, int U, bool FULL, int NT, int R>
__device__ __forceinline__ void rows_dot(const T *const (&row)[R], const T *qs, int D, int lane, double (&acc)[R][NQ])
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N, CH = 64 * N;
    const int e0 = lane * N;
#pragma unroll
    for (int rr = 0; rr < R; rr++)
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[rr][q] = 0.0;
    for (int base = 0; base < D; base += CH * U) {
        V v[R][U];
        if constexpr (NT == 6 || NT == 8) {
            // all R*U loads of the batch issued back to back from one address register pair (immediate offsets), consumed
            // below behind COUNTED waits: hipcc's scheduler otherwise sinks the loads next to their uses (register pressure)
            // and the wave runs with 1-2 KiB in flight instead of U KiB
            static_assert(FULL && U <= 8, "asm path: whole batches only");
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                const char *mid = reinterpret_cast<const char *>(row[rr] + base + e0) + 4096;
#pragma unroll
                for (int u = 0; u < U; u++)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(v[rr][u]) : "v"(mid), "n"(u * 1024 - 4096) : "memory");
            }
        } else {
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = base + u * CH + e0;
                if (FULL || e < D) v[rr][u] = stream_load<NT>(reinterpret_cast<const V *>(row[rr] + e));
            }
        }
        }
        if constexpr (NT >= 2 && NT != 6 && NT != 8) {   // inline-asm loads: the compiler does not track them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int rr = 0; rr < R; rr++)
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" : "+v"(v[rr][u]));
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = base + u * CH + e0;
            if constexpr (NT == 6 || NT == 8) {
#pragma unroll
                for (int rr = 0; rr < R; rr++)   // loads return in order: load (rr, u) is done once at most this many are outstanding
                    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[rr][u]) : "n"((R - 1 - rr) * U + (U - 1 - u) < 63 ? (R - 1 - rr) * U + (U - 1 - u) : 63) : "memory");
            }
            if constexpr (NT == 8) {
                // queries staged as fp64 (scan_q64): per 64 vectors of 4 elements [64 x (d0, d1)][64 x (d2, d3)] = 2 KiB,
                // read as two conflict-free 16-B loads per lane; the products are the same exact fp64 values as (double)w * (double)v
                const char *q64 = reinterpret_cast<const char *>(qs);
                const int t = e >> 2;                                   // vector index of this lane
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const char *blk = q64 + (size_t)q * D * 8 + (size_t)(t >> 6) * 2048 + (t & 63) * 16;
                    const f64x2 wl = *reinterpret_cast<const f64x2 *>(blk), wh = *reinterpret_cast<const f64x2 *>(blk + 1024);
#pragma unroll
                    for (int rr = 0; rr < R; rr++) {
                        acc[rr][q] = __builtin_fma(wl[0], (double)v[rr][u][0], acc[rr][q]);
                        acc[rr][q] = __builtin_fma(wl[1], (double)v[rr][u][1], acc[rr][q]);
                        acc[rr][q] = __builtin_fma(wh[0], (double)v[rr][u][2], acc[rr][q]);
                        acc[rr][q] = __builtin_fma(wh[1], (double)v[rr][u][3], acc[rr][q]);
                    }
                }
            } else
            if (FULL || e < D) {
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const V w = *reinterpret_cast<const V *>(qs + q * D + e);
#pragma unroll
                    for (int rr = 0; rr < R; rr++)
#pragma unroll
                        for (int c = 0; c < N; c++) acc[rr][q] = __builtin_fma((double)w[c], (double)v[rr][u][c], acc[rr][q]);
                }
            }
        }
    }
}

__device__ __forceinline__ double butterfly_sum(double a)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a = a + __shfl_xor(a, m, 64);
    return a;
}

template <typename T, int NQ>
__device__ __forceinline__ void stage_queries(const ScanArgs &a, T *qs, int tid, int nthreads)
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const T *src = static_cast<const T *>(a.q[q]);
        for (int e = tid * N; e < a.D; e += nthreads * N)
            *reinterpret_cast<V *>(qs + q * a.D + e) = *reinterpret_cast<const V *>(src + e);
    }
}

template <typename T>
__device__ __forceinline__ const T *row_base(const ScanArgs &a, int64_t r)
{
    return static_cast<const T *>(a.seg_table[r >> a.seg_shift]) + (r & a.seg_mask) * (int64_t)a.D;
}

// Per-wave running top-K of one query: lane j < K holds the j-th best (score desc, index desc); (thr_s, thr_i) = the K-th.
// Offering a candidate is wave-uniform and rare after warm-up; NaN never enters.
__device__ __forceinline__ void wave_topk_offer(double s, int64_t gi, int K, int lane, double &my_s, int64_t &my_i, double &thr_s, int64_t &thr_i)
{
    if (key_gt(s, gi, thr_s, thr_i)) {
        const bool worse = key_gt(s, gi, my_s, my_i);
        const unsigned long long m = __ballot(worse) & ((1ull << K) - 1ull);
        const int pos = __builtin_ctzll(m);
        const double up_s = __shfl_up(my_s, 1, 64);
        const int64_t up_i = __shfl_up(my_i, 1, 64);
        if (lane < K) {
            if (lane > pos) { my_s = up_s; my_i = up_i; }
            else if (lane == pos) { my_s = s; my_i = gi; }
        }
        thr_s = __shfl(my_s, K - 1, 64);
        thr_i = __shfl(my_i, K - 1, 64);
    }
}

// Block merge: wpb sorted lists of K per query -> one sorted list of K per query in a.partial[blockIdx.x], by waves 0..NQ-1.
template <int NQ>
__device__ __forceinline__ void block_merge_store(const ScanArgs &a, char *smem, const double (&my_s)[NQ], const int64_t (&my_i)[NQ],
                                                  int K, int lane, int wave, int wpb)
{
    __syncthreads();  // all waves done with the staged queries; reuse LDS
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

// The same for a WAVE-UNIFORM row through the scalar cache: the table is read with s_load (constant address space; entries of
// published rows never change), so the lookup neither costs a vector-memory round trip nor touches vmcnt.
template <typename T>
__device__ __forceinline__ const T *row_base_uniform(const ScanArgs &a, int64_t r)
{
    typedef const uint64_t __attribute__((address_space(4))) *ctab;
    const uint64_t seg = ((ctab)(uintptr_t)a.seg_table)[r >> a.seg_shift];
    return reinterpret_cast<const T *>((uintptr_t)seg) + (r & a.seg_mask) * (int64_t)a.D;
}

template <typename T, int NQ, int U, bool FULL, int NT, int R>
__global__ __launch_bounds__(1024) void db_scan_topk(ScanArgs a)
{
    static_assert(R == 1, "one row per wave at a time");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *qs = reinterpret_cast<T *>(smem);  // [NQ][D]
    const int D = a.D;
    const int K = a.K;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpb = blockDim.x >> 6;

    // stage the query descriptors once per block (L2-resident after the first block)
    if constexpr (NT == 8) {
        char *q64 = reinterpret_cast<char *>(smem);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float *src = static_cast<const float *>(a.q[q]);
            for (int t = tid; t < (a.D >> 2); t += blockDim.x) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(src + 4 * t);
                char *blk = q64 + (size_t)q * a.D * 8 + (size_t)(t >> 6) * 2048 + (t & 63) * 16;
                *reinterpret_cast<f64x2 *>(blk) = (f64x2){(double)w[0], (double)w[1]};
                *reinterpret_cast<f64x2 *>(blk + 1024) = (f64x2){(double)w[2], (double)w[3]};
            }
        }
    } else {
        stage_queries<T, NQ>(a, qs, tid, blockDim.x);
    }
    __syncthreads();

    // per-wave running top-K: lane j < K holds the j-th best (score desc, index desc)
    double my_s[NQ], thr_s[NQ];
    int64_t my_i[NQ], thr_i[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { my_s[q] = -INFINITY; my_i[q] = -1; thr_s[q] = -INFINITY; thr_i[q] = -1; }

    // Row -> wave map: wave g (global index) owns rows g, g + tw, g + 2 tw, ... -- adjacent waves read adjacent rows and all
    // waves together read ONE window of tw consecutive rows that slides through the DB.  Measured alternatives, all slower or
    // not worth their machinery: a contiguous run of rows per wave / per workgroup / per XCD (-3 .. -6 %);
    // chunks claimed from a device-wide counter so that the XCDs that get less HBM bandwidth claim fewer (same-address atomics
    // sustain ~85 M/s and VMEM returns in order, so the claiming wave's row loads wait for its atomic: -2 %); a static 3-5 %
    // skew of the row share between even and odd XCDs (+0.5 .. +1.9 % depending on the box).
    const int64_t tw = (int64_t)gridDim.x * wpb;
    for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < a.n_rows; r += tw) {
        const T *row[1] = {row_base<T>(a, r)};
        double acc[1][NQ];
        rows_dot<T, NQ, U, FULL, NT, 1>(row, qs, D, lane, acc);
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[0][q] = butterfly_sum(acc[0][q]);
        const int64_t gi = r * a.idx_mul + a.idx_add;
#pragma unroll
        for (int q = 0; q < NQ; q++) wave_topk_offer(acc[0][q], gi, K, lane, my_s[q], my_i[q], thr_s[q], thr_i[q]);
    }

    block_merge_store<NQ>(a, smem, my_s, my_i, K, lane, wave, wpb);
}

// ------------------------------------------------------------------------------------------------ K1, wide double rows
// Double rows whose NQ query descriptors do not fit the 160 KiB of LDS (the reference's default D = 8192, src/Cerebro.cpp:1021, as a
// MatrixXd of genuine float64 values: 3 x 8192 x 8 B = 192 KiB).  The first NQ - NG queries are staged as usual; the last NG are read
// from global memory next to the row -- every wave of the launch reads the same D x 8 bytes, so they come out of the L2 -- with the
// same per-lane element order as rows_dot (lane L: elements j*128 + 2L + c, j ascending, fma; then the xor butterfly), hence the
// same bits as the staged form and as the oracle (orc_dot_tree_f64).  A rare configuration: plain compiler-scheduled loads.
template <int NQ, int NG, bool FULL>
__global__ __launch_bounds__(1024) void db_scan_topk_wide(ScanArgs a)
{
    static_assert(NG >= 1 && NG < NQ, "at least one query staged, at least one read in place");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *qs = reinterpret_cast<double *>(smem);  // [NQ - NG][D]
    constexpr int NL = NQ - NG, U = 4, CH = 128;
    const int D = a.D, K = a.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpb = blockDim.x >> 6;
    stage_queries<double, NL>(a, qs, tid, blockDim.x);
    __syncthreads();
    const double *qg[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) qg[g] = static_cast<const double *>(a.q[NL + g]);

    double my_s[NQ], thr_s[NQ];
    int64_t my_i[NQ], thr_i[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { my_s[q] = -INFINITY; my_i[q] = -1; thr_s[q] = -INFINITY; thr_i[q] = -1; }
    const int e0 = lane * 2;
    const int64_t tw = (int64_t)gridDim.x * wpb;
    for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < a.n_rows; r += tw) {
        const double *row = row_base<double>(a, r);
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[q] = 0.0;
        for (int base = 0; base < D; base += CH * U) {
            f64x2 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = base + u * CH + e0;
                if (FULL || e < D) v[u] = __builtin_nontemporal_load(as_global(reinterpret_cast<const f64x2 *>(row + e)));
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = base + u * CH + e0;
                if (FULL || e < D) {
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const f64x2 w = q < NL ? *reinterpret_cast<const f64x2 *>(qs + q * D + e)
                                               : *as_global(reinterpret_cast<const f64x2 *>(qg[q < NL ? 0 : q - NL] + e));
                        acc[q] = __builtin_fma(w[0], v[u][0], acc[q]);
                        acc[q] = __builtin_fma(w[1], v[u][1], acc[q]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[q] = butterfly_sum(acc[q]);
        const int64_t gi = r * a.idx_mul + a.idx_add;
#pragma unroll
        for (int q = 0; q < NQ; q++) wave_topk_offer(acc[q], gi, K, lane, my_s[q], my_i[q], thr_s[q], thr_i[q]);
    }
    block_merge_store<NQ>(a, smem, my_s, my_i, K, lane, wave, wpb);
}

// ------------------------------------------------------------------------------------------------ K1, row-batched form
// The same scan with R rows per wave in flight at once and the loads issued as ONE continuous stream (round 3): a wave walks
// the (pass, 4 KiB batch, load slot u) sequence of its rows and re-issues slot u for the NEXT batch as soon as the fmas that
// read slot u of the current batch have been issued, so R x 4 KiB per wave stay in flight for the whole scan instead of
// draining to zero at every batch (db_scan_topk above).  Three things follow for SHORT prefixes (BASELINE config 2: 10 003 rows
// over 4096 waves = 2.4 rows per wave, which the one-row kernel walks as 12 dependent 4 KiB batches per wave):
//   * R = 3 turns a wave's whole share into one pass of 4 batches with 12 KiB in flight;
//   * the queries are staged by LDS-DMA (global_load_lds_dwordx4, no staging VGPRs), issued BEFORE the first row loads and
//     waited for with a counted vmcnt, so the first 12 KiB of DB rows are already on their way while the queries land;
//   * each query element is converted to fp64 once per R rows (R = 3: 20 instead of 28 VALU instructions per KiB of DB).
// Arithmetic and order per (row, query) are those of rows_dot: lane L accumulates elements j*CH + N*L + c, j ascending, into
// one fp64 accumulator by fma, then the xor butterfly -- bit-identical scores, same (score desc, index desc) lists.
// Loads use the SGPR-base form (row base in an SGPR pair, one shared 32-bit lane offset), immediate offsets u * 1 KiB.
// Requires whole 4 KiB batches (D * sizeof(T) % 4096 == 0) and queries that are 16-B aligned rows of D elements.
__device__ __forceinline__ uint32_t lds_byte_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
// 16 bytes per lane, global -> LDS at (wave-uniform LDS base in M0) + lane * 16; M0 is saved and restored around the load
__device__ __forceinline__ void glds16_q(const void *gsrc_lane, uint32_t lds_base_wave_uniform)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_base_wave_uniform)
                 : "memory");
}

// a wave-uniform pointer the compiler cannot prove uniform (it came out of the segment table) -> SGPR pair
template <typename P>
__device__ __forceinline__ const P *uniform_ptr(const P *p)
{
    const uint64_t b = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return reinterpret_cast<const P *>((uintptr_t)(((uint64_t)hi << 32) | lo));
}

// The running top-K lists of the row-batched kernel live in LDS ([wave][query][CHIP_MAX_TOPK] entries behind the staged
// queries), only the admission threshold (the K-th entry) stays in SGPRs: after warm-up an offer is rare, and lists held in
// VGPRs (as in db_scan_topk) would cost 4 NQ registers of a kernel that keeps R x 16 of them in flight as load targets.
__device__ __forceinline__ void wave_topk_offer_lds(double s, int64_t gi, int K, int lane, chip_topk_entry *list, double &thr_s, int64_t &thr_i)
{
    if (key_gt(s, gi, thr_s, thr_i)) {   // wave-uniform; NaN never enters
        chip_topk_entry me, up;
        me.score = -INFINITY; me.idx = -1; up = me;
        if (lane < K) { me = list[lane]; if (lane > 0) up = list[lane - 1]; }
        const bool worse = key_gt(s, gi, me.score, me.idx);
        const unsigned long long m = __ballot(worse) & ((1ull << K) - 1ull);
        const int pos = __builtin_ctzll(m);
        if (lane < K) {
            if (lane > pos) me = up;
            else if (lane == pos) { me.score = s; me.idx = gi; }
            list[lane] = me;      // every read of the old list precedes this write in program order (one wave, in-order LDS)
        }
        thr_s = readlane_f64(me.score, K - 1);
        thr_i = readlane_i64(me.idx, K - 1);
    }
}

// Block merge over the LDS lists: wpb sorted lists of K per query -> one sorted list of K per query in a.partial[blockIdx.x].
// device-coherent 16-byte entry store / load (agent-scope relaxed atomics: write-through `sc1` stores and loads, no cache-wide fence)
__device__ __forceinline__ void store_entry_agent(chip_topk_entry *p, double s, int64_t i)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&p->score), (unsigned long long)__double_as_longlong(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&p->idx), (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ chip_topk_entry load_entry_agent(const chip_topk_entry *p)
{
    chip_topk_entry e;
    e.score = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(&p->score), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    e.idx = (int64_t)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(&p->idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return e;
}

template <int NQ>
__device__ __forceinline__ void block_merge_lists(const ScanArgs &a, const chip_topk_entry *lists, int K, int lane, int wave, int wpb, bool coherent = false)
{
    __syncthreads();
    for (int q = wave; q < NQ; q += wpb) {
        const int ncand = wpb * K;  // <= 16 waves * 16 = 256 -> at most 4 per lane
        double cs[4];
        int64_t ci[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int c = lane + 64 * h;
            if (c < ncand) {
                const chip_topk_entry t = lists[((c / K) * NQ + q) * CHIP_MAX_TOPK + (c % K)];
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
            if (lane == 0) {
                if (coherent) store_entry_agent(outp + j, bs, bi);
                else { chip_topk_entry t; t.score = bs; t.idx = bi; outp[j] = t; }
            }
        }
    }
}

// Load targets of db_scan_topk_rows.  The loads are issued from inline asm and consumed behind counted s_waitcnt -- a register
// that the compiler believed "defined" at the load statement could be copied or spilled by it BEFORE the wait (it happened: the
// first version of this kernel, with "=v" outputs, got v_mov copies of in-flight registers at a control-flow merge).  So the
// targets are PHYSICAL registers the compiler does not own: the kernel is compiled with amdgpu_num_vgpr(kRowsVgprBase / 2) -- on
// gfx950's unified VGPR/AGPR file hipcc (ROCm 7.2) takes that attribute as HALF the ArchVGPR budget (measured: 24 -> 48, 32 -> 64,
// 40 -> 80 registers, spilling beyond) -- which leaves v[kRowsVgprBase ..] to the asm statements; a load names its target
// literally, and the statement that waits for it also moves the data (converted to fp64 for float rows) into compiler-owned
// outputs -- wait and first read are one asm block.  tests/test_codeobj_registers.py disassembles the built library and checks
// the partition: nothing but these loads writes v[80..127], nothing but the take statements reads them.
constexpr int kRowsVgprBase = 80;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "db_scan_topk_rows names physical VGPRs and uses global_load_lds_dwordx4: gfx950 (MI355X) only -- build with --offload-arch=gfx950"
#endif
#define CHIP_ROWS_CLOBBERS                                                                                                      \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97",   \
    "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
    "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

// slot (u, rr) of an R-row batch -> v[base : base + 3]
template <int R> __host__ __device__ constexpr int rows_slot_reg(int u, int rr) { return kRowsVgprBase + 4 * (u * R + rr); }

template <bool NTL, int REG, int OFF>
__device__ __forceinline__ void rows_issue(uint32_t voff, const void *row_uniform)
{
    if constexpr (NTL)   // non-temporal: a stream that is read once (prefixes beyond the Infinity Cache)
        asm volatile("global_load_dwordx4 v[%2:%3], %0, %1 offset:%4 nt" ::"v"(voff), "s"(row_uniform), "n"(REG), "n"(REG + 3), "n"(OFF) : "memory");
    else                 // temporal: a prefix that fits the 256 MiB Infinity Cache is re-read by every tick
        asm volatile("global_load_dwordx4 v[%2:%3], %0, %1 offset:%4" ::"v"(voff), "s"(row_uniform), "n"(REG), "n"(REG + 3), "n"(OFF) : "memory");
}

// wait until at most CNT vector-memory operations are outstanding, then hand one 8-byte HALF (H = 0, 1) of the slot's 16 bytes
// to the compiler as fp64: two elements of a float row, one of a double row (halves keep the compiler-side live set small)
template <int REG, int CNT, int H>
__device__ __forceinline__ void rows_take(double (&d)[2], float)
{
    asm volatile("s_waitcnt vmcnt(%2)\n\tv_cvt_f64_f32 %0, v[%3]\n\tv_cvt_f64_f32 %1, v[%4]"
                 : "=&v"(d[0]), "=&v"(d[1])
                 : "n"(CNT), "n"(REG + 2 * H), "n"(REG + 2 * H + 1)
                 : "memory");
}
template <int REG, int CNT, int H>
__device__ __forceinline__ void rows_take(double (&d)[2], double)
{
    asm volatile("s_waitcnt vmcnt(%1)\n\tv_mov_b64 %0, v[%2:%3]" : "=&v"(d[0]) : "n"(CNT), "n"(REG + 2 * H), "n"(REG + 2 * H + 1) : "memory");
    d[1] = 0.0;
}

// Fused tick: the decision of Cerebro.cpp:1035-1056 needs the BEST entry per query only (chip_tick_result carries argmax / maxv, no
// lists), and the best of the whole prefix is the best of the per-workgroup bests under the same total order (score desc, index
// desc).  So a tick runs the scan with K = 1 and the last workgroup to arrive (device-scope ticket) reduces the <= 512 per-workgroup
// entries -- one per thread, wave butterfly, eight waves through LDS -- applies the accept rule and stores the 64-byte record straight
// into the pinned host slot: no K2 launch, no launch gap, no second kernel's ramp (a 10k-row tick spent ~15 us of its stream's time in
// the one-workgroup merge and ~6 us in front of it, profiles/r03_tick_timeline_10k.md).  Hand-off: write-through stores + completion
// wait + device-scope ticket on the writers' side, L2-bypassing loads on the last workgroup's side (no cache-wide fences).
// What one pass of the row-batched scan is asked to do -- the fields of ScanArgs that differ from pass to pass.  A launch takes them from
// its kernel arguments; the resident instance from the command it was sent (they then live in SGPRs: no arrays, nothing indexed).
struct RowsPass {
    int64_t n_rows;
    const void *q0, *q1, *q2, *q3;
    int32_t dyn_claim;
    chip_tick_result *fused_result;
    unsigned long long *fused_seq;
    unsigned long long fused_seq_val;
    int64_t tick_l;
    int32_t locality;
    double thresh;
};
__device__ __forceinline__ RowsPass rows_pass_of(const ScanArgs &a)
{
    RowsPass t;
    t.n_rows = a.n_rows;
    t.q0 = a.q[0]; t.q1 = a.q[1]; t.q2 = a.q[2]; t.q3 = a.q[3];
    t.dyn_claim = a.dyn_claim;
    t.fused_result = a.fused_result;
    t.fused_seq = a.fused_seq;
    t.fused_seq_val = a.fused_seq_val;
    t.tick_l = a.tick_l;
    t.locality = a.locality;
    t.thresh = a.thresh;
    return t;
}

template <int NQ, bool TOGETHER, bool XSTAMP>
__device__ __forceinline__ void fused_tick_finish(const ScanArgs &a, const RowsPass &t, char *smem, int K, int tid, int lane, int wave, int wpb)
{
    int *last = reinterpret_cast<int *>(smem);                                  // LDS is free again: the lists have been consumed
    chip_topk_entry *wbest = reinterpret_cast<chip_topk_entry *>(smem + 64);    // [wpb][NQ]
    // This workgroup's entries went out as write-through (agent-scope) stores; once they have completed (vmcnt(0)) they are visible
    // device-wide, and the ticket may be taken.  No __threadfence(): an agent-scope release fence writes back this XCD's L2 -- with
    // 256-512 workgroups doing it the tick took 120 us instead of 25 (measured; it is also why round 2's fused merge lost).
    //   Writers (every workgroup): what stands in for the release is an ISA property of gfx942 / gfx950, not the HIP memory model --
    //   an sc1 (agent-scope) store is written THROUGH its XCD's L2 to the fabric, and its vmcnt slot is only returned when the
    //   write has been acknowledged there; "s_waitcnt vmcnt(0)" + the workgroup barrier therefore order this workgroup's entries
    //   before its ticket increment for every agent-scope observer.  The code object is gfx950-only (chip_create refuses anything
    //   else), and tests/test_scan_gpu.py::test_fused_tick_handoff_stress compares >= 200k fused ticks with the two-launch path.
    //   Reader (ONE workgroup): a real agent-scope ACQUIRE after the ticket is won -- it invalidates this XCD's non-coherent L2
    //   lines and the CU's vector L1 (buffer_inv sc1), so no entry of the launch that used this list buffer 64 ticks ago can be
    //   served from a cache; one workgroup per tick pays it, which is free (the 120 us were 256-512 workgroups RELEASING).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *last = __hip_atomic_fetch_add(a.fused_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!*last) return;
    unsigned long long *xstamp = nullptr;   // tuning only (the resident instance with CHIP_SCAN_STAMPS=1): pass-wide stamps behind the waves'
    if constexpr (XSTAMP) xstamp = a.stamps ? a.stamps + (size_t)gridDim.x * wpb * 4 : nullptr;
    if (xstamp && tid == 0) xstamp[5] = (unsigned long long)wall_clock64();
    // (The resident instance leaves the fence out: everything this workgroup reads from memory from here on is read by agent-scope loads,
    //  which are coherent at that scope by themselves -- 0.84 us of the pass, measured; tests/test_resident_gpu.py compares every record
    //  of tens of thousands of consecutive, different commands.  The launched kernels keep it: belt and braces.)
    if constexpr (!XSTAMP) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (xstamp && tid == 0) xstamp[2] = (unsigned long long)wall_clock64();
    // One entry per workgroup of the launch and query; grid-stride, so a block smaller than the grid (CHIP_SCAN_BLOCK=256) drops
    // nothing.  TOGETHER (the R = 1 kernels, which have the registers for it): the NQ loads of a workgroup's entries are issued
    // before the first compare -- one round trip to memory instead of NQ; the others reduce one query at a time.
    auto reduce_store = [&](double s, int64_t i, int q) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double os = __shfl_xor(s, m, 64);
            const int64_t oi = __shfl_xor(i, m, 64);
            if (key_gt(os, oi, s, i)) { s = os; i = oi; }
        }
        if (lane == 0) { chip_topk_entry e; e.score = s; e.idx = i; wbest[wave * NQ + q] = e; }
    };
    if constexpr (TOGETHER) {
        double bs[NQ];
        int64_t bi[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) { bs[q] = -INFINITY; bi[q] = -1; }
        for (int wg = tid; wg < (int)gridDim.x; wg += (int)blockDim.x) {
            chip_topk_entry e[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) e[q] = load_entry_agent(a.partial + ((int64_t)wg * NQ + q) * K);   // bypasses this XCD's L2
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (key_gt(e[q].score, e[q].idx, bs[q], bi[q])) { bs[q] = e[q].score; bi[q] = e[q].idx; }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) reduce_store(bs[q], bi[q], q);
    } else {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            double bs = -INFINITY;
            int64_t bi = -1;
            for (int wg = tid; wg < (int)gridDim.x; wg += (int)blockDim.x) {
                const chip_topk_entry e = load_entry_agent(a.partial + ((int64_t)wg * NQ + q) * K);   // bypasses this XCD's L2
                if (key_gt(e.score, e.idx, bs, bi)) { bs = e.score; bi = e.idx; }
            }
            reduce_store(bs, bi, q);
        }
    }
    if (xstamp && tid == 0) xstamp[6] = (unsigned long long)wall_clock64();
    // the workgroup pairs' row-claim counters of this launch's list buffer: back to zero for the launch that reuses it (every workgroup
    // has claimed its last row before it took the ticket, and this is the workgroup that took the last one)
    if (a.pair_ctr != nullptr)
        for (int i = tid; i < (int)(gridDim.x >> 1); i += (int)blockDim.x) __hip_atomic_store(a.pair_ctr + (size_t)i * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (wave != 0) return;
    if (xstamp && tid == 0) xstamp[7] = (unsigned long long)wall_clock64();
    // the waves' bests -> the launch's best per query.  TOGETHER: lane 16 q + w takes wave w's entry of query q and a four-step
    // butterfly inside each group of 16 lanes finishes it (a max under a total order: the same entry whatever the order); the others
    // keep the serial loop of one lane (1.8 us of a 10k-row tick, measured -- but no extra registers).
    double maxv[3] = {-INFINITY, -INFINITY, -INFINITY};
    int64_t argmax[3] = {-1, -1, -1};
    if constexpr (TOGETHER) {
        const int q = lane >> 4, w = lane & 15;
        double sc = -INFINITY;
        int64_t ix = -1;
        if (q < NQ && w < wpb) { const chip_topk_entry e = wbest[w * NQ + q]; sc = e.score; ix = e.idx; }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            const double os = __shfl_xor(sc, m, 64);
            const int64_t oi = __shfl_xor(ix, m, 64);
            if (key_gt(os, oi, sc, ix)) { sc = os; ix = oi; }
        }
#pragma unroll
        for (int qq = 0; qq < 3; qq++)
            if (qq < NQ) { maxv[qq] = readlane_f64(sc, 16 * qq); argmax[qq] = readlane_i64(ix, 16 * qq); }
    } else {
        for (int q = 0; q < 3; q++)
            if (q < NQ)
                for (int w = 0; w < wpb; w++) {
                    const chip_topk_entry e = wbest[w * NQ + q];
                    if (key_gt(e.score, e.idx, maxv[q], argmax[q])) { maxv[q] = e.score; argmax[q] = e.idx; }
                }
    }
    if (lane == 0) {
        chip_tick_result res;
        res.status = CHIP_TICK_SCANNED;
        res.found = 0;
        res.idx_curr = -1;
        res.idx_prev = -1;
        res.score = 0.0;
        for (int q = 0; q < 3; q++) { res.argmax[q] = argmax[q]; res.maxv[q] = maxv[q]; }
        if (NQ >= 3 && res.argmax[0] >= 0 && res.argmax[1] >= 0 && res.argmax[2] >= 0) {
            // Cerebro.cpp:1056  abs(u_argmax-um_argmax) < LOCALITY && abs(u_argmax-umm_argmax) < LOCALITY && u_max > THRESH
            int64_t d1 = res.argmax[0] - res.argmax[1];
            int64_t d2 = res.argmax[0] - res.argmax[2];
            if (d1 < 0) d1 = -d1;
            if (d2 < 0) d2 = -d2;
            if (d1 < t.locality && d2 < t.locality && res.maxv[0] > t.thresh) {
                res.found = 1;
                res.idx_curr = t.tick_l - 1;  // Cerebro.cpp:1080
                res.idx_prev = res.argmax[0];
                res.score = res.maxv[0];
            }
        }
        *t.fused_result = res;
        // ready for the pass that reuses this list buffer: a launch is ordered behind this one by its event; a command to the resident
        // instance arrives with no kernel boundary in between, so the reset is written THROUGH like the entries (agent-scope store)
        __hip_atomic_store(a.fused_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (xstamp) xstamp[3] = (unsigned long long)wall_clock64();
        // completion word of the tick's slot (pinned host memory), system-scope RELEASE: a host polling it (acquire) sees the record above.
        // (The release is needed as it stands: the record's stores to host memory are held in this XCD's L2 until it is written back --
        //  with a completed-stores wait and a relaxed store of the word instead, the host read the PREVIOUS tick's record, every time.)
        if (t.fused_seq) __hip_atomic_store(t.fused_seq, t.fused_seq_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (xstamp) xstamp[4] = (unsigned long long)wall_clock64();
    }
}

// -DCHIP_NO_ROWS_FORM leaves the row-batched kernel out of the build: it rests on how THIS hipcc maps amdgpu_num_vgpr to an ArchVGPR
// budget (above), which `make verify` checks on the built code object -- a toolchain that allocates differently gets a library without
// the rows form (every scan takes the one-row kernel, same bits, short prefixes ~10-25 % slower) instead of no library at all;
// chip_get_info().scan_forms says which build this is.
#ifndef CHIP_NO_ROWS_FORM
// The body of the row-batched scan: one pass of the hot path over a prefix for the workgroup that runs it.  Two kernels inline it:
// db_scan_topk_rows (one launch per scan) and db_scan_resident (below: a kernel that stays on the chip and runs it once per command).
template <typename T, int NQ, int R, bool NTL, bool RESIDENT = false>
__device__ __forceinline__ void scan_rows_body(const ScanArgs &a, const RowsPass &t, char *smem)
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N, CH = 64 * N, U = 4;
    static_assert(kRowsVgprBase + 16 * R <= 128, "load targets beyond the 128 VGPRs of a 16-waves-per-CU kernel");
    const T *qs = reinterpret_cast<const T *>(smem);  // [NQ][D]
    const int D = a.D;
    const int K = a.K;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpb = blockDim.x >> 6;
    const int64_t tw = (int64_t)gridDim.x * wpb;
    const int64_t g = (int64_t)blockIdx.x * wpb + wave;           // this wave's first row (same row -> wave map as db_scan_topk)
    const int nb = D / (CH * U);                                   // 4 KiB batches per row
    const int64_t rbase = g, rstep = tw, rlim = t.n_rows;        // this wave's rows: rbase + i * rstep < rlim (the same row -> wave map as db_scan_topk)
    const int npass = rbase < rlim ? (int)((rlim - 1 - rbase) / (R * rstep)) + 1 : 0;
    const int total = npass * nb;                                  // batches this wave consumes (wave-uniform)

    unsigned long long *stamp = a.stamps ? a.stamps + (size_t)g * 4 : nullptr;   // tuning only
    if (stamp && lane == 0) stamp[0] = wall_clock64();
    // ---- queries -> LDS by LDS-DMA, 1 KiB per wave-instruction, chunks dealt round robin to the waves ----
    {
        const int cpq = (int)((size_t)D * sizeof(T) / 1024);
        const uint32_t lds0 = lds_byte_addr(smem);
        // the query bases as opaque SGPR values, SELECTED below, not indexed: the resident instance holds the pass in registers, and
        // a select between loads of neighbouring fields would be turned back into an indexed load of a stack copy
        uint64_t qb0 = (uint64_t)(uintptr_t)t.q0, qb1 = (uint64_t)(uintptr_t)t.q1, qb2 = (uint64_t)(uintptr_t)t.q2, qb3 = (uint64_t)(uintptr_t)t.q3;
        asm volatile("" : "+s"(qb0), "+s"(qb1), "+s"(qb2), "+s"(qb3));
        for (int i = wave; i < NQ * cpq; i += wpb) {
            const int q = i / cpq, ch = i - q * cpq;
            uint64_t qp = qb0;
            if (NQ > 1 && q == 1) qp = qb1;
            if (NQ > 2 && q == 2) qp = qb2;
            if (NQ > 3 && q == 3) qp = qb3;
            glds16_q(reinterpret_cast<const char *>((uintptr_t)qp) + (size_t)ch * 1024 + lane * 16, lds0 + (uint32_t)i * 1024u);
        }
    }

    const T *row[R];                                               // bases of the rows the NEXT issue reads (wave-uniform, SGPRs)
    const uint32_t lane_off = (uint32_t)(lane * 16);
    auto set_rows = [&](int pass) {
        const int64_t r0 = rbase + (int64_t)pass * R * rstep;
#pragma unroll
        for (int rr = 0; rr < R; rr++) {
            const int64_t r = r0 + rr * rstep;
            row[rr] = uniform_ptr(row_base_uniform<T>(a, r < rlim ? r : r0));   // a missing row re-reads the pass's first row; its result is dropped
        }
    };
    // load slot u of every row of the batch at byte offset byte_off of the rows
#define CHIP_ROWS_ISSUE_SLOT(u, byte_off)                                                                               \
    do {                                                                                                                \
        const uint32_t vo_ = (byte_off) + lane_off;                                                                     \
        rows_issue<NTL, rows_slot_reg<R>(u, 0), (u) * 1024>(vo_, row[0]);                                               \
        if constexpr (R > 1) rows_issue<NTL, rows_slot_reg<R>(u, R > 1 ? 1 : 0), (u) * 1024>(vo_, row[R > 1 ? 1 : 0]);  \
        if constexpr (R > 2) rows_issue<NTL, rows_slot_reg<R>(u, R > 2 ? 2 : 0), (u) * 1024>(vo_, row[R > 2 ? 2 : 0]);  \
    } while (0)

    if constexpr (R == 1) if (t.dyn_claim && tid == 0)   // the claim counter: units 0 .. wpb-1 are the waves' first rows (set before the barrier below)
        *reinterpret_cast<uint32_t *>(smem + (size_t)NQ * D * sizeof(T) + (size_t)wpb * NQ * CHIP_MAX_TOPK * sizeof(chip_topk_entry)) = (uint32_t)wpb;
    if (total > 0) {
        set_rows(0);
        CHIP_ROWS_ISSUE_SLOT(0, 0u); CHIP_ROWS_ISSUE_SLOT(1, 0u); CHIP_ROWS_ISSUE_SLOT(2, 0u); CHIP_ROWS_ISSUE_SLOT(3, 0u);
        // the LDS-DMA loads are older than the R*U row loads: "at most R*U outstanding" == the queries have landed
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(R * U) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }

#ifdef CHIP_SCAN_TUNING_VARIANTS
    // CHIP_SCAN_STAGGER=n (tuning builds): wave w of a workgroup starts consuming n x 64 x w cycles late -- are the one-workgroup-per-CU
    // shapes slow (6.6-6.9 TB/s) because their waves run in lockstep?
    if constexpr (!RESIDENT) {
        const int st = t.dyn_claim >> 8;
        for (int i = 0; i < st * wave; i++) __builtin_amdgcn_s_sleep(1);
    }
#endif
    if (stamp && lane == 0) stamp[1] = wall_clock64();
    // running top-K lists: LDS behind the queries (this wave's NQ lists are touched by this wave only)
    chip_topk_entry *lists = reinterpret_cast<chip_topk_entry *>(smem + (size_t)NQ * D * sizeof(T));
    chip_topk_entry *mylists = lists + (size_t)wave * NQ * CHIP_MAX_TOPK;
    double thr_s[NQ];
    int64_t thr_i[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        thr_s[q] = -INFINITY; thr_i[q] = -1;
        if (lane < CHIP_MAX_TOPK) { chip_topk_entry t; t.score = -INFINITY; t.idx = -1; mylists[q * CHIP_MAX_TOPK + lane] = t; }
    }

    double acc[R][NQ];
#pragma unroll
    for (int rr = 0; rr < R; rr++)
#pragma unroll
        for (int q = 0; q < NQ; q++) acc[rr][q] = 0.0;

    // Slot u of the current batch: wait for its R loads (CNT0 + R-1-rr operations may stay outstanding behind row rr's), take the
    // data as fp64 and run the fmas of elements base + u*CH + N*lane + c against the staged queries.  Element-major, so one
    // query element is converted once and used by every row; per (row, query) the order is c ascending -- rows_dot's order.
#define CHIP_ROWS_FMA_HALF(u, h, CNT0)                                                                                   \
    do {                                                                                                                \
        double vd_[R][2];                                                                                               \
        rows_take<rows_slot_reg<R>(u, 0), (CNT0) + R - 1, h>(vd_[0], T());                                              \
        if constexpr (R > 1) rows_take<rows_slot_reg<R>(u, R > 1 ? 1 : 0), (CNT0) + R - 2, h>(vd_[R > 1 ? 1 : 0], T()); \
        if constexpr (R > 2) rows_take<rows_slot_reg<R>(u, R > 2 ? 2 : 0), (CNT0) + R - 3, h>(vd_[R > 2 ? 2 : 0], T()); \
        _Pragma("unroll") for (int c = 0; c < N / 2; c++) {                                                             \
            double wd_[NQ];                                                                                             \
            _Pragma("unroll") for (int q = 0; q < NQ; q++) wd_[q] = (double)w_[q][(h) * (N / 2) + c];                   \
            _Pragma("unroll") for (int rr = 0; rr < R; rr++)                                                            \
                _Pragma("unroll") for (int q = 0; q < NQ; q++) acc[rr][q] = __builtin_fma(wd_[q], vd_[rr][c], acc[rr][q]); \
        }                                                                                                               \
    } while (0)
#define CHIP_ROWS_FMA_SLOT(u, base, CNT0)                                                                               \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        const int e_ = (base) + (u) * CH + lane * N;                                                                    \
        V w_[NQ];                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < NQ; q++) w_[q] = *reinterpret_cast<const V *>(qs + q * D + e_);           \
        CHIP_ROWS_FMA_HALF(u, 0, CNT0);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        CHIP_ROWS_FMA_HALF(u, 1, CNT0);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    int pass = 0, b = 0;
    bool dyn = false;
    if constexpr (R == 1) dyn = t.dyn_claim != 0;
    if (dyn) {
      if constexpr (R == 1) {
        // ---- rows CLAIMED within the workgroup (round 5) -------------------------------------------------------------------------
        // Which wave scores which row is free: a row's dot products are one wave's fixed per-lane chain whoever that wave is, and the
        // lists merge under a total order.  The static map (wave g: rows g, g + tw, ...) makes every wave finish its LAST row when
        // its own speed says so -- and the speeds differ by the wave's age on its SIMD (stamps: 61 / 65 / 70 / 74 us at 29k rows).
        // Here the workgroup's rows -- unit u = row (u / wpb) * tw + blockIdx * wpb + u % wpb, the same set as before -- are handed
        // out by an LDS counter: a wave claims its next unit while it starts on the current one (the ds_add_rtn has a whole row,
        // ~8 us, to come back), so the waves of a workgroup run dry within one row of each other.
        uint32_t *ctr = reinterpret_cast<uint32_t *>(smem + (size_t)NQ * D * sizeof(T) + (size_t)wpb * NQ * CHIP_MAX_TOPK * sizeof(chip_topk_entry));
        const int64_t wg0 = (int64_t)blockIdx.x * wpb;
        int n_units = 0;
        if (t.n_rows > wg0) {
            const int64_t span = t.n_rows - wg0, pf = span / tw, rem = span - pf * tw;
            n_units = (int)(pf * wpb + (rem < wpb ? rem : wpb));
        }
        auto unit_row = [&](int u) { return (int64_t)(u / wpb) * tw + wg0 + (u % wpb); };
        // (the counter was set to wpb by wave 0 before the staging barrier; unit `wave` is this wave's first, loaded above)
        int cur = wave;
        bool experimental = false;
#ifdef CHIP_SCAN_TUNING_VARIANTS   // A/B builds only (make EXTRA_HIPFLAGS=-DCHIP_SCAN_TUNING_VARIANTS): the product kernel carries the product stream alone
        // ---- experimental forms of the claimed stream, selected by t.dyn_claim (CHIP_SCAN_DEPTH; the default, 1, is the loop below);
        //      all measured neutral or negative (profiles/r06_short_scan.md §2) ----
        //   2: TWO batches (8 KiB) of every wave in flight, the unit still one claimed row (round 6; profiles/r06_short_scan.md §2):
        //      slots 0..3 (v80..v95) hold the even batches of a row, slots 4..7 (v96..v111) the odd ones; consuming slot s re-issues it for
        //      the batch two ahead, across the row boundary into the next claimed row, so behind every slot exactly 7 younger loads are
        //      outstanding.  Steady state 7.6 -> 8.6 TB/s at 29k rows, but the OLDER workgroup of every CU takes all of the gain.
        //   3 / 4: depth 1 / depth 2 with the slot's re-issue moved IN FRONT of its arithmetic and both at raised issue priority: what
        //      starves the younger workgroup is not memory but the vector pipe's oldest-first arbitration -- a wave re-issues a slot only
        //      after that slot's 12 fp64 fmas, and a young wave's fmas queue behind every older wave's burst, so its load pipeline runs
        //      with gaps.  Here the wait + conversion of a slot (4 v_cvt) and the load that refills it run at s_setprio 3, the fmas at 0.
        //   One accumulator chain per query in ascending element order in every form: same bits.
#define CHIP_ROWSX_ISSUE(slot, chunk, byte_off, rowp) rows_issue<NTL, rows_slot_reg<1>(slot, 0), (chunk) * 1024>((byte_off) + lane_off, rowp)
#define CHIP_ROWSX_STEP(slot, chunk, base, CNT, DO_ISSUE, byte_off, rowp)                                               \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        const int e_ = (base) + (chunk) * CH + lane * N;                                                                \
        V w_[NQ];                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < NQ; q++) w_[q] = *reinterpret_cast<const V *>(qs + q * D + e_);           \
        if constexpr (EARLY) {                                                                                          \
            double va_[2], vb_[2];                                                                                      \
            __builtin_amdgcn_s_setprio(3);                                                                              \
            rows_take<rows_slot_reg<1>(slot, 0), CNT, 0>(va_, T());                                                     \
            rows_take<rows_slot_reg<1>(slot, 0), CNT, 1>(vb_, T());                                                     \
            if (DO_ISSUE) CHIP_ROWSX_ISSUE(slot, chunk, byte_off, rowp);                                                \
            __builtin_amdgcn_s_setprio(0);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            _Pragma("unroll") for (int c = 0; c < N / 2; c++)                                                           \
                _Pragma("unroll") for (int q = 0; q < NQ; q++) acc[0][q] = __builtin_fma((double)w_[q][c], va_[c], acc[0][q]); \
            _Pragma("unroll") for (int c = 0; c < N / 2; c++)                                                           \
                _Pragma("unroll") for (int q = 0; q < NQ; q++) acc[0][q] = __builtin_fma((double)w_[q][N / 2 + c], vb_[c], acc[0][q]); \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        } else {                                                                                                        \
            _Pragma("unroll") for (int h_ = 0; h_ < 2; h_++) {                                                          \
                double vd_[2];                                                                                          \
                if (h_ == 0) rows_take<rows_slot_reg<1>(slot, 0), CNT, 0>(vd_, T());                                    \
                else rows_take<rows_slot_reg<1>(slot, 0), CNT, 1>(vd_, T());                                            \
                _Pragma("unroll") for (int c = 0; c < N / 2; c++)                                                       \
                    _Pragma("unroll") for (int q = 0; q < NQ; q++)                                                      \
                        acc[0][q] = __builtin_fma((double)w_[q][h_ * (N / 2) + c], vd_[c], acc[0][q]);                  \
                __builtin_amdgcn_sched_barrier(0);                                                                      \
            }                                                                                                           \
            if (DO_ISSUE) CHIP_ROWSX_ISSUE(slot, chunk, byte_off, rowp);                                                \
        }                                                                                                               \
    } while (0)
        auto finish_row = [&](int unit) {
            const int64_t r = unit_row(unit);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const double s = butterfly_sum(acc[0][q]);
                acc[0][q] = 0.0;
                wave_topk_offer_lds(s, r * a.idx_mul + a.idx_add, K, lane, mylists + q * CHIP_MAX_TOPK, thr_s[q], thr_i[q]);
            }
        };
        auto stream_depth2 = [&](auto early_tag) {
            constexpr bool EARLY = decltype(early_tag)::value;
            const T *rowp = row[0], *nrow = row[0];
            if (cur < n_units) {   // batch 1 of the first row (batch 0 went out in front of the staging barrier)
                CHIP_ROWSX_ISSUE(4, 0, 4096u, rowp); CHIP_ROWSX_ISSUE(5, 1, 4096u, rowp); CHIP_ROWSX_ISSUE(6, 2, 4096u, rowp); CHIP_ROWSX_ISSUE(7, 3, 4096u, rowp);
            }
            while (cur < n_units) {
                uint32_t nxt_v = 0;
                if (lane == 0) nxt_v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int nxt = 0;
                bool more = false;
                for (int bb = 0; bb < nb; bb += 2) {
                    const int base0 = bb * (CH * U), base1 = base0 + CH * U;
                    const T *ra = rowp;
                    uint32_t oa = (uint32_t)(bb + 2) * 4096u;
                    bool ahead = true;
                    if (bb + 2 == nb) {                      // the two batches ahead are the next claimed row's first two
                        nxt = __builtin_amdgcn_readfirstlane((int)nxt_v);
                        more = nxt < n_units;
                        ahead = more;
                        oa = 0u;
                        if (more) { nrow = uniform_ptr(row_base_uniform<T>(a, unit_row(nxt))); ra = nrow; }
                    }
                    if (ahead) {
                        CHIP_ROWSX_STEP(0, 0, base0, 7, 1, oa, ra); CHIP_ROWSX_STEP(1, 1, base0, 7, 1, oa, ra);
                        CHIP_ROWSX_STEP(2, 2, base0, 7, 1, oa, ra); CHIP_ROWSX_STEP(3, 3, base0, 7, 1, oa, ra);
                        CHIP_ROWSX_STEP(4, 0, base1, 7, 1, oa + 4096u, ra); CHIP_ROWSX_STEP(5, 1, base1, 7, 1, oa + 4096u, ra);
                        CHIP_ROWSX_STEP(6, 2, base1, 7, 1, oa + 4096u, ra); CHIP_ROWSX_STEP(7, 3, base1, 7, 1, oa + 4096u, ra);
                    } else {                                 // the wave's last two batches: nothing is re-issued
                        CHIP_ROWSX_STEP(0, 0, base0, 7, 0, 0u, ra); CHIP_ROWSX_STEP(1, 1, base0, 6, 0, 0u, ra);
                        CHIP_ROWSX_STEP(2, 2, base0, 5, 0, 0u, ra); CHIP_ROWSX_STEP(3, 3, base0, 4, 0, 0u, ra);
                        CHIP_ROWSX_STEP(4, 0, base1, 3, 0, 0u, ra); CHIP_ROWSX_STEP(5, 1, base1, 2, 0, 0u, ra);
                        CHIP_ROWSX_STEP(6, 2, base1, 1, 0, 0u, ra); CHIP_ROWSX_STEP(7, 3, base1, 0, 0, 0u, ra);
                    }
                }
                finish_row(cur);
                cur = more ? nxt : n_units;
                rowp = nrow;
            }
        };
        auto stream_depth1 = [&](auto early_tag) {
            constexpr bool EARLY = decltype(early_tag)::value;
            const T *rowp = row[0];
            while (cur < n_units) {
                uint32_t nxt_v = 0;
                if (lane == 0) nxt_v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                int nxt = 0;
                bool more = false;
                for (int bb = 0; bb < nb; bb++) {
                    const int base0 = bb * (CH * U);
                    const T *ra = rowp;
                    uint32_t oa = (uint32_t)(bb + 1) * 4096u;
                    bool ahead = true;
                    if (bb + 1 == nb) {
                        nxt = __builtin_amdgcn_readfirstlane((int)nxt_v);
                        more = nxt < n_units;
                        ahead = more;
                        oa = 0u;
                        if (more) { rowp = uniform_ptr(row_base_uniform<T>(a, unit_row(nxt))); ra = rowp; }
                    }
                    if (ahead) {
                        CHIP_ROWSX_STEP(0, 0, base0, 3, 1, oa, ra); CHIP_ROWSX_STEP(1, 1, base0, 3, 1, oa, ra);
                        CHIP_ROWSX_STEP(2, 2, base0, 3, 1, oa, ra); CHIP_ROWSX_STEP(3, 3, base0, 3, 1, oa, ra);
                    } else {
                        CHIP_ROWSX_STEP(0, 0, base0, 3, 0, 0u, ra); CHIP_ROWSX_STEP(1, 1, base0, 2, 0, 0u, ra);
                        CHIP_ROWSX_STEP(2, 2, base0, 1, 0, 0u, ra); CHIP_ROWSX_STEP(3, 3, base0, 0, 0, 0u, ra);
                    }
                }
                finish_row(cur);
                cur = more ? nxt : n_units;
            }
        };
        //   7: depth 2 with the rows claimed by PAIRS of workgroups -- b and b + gridDim / 2, i.e. (round-robin dispatch) the older and the
        //      younger workgroup of a CU, on the same XCD -- from ONE counter in device memory: the older workgroup, which the memory path
        //      prefers, simply takes more rows, so both run dry together.  The claim is a global atomic with return, issued from asm at the
        //      START of a row into v112 (an asm-owned register depth 2 leaves free) and read two batches later: it sits in the in-order
        //      vector-memory queue, so the waits of the row's first two batches allow ONE more outstanding operation (vmcnt 8 instead of 7)
        //      and by the time the third batch is waited for it has returned -- no wait of its own.  Pair unit p -> row: ascending in p
        //      (round, workgroup b's wpb rows, workgroup b + G/2's wpb rows), so "p names a row" is a prefix property.
        auto stream_pair2 = [&]() {
            constexpr bool EARLY = false;
            const int G = (int)gridDim.x, Hh = G >> 1;
            const int b1 = (int)blockIdx.x < Hh ? (int)blockIdx.x : (int)blockIdx.x - Hh, b2 = b1 + Hh;
            const int second = (int)blockIdx.x >= Hh ? 1 : 0;
            uint32_t *pctr = a.pair_ctr + (size_t)b1 * 32;   // a 128-byte line per pair (Ctx::kPairStride)
            auto units_of = [&](int b) {
                const int64_t w0 = (int64_t)b * wpb;
                if (t.n_rows <= w0) return 0;
                const int64_t span = t.n_rows - w0, pf = span / tw, rem = span - pf * tw;
                return (int)(pf * wpb + (rem < wpb ? rem : wpb));
            };
            const int n_pair = units_of(b1) + units_of(b2);
            auto prow = [&](int p) {
                const int round = p / (2 * wpb), o = p - round * 2 * wpb;
                return (int64_t)round * tw + (o < wpb ? (int64_t)b1 * wpb + o : (int64_t)b2 * wpb + (o - wpb));
            };
            auto finish_prow = [&](int p) {
                const int64_t r = prow(p);
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const double s = butterfly_sum(acc[0][q]);
                    acc[0][q] = 0.0;
                    wave_topk_offer_lds(s, r * a.idx_mul + a.idx_add, K, lane, mylists + q * CHIP_MAX_TOPK, thr_s[q], thr_i[q]);
                }
            };
            const uint64_t pc = (uint64_t)(uintptr_t)pctr;
            const uint32_t one = 1u;
            int pcur = second * wpb + wave;          // this wave's first pair unit = the row whose first batch is already in flight
            const T *rowp = row[0], *nrow = row[0];
            if (pcur < n_pair) {
                CHIP_ROWSX_ISSUE(4, 0, 4096u, rowp); CHIP_ROWSX_ISSUE(5, 1, 4096u, rowp); CHIP_ROWSX_ISSUE(6, 2, 4096u, rowp); CHIP_ROWSX_ISSUE(7, 3, 4096u, rowp);
            }
            while (pcur < n_pair) {
                // the claim of the NEXT row: lane 0 alone, result in v112 (pre-add value; units 0 .. 2 wpb - 1 are the waves' first rows).
                // Agent scope (sc1): the counter is reset by the launch's LAST workgroup, which may sit on another XCD -- an atomic resolved
                // in this XCD's L2 alone would leave a dirty line there for the end-of-kernel write-back to put on top of that reset
                asm volatile("s_mov_b64 s[72:73], exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add v112, %0, %1, off sc0 sc1\n\ts_mov_b64 exec, s[72:73]"
                             ::"v"(pc), "v"(one) : "memory", "s72", "s73");
                int nxt = 0;
                bool more = false;
                for (int bb = 0; bb < nb; bb += 2) {
                    const int base0 = bb * (CH * U), base1 = base0 + CH * U;
                    const T *ra = rowp;
                    uint32_t oa = (uint32_t)(bb + 2) * 4096u;
                    bool ahead = true;
                    if (bb + 2 == nb) {
                        // (nb >= 4: the claim was issued two batches ago; the wait the next take would do anyway also covers it)
                        int got;
                        asm volatile("s_waitcnt vmcnt(7)\n\tv_readfirstlane_b32 %0, v112" : "=s"(got)::"memory");
                        nxt = got + 2 * wpb;
                        more = nxt < n_pair;
                        ahead = more;
                        oa = 0u;
                        if (more) { nrow = uniform_ptr(row_base_uniform<T>(a, prow(nxt))); ra = nrow; }
                    }
                    if (bb == 0) {       // the claim is younger than these two batches' loads: one more operation may stay outstanding
                        CHIP_ROWSX_STEP(0, 0, base0, 8, 1, oa, ra); CHIP_ROWSX_STEP(1, 1, base0, 8, 1, oa, ra);
                        CHIP_ROWSX_STEP(2, 2, base0, 8, 1, oa, ra); CHIP_ROWSX_STEP(3, 3, base0, 8, 1, oa, ra);
                        CHIP_ROWSX_STEP(4, 0, base1, 8, 1, oa + 4096u, ra); CHIP_ROWSX_STEP(5, 1, base1, 8, 1, oa + 4096u, ra);
                        CHIP_ROWSX_STEP(6, 2, base1, 8, 1, oa + 4096u, ra); CHIP_ROWSX_STEP(7, 3, base1, 8, 1, oa + 4096u, ra);
                    } else if (ahead) {
                        CHIP_ROWSX_STEP(0, 0, base0, 7, 1, oa, ra); CHIP_ROWSX_STEP(1, 1, base0, 7, 1, oa, ra);
                        CHIP_ROWSX_STEP(2, 2, base0, 7, 1, oa, ra); CHIP_ROWSX_STEP(3, 3, base0, 7, 1, oa, ra);
                        CHIP_ROWSX_STEP(4, 0, base1, 7, 1, oa + 4096u, ra); CHIP_ROWSX_STEP(5, 1, base1, 7, 1, oa + 4096u, ra);
                        CHIP_ROWSX_STEP(6, 2, base1, 7, 1, oa + 4096u, ra); CHIP_ROWSX_STEP(7, 3, base1, 7, 1, oa + 4096u, ra);
                    } else {
                        CHIP_ROWSX_STEP(0, 0, base0, 7, 0, 0u, ra); CHIP_ROWSX_STEP(1, 1, base0, 6, 0, 0u, ra);
                        CHIP_ROWSX_STEP(2, 2, base0, 5, 0, 0u, ra); CHIP_ROWSX_STEP(3, 3, base0, 4, 0, 0u, ra);
                        CHIP_ROWSX_STEP(4, 0, base1, 3, 0, 0u, ra); CHIP_ROWSX_STEP(5, 1, base1, 2, 0, 0u, ra);
                        CHIP_ROWSX_STEP(6, 2, base1, 1, 0, 0u, ra); CHIP_ROWSX_STEP(7, 3, base1, 0, 0, 0u, ra);
                    }
                }
                finish_prow(pcur);
                pcur = more ? nxt : n_pair;
                rowp = nrow;
            }
        };
        // (the resident instance runs the product form only: its register budget is the tightest of the three kernels that inline this body)
        if constexpr (!RESIDENT) {
            const int form = t.dyn_claim & 255;
            experimental = form >= 2;
            if (form == 7 && (nb & 1) == 0 && nb >= 4 && a.pair_ctr != nullptr) stream_pair2();
            else if (form == 2 && (nb & 1) == 0) stream_depth2(std::false_type{});
            else if (form == 4 && (nb & 1) == 0) stream_depth2(std::true_type{});
            else if (form == 3 || form == 4) stream_depth1(std::true_type{});
            else experimental = false;
        }
#endif
        if (!experimental)
        while (cur < n_units) {
            uint32_t nxt_v = 0;                                 // next unit, claimed by lane 0 alone: needed at this row's last batch
            if (lane == 0) nxt_v = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int nxt = 0;
            bool more = false;
            for (b = 0; b < nb; b++) {
                const int base = b * (CH * U);
                bool next_batch = true;
                int nbase = base + CH * U;
                if (b + 1 == nb) {
                    nxt = __builtin_amdgcn_readfirstlane((int)nxt_v);
                    more = nxt < n_units;
                    next_batch = more;
                    nbase = 0;
                    if (more) row[0] = uniform_ptr(row_base_uniform<T>(a, unit_row(nxt)));
                }
                if (next_batch) {
                    const uint32_t noff = (uint32_t)nbase * (uint32_t)sizeof(T);
                    CHIP_ROWS_FMA_SLOT(0, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(0, noff);
                    CHIP_ROWS_FMA_SLOT(1, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(1, noff);
                    CHIP_ROWS_FMA_SLOT(2, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(2, noff);
                    CHIP_ROWS_FMA_SLOT(3, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(3, noff);
                } else {
                    CHIP_ROWS_FMA_SLOT(0, base, 3 * R);
                    CHIP_ROWS_FMA_SLOT(1, base, 2 * R);
                    CHIP_ROWS_FMA_SLOT(2, base, 1 * R);
                    CHIP_ROWS_FMA_SLOT(3, base, 0);
                }
            }
            const int64_t r = unit_row(cur);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const double s = butterfly_sum(acc[0][q]);
                acc[0][q] = 0.0;
                wave_topk_offer_lds(s, r * a.idx_mul + a.idx_add, K, lane, mylists + q * CHIP_MAX_TOPK, thr_s[q], thr_i[q]);
            }
            cur = more ? nxt : n_units;
        }
#ifdef CHIP_SCAN_TUNING_VARIANTS
#undef CHIP_ROWSX_STEP
#undef CHIP_ROWSX_ISSUE
#endif
      }
    } else
    for (int t = 0; t < total; t++) {
        const int base = b * (CH * U);
        if (t + 1 < total) {
            // steady state: slot (u, rr) has (U-1)*R + (R-1-rr) younger loads behind it whatever u is (the slots of the next
            // batch issued so far take the place of the slots of this batch already consumed)
            int nbase = base + CH * U;
            if (b + 1 == nb) { nbase = 0; set_rows(pass + 1); }
            const uint32_t noff = (uint32_t)nbase * (uint32_t)sizeof(T);
            CHIP_ROWS_FMA_SLOT(0, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(0, noff);
            CHIP_ROWS_FMA_SLOT(1, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(1, noff);
            CHIP_ROWS_FMA_SLOT(2, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(2, noff);
            CHIP_ROWS_FMA_SLOT(3, base, (U - 1) * R); CHIP_ROWS_ISSUE_SLOT(3, noff);
        } else {
            // the wave's last batch: nothing is re-issued, the loads behind slot (u, rr) are the rest of this batch
            CHIP_ROWS_FMA_SLOT(0, base, 3 * R);
            CHIP_ROWS_FMA_SLOT(1, base, 2 * R);
            CHIP_ROWS_FMA_SLOT(2, base, 1 * R);
            CHIP_ROWS_FMA_SLOT(3, base, 0);
        }
        if (++b == nb) {   // the rows of this pass are complete: butterfly, offer, next pass
            const int64_t r0 = rbase + (int64_t)pass * R * rstep;
#pragma unroll
            for (int rr = 0; rr < R; rr++) {
                const int64_t r = r0 + rr * rstep;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const double s = butterfly_sum(acc[rr][q]);
                    acc[rr][q] = 0.0;
                    if (r < rlim) wave_topk_offer_lds(s, r * a.idx_mul + a.idx_add, K, lane, mylists + q * CHIP_MAX_TOPK, thr_s[q], thr_i[q]);
                }
            }
            b = 0;
            pass++;
        }
    }
#undef CHIP_ROWS_FMA_SLOT
#undef CHIP_ROWS_FMA_HALF
#undef CHIP_ROWS_ISSUE_SLOT
    if (stamp && lane == 0) stamp[2] = wall_clock64();
    block_merge_lists<NQ>(a, lists, K, lane, wave, wpb, t.fused_result != nullptr);
    if (t.fused_result != nullptr) fused_tick_finish<NQ, R == 1, RESIDENT>(a, t, smem, K, tid, lane, wave, wpb);
    if (stamp && lane == 0) stamp[3] = wall_clock64();
}

template <typename T, int NQ, int R, bool NTL>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(kRowsVgprBase / 2))) void db_scan_topk_rows(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: CHIP_ROWS_CLOBBERS);                       // makes the code object allocate the asm-owned registers
    scan_rows_body<T, NQ, R, NTL>(a, rows_pass_of(a), smem);
}

template <typename T, int NQ, int R, bool NTL>
static int launch_scan_rows(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
    if (lds > 65536) CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_topk_rows<T, NQ, R, NTL>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((db_scan_topk_rows<T, NQ, R, NTL>), dim3(grid), dim3(block), lds, s, a);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

// ---- the resident instance of the row-batched scan (opt-in, chip_internal.h ResidentCmd / ResidentArgs) ----
// One workgroup per CU stays on the chip and runs scan_rows_body<T, 3, 1, temporal> once per command -- the fused tick with nothing
// launched: no packet, no dispatch, no ramp of workgroup starts.  Wave 0 of workgroup 0 polls the host's 64-byte command line
// (system-scope loads over PCIe, 16 lanes = the whole line in one request; a line is accepted when head == tail) and copies it into
// every other workgroup's OWN line in device memory (one 64-byte write-through store each); wave 0 of the others polls its line
// (agent-scope loads: 256 pollers on 256 different lines, no hot spot).  The command reaches the waves of a workgroup through LDS.
// What the instance may read without a cache invalidation per command: published DB rows are immutable and a row is first read after
// it has been published, so no cache ever holds an older copy of one; the SEGMENT TABLE does change when the DB opens a new segment,
// so the host retires the instance before it touches the table (chip_api.hip resident_stop) and the next tick launches a new one.
// Every wait has a way out: workgroup 0 leaves (and sends the others home) after lease_ticks without a command, the others after four
// times that without a go word, so a host that died -- or a workgroup that never became resident -- cannot hold the chip.
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); }
__device__ __forceinline__ unsigned long long uni_clock() { return uni64((uint64_t)wall_clock64()); }   // (the compiler takes the clock for a per-lane value)

template <typename T>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(kRowsVgprBase / 2))) void db_scan_resident(ResidentArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t cmd_s[16];
    asm volatile("" ::: CHIP_ROWS_CLOBBERS);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t done = ra.done;
    unsigned long long t_last = uni_clock();
    unsigned long long *xstamp = ra.base.stamps ? ra.base.stamps + (size_t)gridDim.x * (blockDim.x >> 6) * 4 : nullptr;   // tuning only
    for (;;) {
        if (wave == 0) {
            // lanes 16 k .. 16 k + 15 hold the 16 words of a line; a line is accepted when head == tail, is not the last one run, is not 0
            uint32_t v = 0, head = 0;
            bool leave = false;
            auto fresh = [&](uint32_t x) {
                head = (uint32_t)__builtin_amdgcn_readlane((int)x, 0);
                return head == (uint32_t)__builtin_amdgcn_readlane((int)x, 15) && head != done && head != 0u;
            };
            if (blockIdx.x == 0) {
                // The host's line, over PCIe: FOUR loads in flight, a quarter of a round trip apart (each is re-issued when it returns, so
                // the spacing holds) -- a command waits for the next load to leave, then for that load's round trip: 1.125 instead of 1.5.
                const uint32_t *line = ra.cmd_host + (lane & 15);
#define CHIP_POLL_HOST() __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define CHIP_POLL_STEP(P)                                                         \
    if (fresh(P)) { v = P; break; }                                               \
    if (uni_clock() - t_last > ra.lease_ticks) { leave = true; break; }           \
    P = CHIP_POLL_HOST();
                uint32_t p0 = CHIP_POLL_HOST();
                __builtin_amdgcn_s_sleep(10);
                uint32_t p1 = CHIP_POLL_HOST();
                __builtin_amdgcn_s_sleep(10);
                uint32_t p2 = CHIP_POLL_HOST();
                __builtin_amdgcn_s_sleep(10);
                uint32_t p3 = CHIP_POLL_HOST();
                for (;;) {
                    CHIP_POLL_STEP(p0)
                    CHIP_POLL_STEP(p1)
                    CHIP_POLL_STEP(p2)
                    CHIP_POLL_STEP(p3)
                }
#undef CHIP_POLL_STEP
#undef CHIP_POLL_HOST
            } else {
                const uint32_t *line = ra.cmd_dev + (size_t)blockIdx.x * 16 + (lane & 15);
                for (;;) {
                    v = __hip_atomic_load(line, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (written by workgroup 0, or by the host through the BAR)
                    if (fresh(v)) break;
                    if (uni_clock() - t_last > 4 * ra.lease_ticks) { leave = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (!leave) leave = __builtin_amdgcn_readlane((int)v, 3) < 0;   // n_rows < 0 (words 2, 3): the host retires the instance
            if (leave) {   // the line every workgroup acts on: n_rows = -1 under the leave mark
                v = (lane & 15) == 0 || (lane & 15) == 15 ? kResidentLeave : ((lane & 15) == 2 || (lane & 15) == 3 ? 0xffffffffu : 0u);
            }
            if (xstamp && blockIdx.x == 0 && lane == 0 && !leave) xstamp[0] = (unsigned long long)wall_clock64();
            if (lane < 16) cmd_s[lane] = v;
        }
        __syncthreads();
        if (blockIdx.x == 0 && (!ra.direct || cmd_s[0] == kResidentLeave)) {
            // the relay, by ALL waves of workgroup 0 (one wave alone issues a write-through store instruction every ~60 ns: 3.9 us for
            // 64 of them, measured): one 64-byte store per workgroup, four workgroups per instruction -- a line arrives whole
            const uint32_t w = cmd_s[lane & 15];
            for (int j = tid >> 4; j < (int)gridDim.x; j += (int)(blockDim.x >> 4))
                if (j > 0) __hip_atomic_store(ra.cmd_dev + (size_t)j * 16 + (lane & 15), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (xstamp && tid == 0 && cmd_s[0] != kResidentLeave) xstamp[1] = (unsigned long long)wall_clock64();
        }
        const ResidentCmd *cm = reinterpret_cast<const ResidentCmd *>(cmd_s);
        const int64_t n_rows = (int64_t)uni64((uint64_t)cm->n_rows);
        if (n_rows < 0) {
            if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(ra.exit_host, ra.instance, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        const ScanArgs &a = ra.base;      // K = 1 and the fused-tick fields that do not change (ticket, list buffer) are set by the host
        RowsPass t;
        t.n_rows = n_rows;
        t.tick_l = (int64_t)uni64((uint64_t)cm->tick_l);
        t.locality = (int32_t)uni32((uint32_t)cm->locality);
        t.thresh = __longlong_as_double((long long)uni64((uint64_t)__double_as_longlong(cm->thresh)));
        t.fused_result = reinterpret_cast<chip_tick_result *>((uintptr_t)uni64(cm->result));
        t.fused_seq = reinterpret_cast<unsigned long long *>((uintptr_t)uni64(cm->seq_ptr));
        t.fused_seq_val = uni64(cm->seq_val);
        t.dyn_claim = (int32_t)uni32(cm->dyn_claim);
        const uint32_t number = uni32(cm->head);
        t.q0 = row_base_uniform<T>(a, t.tick_l - 1);   // v, vm, vmm (Cerebro.cpp:987-989)
        t.q1 = row_base_uniform<T>(a, t.tick_l - 2);
        t.q2 = row_base_uniform<T>(a, t.tick_l - 3);
        t.q3 = nullptr;
        scan_rows_body<T, 3, 1, false, true>(a, t, smem);
        done = number;
        t_last = uni_clock();
        __syncthreads();   // cmd_s and the lists are free again
    }
}

template <typename T>
static int launch_resident_T(Ctx *c, hipStream_t s, const ResidentArgs &ra, int grid, size_t lds, int block)
{
    if (lds > 65536) CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_resident<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((db_scan_resident<T>), dim3(grid), dim3(block), lds, s, ra);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}
#endif  // CHIP_NO_ROWS_FORM

template <typename T, int NQ, int U, bool FULL, int NT, int R>
static int launch_scan_k(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
    if (lds > 65536) CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_topk<T, NQ, U, FULL, NT, R>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((db_scan_topk<T, NQ, U, FULL, NT, R>), dim3(grid), dim3(block), lds, s, a);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

template <typename T, int NQ, int U, int NT, int R>
static int launch_scan_t(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
    return a.D % (64 * Vec16<T>::N * U) == 0 ? launch_scan_k<T, NQ, U, true, NT, R>(c, s, a, grid, lds, block)
                                             : launch_scan_k<T, NQ, U, false, NT, R>(c, s, a, grid, lds, block);
}

// Load path of K1.  Production (scan_variant 0): 4 x 16-B non-temporal loads per lane issued back to back from inline asm and
// consumed behind counted waits (rows_dot, NT == 6) -- with compiler-scheduled loads (variant 1, the round-1 kernel: U = 8
// builtin loads) hipcc sinks each load next to its use and a wave runs with 1-2 KiB in flight at 110 VGPRs; the asm form
// keeps 4 KiB per wave in flight at 66 VGPRs (short scans +5-7 %, 1M rows equal: profiles/r02_scan_load_path.txt).  Rows
// of whole 2 KiB batches use 2 loads per batch (D = 1536: +6 %); anything else (D * elem % 2048 != 0) takes the builtin path.
// CHIP_SCAN_VARIANT >= 2: further A/B variants, only in builds with -DCHIP_SCAN_TUNING_VARIANTS.
template <typename T, int NQ>
static int launch_scan_q(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
#ifdef CHIP_SCAN_TUNING_VARIANTS
    switch (c->scan_variant) {
        case 2: return launch_scan_t<T, NQ, 16, 1, 1>(c, s, a, grid, lds, block);
        case 3: return launch_scan_t<T, NQ, 8, 0, 1>(c, s, a, grid, lds, block);
        case 4: return launch_scan_t<T, NQ, 4, 1, 1>(c, s, a, grid, lds, block);
        case 7: return launch_scan_t<T, NQ, 8, 2, 1>(c, s, a, grid, lds, block);
        case 8: return launch_scan_t<T, NQ, 8, 3, 1>(c, s, a, grid, lds, block);
        case 9: return launch_scan_t<T, NQ, 8, 4, 1>(c, s, a, grid, lds, block);
        case 10: return launch_scan_t<T, NQ, 8, 5, 1>(c, s, a, grid, lds, block);
        case 11: if ((int64_t)a.D * sizeof(T) % 8192 == 0) return launch_scan_k<T, NQ, 8, true, 6, 1>(c, s, a, grid, lds, block); break;
        default: break;
    }
#endif
    if constexpr (sizeof(T) == 4) {
        if (a.q64) return launch_scan_k<T, NQ, 4, true, 8, 1>(c, s, a, grid, lds, block);   // scan_q64() said so (float rows only)
    }
#ifndef CHIP_NO_ROWS_FORM
    if (a.rows_form > 0) {   // scan_rows_form() said so: whole 4 KiB batches, R rows per wave in one continuous load stream
        const bool ntl = a.plain_loads == 0;
        if constexpr (NQ <= 3) {   // (four queries x R > 1 rows of accumulators do not fit the compiler's 80 registers: R = 1 only)
            if (a.rows_form >= 3) return ntl ? launch_scan_rows<T, NQ, 3, true>(c, s, a, grid, lds, block) : launch_scan_rows<T, NQ, 3, false>(c, s, a, grid, lds, block);
            if (a.rows_form == 2) return ntl ? launch_scan_rows<T, NQ, 2, true>(c, s, a, grid, lds, block) : launch_scan_rows<T, NQ, 2, false>(c, s, a, grid, lds, block);
        }
        return ntl ? launch_scan_rows<T, NQ, 1, true>(c, s, a, grid, lds, block) : launch_scan_rows<T, NQ, 1, false>(c, s, a, grid, lds, block);
    }
#endif
    if (c->scan_variant != 1) {   // rows of whole 4 KiB / 2 KiB batches (one load per batch measured slower than the builtin path)
        const int64_t row_bytes = (int64_t)a.D * sizeof(T);
        if (row_bytes % 4096 == 0) return launch_scan_k<T, NQ, 4, true, 6, 1>(c, s, a, grid, lds, block);
        if (row_bytes % 2048 == 0) return launch_scan_k<T, NQ, 2, true, 6, 1>(c, s, a, grid, lds, block);
    }
    return launch_scan_t<T, NQ, 8, 1, 1>(c, s, a, grid, lds, block);
}

// Workgroup shape of K1.  The nq query descriptors sit in LDS (nq*D*elem bytes per workgroup), so the shape follows D:
// 2 workgroups x 512 threads per CU while two copies fit in the 160 KiB (D = 4096 fp32: 48 KiB each), else 1 x 1024 threads
// (D = 8192 fp32, the reference's default model, or D = 4096 fp64: 96 KiB) -- the same 16 waves per CU either way
// (measured: 6.7 TB/s vs 5.5 with 512 x 1).  CHIP_SCAN_BLOCK / CHIP_SCAN_BPC override (tuning only).
// Double rows: how many of the nq queries are NOT staged in LDS but read in place (db_scan_topk_wide); 0 = all of them fit.
int scan_wide_ng(const Ctx *c, int nq)
{
    if (c->elem != 8 || (size_t)nq * c->D * 8 <= 160 * 1024) return 0;
    const int fit = (int)((160 * 1024) / ((size_t)c->D * 8));
    return nq - (fit < 1 ? 1 : fit);
}

static size_t scan_lds_bytes(const Ctx *c, int nq, int K, int block, bool q64, bool rows_form = false)
{
    if (rows_form) return (size_t)nq * c->D * c->elem + (size_t)(block / 64) * nq * CHIP_MAX_TOPK * sizeof(chip_topk_entry) + 16;   // queries + running lists + the claim counter
    const size_t lds_q = (size_t)(q64 ? nq : nq - scan_wide_ng(c, nq)) * c->D * (q64 ? 8 : c->elem);   // the staged queries
    const size_t lds_m = (size_t)(block / 64) * nq * K * sizeof(chip_topk_entry);
    return lds_q > lds_m ? lds_q : lds_m;
}

static void scan_shape(const Ctx *c, int nq, bool q64, int *block, int *bpc)
{
    if (c->scan_block > 0) { *block = c->scan_block; *bpc = c->scan_blocks_per_cu; return; }
    const bool two_fit = 2 * (scan_lds_bytes(c, nq, CHIP_MAX_TOPK, 512, q64) + 1024) <= 160 * 1024;
    *block = two_fit ? 512 : 1024;
    *bpc = two_fit ? 2 : 1;
}

// Queries staged in LDS as fp64 (NT == 8: no per-row conversion of the query side, 21 instead of 34 VALU instructions per KiB
// of DB) for LONG scans of float rows: 8 bytes per query element make it one 1024-thread workgroup per CU, whose per-launch
// cost (96 KiB of staging per workgroup) only pays off on launches long enough to run alone on the scan stream -- measured
// +1.0-1.1 % at 750k / 1M rows, -0.5 .. -6 % at 500k .. 100k rows.  Same arithmetic, same bits.  CHIP_SCAN_VARIANT=1 / 7 disable it.
bool scan_q64(const Ctx *c, int nq, bool long_scan)
{
    return long_scan && c->elem == 4 && c->scan_variant != 1 && c->scan_variant != 7 && (int64_t)c->D * 4 % 4096 == 0 &&
           (size_t)nq * c->D * 8 <= 150 * 1024;
}

// Row-batched form of K1 (db_scan_topk_rows): R rows per wave in flight.  Returns R (1..3), or 0 for the one-row kernel.
// CHIP_SCAN_ROWS forces R for every scan (1..3) or disables the form (-1).
int scan_forms_built()
{
#ifdef CHIP_NO_ROWS_FORM
    return CHIP_SCAN_FORM_ONE_ROW;
#else
    return CHIP_SCAN_FORM_ONE_ROW | CHIP_SCAN_FORM_ROWS;
#endif
}

int scan_rows_form(const Ctx *c, int64_t n_rows, int nq, int grid, bool q64, bool sync_tick)
{
#ifdef CHIP_NO_ROWS_FORM
    (void)c; (void)n_rows; (void)nq; (void)grid; (void)q64; (void)sync_tick;
    return 0;
#endif
    if (q64 || c->scan_rows < 0 || c->scan_variant == 1 || (int64_t)c->D * c->elem % 4096 != 0) return 0;
    int block, bpc;
    scan_shape(c, nq, false, &block, &bpc);
    if ((size_t)bpc * scan_lds_bytes(c, nq, CHIP_MAX_TOPK, block, false, true) > 160 * 1024) return 0;   // queries + LDS lists must fit
    const int64_t waves = (int64_t)grid * (block / 64);
    const int64_t rpw = (n_rows + waves - 1) / waves;
    const int rmax = nq >= 4 ? 1 : 3;
    if (c->scan_rows > 0) return c->scan_rows > rmax ? rmax : c->scan_rows;
    (void)rpw;
    // Default policy, from A/B runs on one box (us per tick, 4096-D; profiles/r03_short_scan_ab_2.txt, r03_mid_scan_ab.txt):
    //   10k rows (164 MB): rows form R = 1 / 2 / 3 at half occupancy 25.1 / 26.5 / 27.7, one-row kernel 30.0;
    //   29k rows (475 MB, the reference's own capacity, Cerebro.cpp:946): rows form R = 1 with temporal loads 66.4, one-row kernel
    //   71.7, rows form with non-temporal loads R = 1 / 2 / 3 75.6 / 75.1 / 83.8;   60k rows (983 MB): 142.1 vs 141.3;
    //   100k / 1M rows: one-row kernel 232 / 2355 vs 245 / 2353.
    // => prefixes up to scan_plain_bytes (768 MiB: a good part of them survives in the 256 MiB Infinity Cache from tick to tick) take
    //    the rows form with R = 1 and temporal loads; longer ones the one-row kernel with non-temporal loads.
    // Round 6, beyond 768 MiB (non-temporal loads either way; us: isolated kernel / pipelined step / SYNCHRONOUS tick, two boxes,
    // profiles/r06_scan_rows_policy_ab.txt, r06_shape_ab_8192.txt):
    //   4096-D x  60k (0.98 GB): one-row 167 / 155 / 188   rows R = 1 165 / 149 / 162   R = 2 169 / 149 / 171
    //   4096-D x 100k (1.64 GB): one-row 257 / 245 / 282   rows R = 1 257 / 268 / 259   R = 2 264 / 256 / 274
    //   8192-D x  29k (0.95 GB, the reference's default model at its own capacity, Cerebro.cpp:946,1021):
    //                            one-row 180 / 166 / 207   rows R = 1 180 / 154 / 182   R = 2 168-172 / 153-157 / 170-178
    // => a SYNCHRONOUS tick (chip_loop_tick: what the reference's 10 Hz thread issues) takes the rows form -- fused, ONE launch, no
    //    merge kernel behind the scan -- up to scan_sync_plain_bytes (4 GiB; beyond that the 25 us are < 1 % of the tick); rows of
    //    32 KiB and more (8192-D floats) with two rows per wave in flight there, pipelined ticks of such rows likewise up to 2 GiB.
    const double bytes = (double)n_rows * c->D * c->elem;
    if (bytes <= c->scan_plain_bytes) return 1;
    const bool wide_rows = (int64_t)c->D * c->elem >= 32768 && nq < 4;
    if (sync_tick && bytes <= c->scan_sync_plain_bytes) return wide_rows ? 2 : 1;
    if (wide_rows && bytes <= 2048.0 * 1024 * 1024) return 2;
    return 0;
}

int scan_grid_for(const Ctx *c, int64_t n_rows, int nq, bool q64)
{
    int block, bpc;
    scan_shape(c, nq, q64, &block, &bpc);
    const int wpb = block / 64;
    int64_t want = (n_rows + wpb - 1) / wpb;
    // scan_reserve leaves workgroup slots free for the small kernels.  With the round-1 load path a full grid held every
    // CU's registers (2 x 8 waves x 110 VGPRs), so a merge (or the RCCL all-gather kernel) launched underneath a running scan
    // waited for a scan workgroup to retire -- and grid-stride workgroups retire only at the END of the scan: reserving 4 slots
    // cut the merge's wait from ~1.7 ms to ~30 us at 1M rows.  At 66 VGPRs and 96 of 160 KiB LDS per CU our own merges fit
    // next to a full grid (reserve 0 and 4 measure the same at 125k rows); the reserve stays on where an exchange is attached
    // because the collective's kernel is not ours to size.  Default: 0 on a plain ctx, 4 on a sharded ctx / group
    // sub-context.  CHIP_SCAN_RESERVE overrides.
    int64_t cap = (int64_t)c->n_cus * bpc - c->scan_reserve;
    // Cache-sized prefixes (BASELINE config 2: 164 MB at 10k rows) are ticked over faster than a launch can ramp up and drain:
    // per-wave stamps of a 10k-row launch show ~20 us of streaming at the memory system's limit inside a 33-40 us launch (spread of
    // the wave starts, query staging, the workgroups' tails).  A launch that takes only HALF of every CU's workgroup slots lets the
    // launches of consecutive ticks (alternating scan streams) be resident TOGETHER, so one tick streams while its neighbours ramp
    // up / finish; 8 waves x 12 KiB in flight per CU already saturate the memory system.
    if (bpc > 1 && c->scan_short_bpc > 0 && c->scan_short_bpc < bpc && (double)n_rows * c->D * c->elem <= c->scan_half_bytes)
        cap = (int64_t)c->n_cus * c->scan_short_bpc - c->scan_reserve;
    if (cap > c->max_grid) cap = c->max_grid;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

template <typename T>
static int launch_scan_T(Ctx *c, hipStream_t s, const ScanArgs &a, int nq, int grid, size_t lds, int block)
{
    switch (nq) {
        case 1: return launch_scan_q<T, 1>(c, s, a, grid, lds, block);
        case 2: return launch_scan_q<T, 2>(c, s, a, grid, lds, block);
        case 3: return launch_scan_q<T, 3>(c, s, a, grid, lds, block);
        case 4: return launch_scan_q<T, 4>(c, s, a, grid, lds, block);
    }
    return CHIP_ERR_UNSUPPORTED;
}

template <int NQ, int NG>
static int launch_scan_wide(Ctx *c, hipStream_t s, const ScanArgs &a, int grid, size_t lds, int block)
{
    if (a.D % 512 == 0) {
        CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_topk_wide<NQ, NG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((db_scan_topk_wide<NQ, NG, true>), dim3(grid), dim3(block), lds, s, a);
    } else {
        CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_topk_wide<NQ, NG, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((db_scan_topk_wide<NQ, NG, false>), dim3(grid), dim3(block), lds, s, a);
    }
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

int launch_scan(Ctx *c, hipStream_t s, const ScanArgs &a, int nq, int grid)
{
    int block, bpc;
    scan_shape(c, nq, a.q64 != 0, &block, &bpc);
    const size_t lds = scan_lds_bytes(c, nq, a.K, block, a.q64 != 0, a.rows_form > 0);
    if (lds > 160 * 1024 || grid > 512) return CHIP_ERR_UNSUPPORTED;  // K2 holds one partial list per thread
    if (const int ng = scan_wide_ng(c, nq); ng > 0 && !a.q64 && a.rows_form == 0) {   // double rows wider than the LDS holds nq queries of
        if (nq == 3 && ng == 1) return launch_scan_wide<3, 1>(c, s, a, grid, lds, block);
        if (nq == 4 && ng == 1) return launch_scan_wide<4, 1>(c, s, a, grid, lds, block);
        if (nq == 4 && ng == 2) return launch_scan_wide<4, 2>(c, s, a, grid, lds, block);
        return CHIP_ERR_UNSUPPORTED;   // chip_create bounds D so that two double queries always fit
    }
    return c->elem == 8 ? launch_scan_T<double>(c, s, a, nq, grid, lds, block) : launch_scan_T<float>(c, s, a, nq, grid, lds, block);
}

int launch_resident(Ctx *c, hipStream_t s, const ResidentArgs &ra, int grid)
{
#ifdef CHIP_NO_ROWS_FORM
    (void)c; (void)s; (void)ra; (void)grid;
    return CHIP_ERR_UNSUPPORTED;
#else
    int block, bpc;
    scan_shape(c, 3, false, &block, &bpc);
    const size_t lds = scan_lds_bytes(c, 3, 1, block, false, true);
    if (lds + 64 > 160 * 1024 || grid > 512) return CHIP_ERR_UNSUPPORTED;
    return c->elem == 8 ? launch_resident_T<double>(c, s, ra, grid, lds, block) : launch_resident_T<float>(c, s, ra, grid, lds, block);
#endif
}

// ------------------------------------------------------------------------------------------------ K1s
// The whole score vector u = v^T * M.leftCols(k) of ONE query (src/Cerebro.cpp:1026; the reference's debug plot consumes all
// of u, :1047-1052): same per-row arithmetic as K1 (rows_dot + butterfly => the same bits), one wave per row, out[local row].
template <typename T, bool FULL>
__global__ __launch_bounds__(512) void db_scan_scores(ScanArgs a, double *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *qs = reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpb = blockDim.x >> 6;
    stage_queries<T, 1>(a, qs, tid, blockDim.x);
    __syncthreads();
    const int64_t tw = (int64_t)gridDim.x * wpb;
    for (int64_t r = (int64_t)blockIdx.x * wpb + wave; r < a.n_rows; r += tw) {
        const T *row[1] = {row_base<T>(a, r)};
        double acc[1][1];
        rows_dot<T, 1, 4, FULL, FULL ? 6 : 1, 1>(row, qs, a.D, lane, acc);   // whole 4 KiB batches: the asm load path of K1
        const double s = butterfly_sum(acc[0][0]);
        if (lane == 0) out[r] = s;
    }
}

int launch_scores(Ctx *c, hipStream_t s, const ScanArgs &a, double *out_dev)
{
    const size_t lds = (size_t)c->D * c->elem;
    int64_t grid = (a.n_rows + 7) / 8;
    if (grid > (int64_t)c->n_cus * 4) grid = (int64_t)c->n_cus * 4;
    if (grid < 1) grid = 1;
    const int chunk = 64 * (16 / c->elem) * 4;
    const bool full = a.D % chunk == 0;
    if (c->elem == 8) {
        if (lds > 65536) {
            CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_scores<double, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CHIP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(db_scan_scores<double, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (full) hipLaunchKernelGGL((db_scan_scores<double, true>), dim3((int)grid), dim3(512), lds, s, a, out_dev);
        else hipLaunchKernelGGL((db_scan_scores<double, false>), dim3((int)grid), dim3(512), lds, s, a, out_dev);
    } else {
        if (full) hipLaunchKernelGGL((db_scan_scores<float, true>), dim3((int)grid), dim3(512), lds, s, a, out_dev);
        else hipLaunchKernelGGL((db_scan_scores<float, false>), dim3((int)grid), dim3(512), lds, s, a, out_dev);
    }
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
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
    void *const *seg_table;
    int32_t seg_shift;
    int64_t seg_mask;
    void *ring;
    int32_t D;
    int32_t rank, nranks;
    int64_t first_global;
    int64_t row_stride;  // staged row r is global row first_global + r * row_stride (1: a contiguous batch; G: one shard's rows only)
    int64_t n;
    int64_t ring_from;   // only global rows >= ring_from are mirrored into the ring (the newest CHIP_RING_ROWS)
    uint32_t *flags;
};

// four consecutive elements of global row g, starting at element e, into the DB (if this rank owns the row) and the ring
template <typename T>
__device__ __forceinline__ void store_row4(const StoreArgs &a, int64_t g, int e, const T (&v)[4])
{
    typedef typename Vec16<T>::type V;
    constexpr int N = Vec16<T>::N;
    V pk[4 / N];
#pragma unroll
    for (int c = 0; c < 4; c++) pk[c / N][c % N] = v[c];
    if (a.nranks == 1 || (g % a.nranks) == a.rank) {
        const int64_t loc = a.nranks == 1 ? g : g / a.nranks;
        T *dst = static_cast<T *>(a.seg_table[loc >> a.seg_shift]) + (loc & a.seg_mask) * (int64_t)a.D + e;
#pragma unroll
        for (int h = 0; h < 4 / N; h++) reinterpret_cast<V *>(dst)[h] = pk[h];
    }
    if (a.ring && g >= a.ring_from) {
        T *dst = static_cast<T *>(a.ring) + (g % CHIP_RING_ROWS) * (int64_t)a.D + e;
#pragma unroll
        for (int h = 0; h < 4 / N; h++) reinterpret_cast<V *>(dst)[h] = pk[h];
    }
}

// M.col(_s) = desc (Cerebro.cpp:1005-1006): wire type S (float64[] of the .srv, or float) -> storage type T.
//   double -> float : round-to-nearest-even, flag bit0 unless (double)(float)x == x (lossless narrowing is the contract)
//   float  -> double, same -> same : exact.        NaN / Inf in any combination: flag bit1.
template <typename S, typename T>
__global__ __launch_bounds__(256) void convert_rows(StoreArgs a, const S *__restrict__ src)
{
    const int64_t per_row = a.D / 4;
    const int64_t total = a.n * per_row;
    uint32_t bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int e = (int)(i - r * per_row) * 4;
        S x[4];
        if constexpr (sizeof(S) == 8) {
            const f64x2 lo = *reinterpret_cast<const f64x2 *>(src + r * a.D + e);
            const f64x2 hi = *reinterpret_cast<const f64x2 *>(src + r * a.D + e + 2);
            x[0] = lo.x; x[1] = lo.y; x[2] = hi.x; x[3] = hi.y;
        } else {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(src + r * a.D + e);
            x[0] = w.x; x[1] = w.y; x[2] = w.z; x[3] = w.w;
        }
        T v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            v[c] = (T)x[c];
            if (!(fabs((double)x[c]) <= 1.79769313486231570e308)) bad |= 2u;   // NaN / Inf
            else if (sizeof(T) < sizeof(S) && (S)v[c] != x[c]) bad |= 1u;        // not fp32-representable (incl. overflow)
        }
        store_row4<T>(a, a.first_global + r * a.row_stride, e, v);
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
    a.row_stride = 1;
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

// src_elem: 8 = float64 rows (the wire type), 4 = float rows; the destination type is the ctx's storage type
int launch_store_rows(Ctx *c, hipStream_t s, const void *src, int src_elem, int64_t n, int64_t first_global, uint32_t *flags_dev, bool write_ring,
                      int64_t row_stride)
{
    StoreArgs a = make_store_args(c, first_global, n, flags_dev, write_ring);
    a.row_stride = row_stride;
    const dim3 grid(grid_for_elems(c, n * (c->D / 4))), block(256);
    if (src_elem == 8 && c->elem == 4) hipLaunchKernelGGL((convert_rows<double, float>), grid, block, 0, s, a, static_cast<const double *>(src));
    else if (src_elem == 8) hipLaunchKernelGGL((convert_rows<double, double>), grid, block, 0, s, a, static_cast<const double *>(src));
    else if (c->elem == 4) hipLaunchKernelGGL((convert_rows<float, float>), grid, block, 0, s, a, static_cast<const float *>(src));
    else hipLaunchKernelGGL((convert_rows<float, double>), grid, block, 0, s, a, static_cast<const float *>(src));
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

template <typename T>
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
        T v[4];   // the generator is defined in float; a double-storage DB holds the same values widened
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (kind == 0) v[c] = (T)((float)synth_from_key(key, e + c) * a.scale);
            else if (kind == 2) v[c] = (T)((float)synth_from_key(skey, e + c) * a.scale);
            else v[c] = (T)((float)(5 * synth_from_key(skey, e + c) + synth_from_key(key, e + c)) * a.scale_planted);
        }
        store_row4<T>(a.st, g, e, v);
    }
}

// Unit-L2 form of the generator (spec: oracle/dot_scan.c orc_synth_row_unit_f32; SURVEY.md 8d "rows = unit-L2-norm"): the row's integers
// v_e as above, S = sum v_e^2 EXACTLY (64-bit integers: any summation order gives the same S), inv = 1 / sqrt((double)S) (two correctly
// rounded fp64 operations), element = (float)((double)v_e * inv) -- one correctly rounded multiply, one conversion.  So the rows are
// unit vectors to fp32 rounding (| |row| - 1 | ~ 1e-8) AND bit-identical to the CPU generator.  One workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void synth_rows_unit(SynthArgs a)
{
    __shared__ unsigned long long part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_row = a.st.D / 4;
    for (int64_t r = blockIdx.x; r < a.st.n; r += gridDim.x) {
        const int64_t g = a.st.first_global + r;
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
        auto value = [&](int e) -> int64_t {
            if (kind == 0) return (int64_t)synth_from_key(key, e);
            if (kind == 2) return (int64_t)synth_from_key(skey, e);
            return 5 * (int64_t)synth_from_key(skey, e) + (int64_t)synth_from_key(key, e);
        };
        unsigned long long ss = 0;   // |v| <= 6 * 131070: v^2 < 2^40, a row of 8192 of them < 2^53
        for (int ch = tid; ch < per_row; ch += 256)
#pragma unroll
            for (int c = 0; c < 4; c++) { const int64_t v = value(4 * ch + c); ss += (unsigned long long)(v * v); }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
        __syncthreads();   // (part[] of the previous row has been read by everybody)
        if (lane == 0) part[wave] = ss;
        __syncthreads();
        const unsigned long long S = part[0] + part[1] + part[2] + part[3];
        const double inv = S ? 1.0 / sqrt((double)S) : 0.0;
        for (int ch = tid; ch < per_row; ch += 256) {
            T v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) v[c] = (T)(float)((double)value(4 * ch + c) * inv);
            store_row4<T>(a.st, g, 4 * ch, v);
        }
    }
}

int launch_synth(Ctx *c, hipStream_t s, int64_t first_global, int64_t n, uint64_t seed,
                 const int64_t *plant_dst_dev, const int64_t *plant_src_dev, const int32_t *plant_kind_dev, int64_t n_plant, int unit)
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
    if (unit) {
        const int grid = (int)(n < (int64_t)c->n_cus * 32 ? n : (int64_t)c->n_cus * 32);
        if (c->elem == 8) hipLaunchKernelGGL(synth_rows_unit<double>, dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(synth_rows_unit<float>, dim3(grid), dim3(256), 0, s, a);
    } else if (c->elem == 8) hipLaunchKernelGGL(synth_rows<double>, dim3(grid_for_elems(c, n * (c->D / 4))), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(synth_rows<float>, dim3(grid_for_elems(c, n * (c->D / 4))), dim3(256), 0, s, a);
    CHIP_HIP(c, hipGetLastError());
    return CHIP_OK;
}

}  // namespace chip
