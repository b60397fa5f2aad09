// topk_merge.h -- exact top-K merge of sorted candidate lists by one workgroup (shared by kernels.hip and batch.hip).
#pragma once
#include "chip_internal.h"
#include <cmath>

namespace chip {

__device__ __forceinline__ bool key_gt(double s, int64_t i, double s2, int64_t i2)
{
    return s > s2 || (s == s2 && i > i2);
}

// ------------------------------------------------------------------------------------------------ list merge
// Exact top-K of n_lists (<= blockDim.x = 512) SORTED lists of K candidates per query, by ONE workgroup, plus the
// accept rule of Cerebro.cpp:1056.  Selection instead of K serial block-wide argmax rounds:
//   1. thread t loads the HEAD of list t;                       2. every wave ranks its 64 heads (v_readlane loops)
//   and posts its K best to LDS;  3. one wave per query ranks those <= 8K heads: T1 = K-th best head overall.
//   Every global top-K entry is >= T1 (K heads already are), and only the K lists whose head >= T1 can contribute,
//   each at most K entries => at most K*K <= 256 survivors;     4. survivors are appended to an LDS list (LDS atomic
//   cursor) and ranked by one wave per query; rank r < K is written to out[r].  Keys (score desc, idx desc) are
//   unique for valid entries (indices are), so ranks are a permutation and the result is partition independent.
// `in` is [n_lists][list_stride queries][K]; this workgroup merges queries q_off .. q_off+NQ-1 into out[0..NQ)[K].
// smem: kMergeSmem bytes.
constexpr int kMergeSmem = 28 * 1024;
constexpr int64_t kFailedShardIdx = -2;   // index of every entry of the list a failed shard contributes (unused slots carry -1)
constexpr int kSurvCap = CHIP_MAX_TOPK * CHIP_MAX_TOPK;  // 256

__device__ __forceinline__ double readlane_f64(double v, int j)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), j);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), j);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int j)
{
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), j);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), j);
    return ((int64_t)hi << 32) | (unsigned int)lo;
}

// Order-preserving integer image of a finite (or -inf) double: a > b  <=>  okey(a) > okey(b).
__device__ __forceinline__ unsigned long long okey(double s)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Number of lanes of the wave whose key (score desc, idx desc) is strictly greater than this lane's.  Fast path: the
// upper 32 bits of okey (sign, exponent, 20 mantissa bits) are compared with full-rate 32-bit ops; only when two VALID
// lanes agree in those bits (scores within ~1e-6 relative, e.g. exact duplicates) does the wave redo the loop with the
// exact (f64, i64) comparison.  Both paths give the same ranks; the branch is wave-uniform.
__device__ __forceinline__ int wave_rank(double s, int64_t i, bool valid)
{
    const unsigned kh = (unsigned)(okey(s) >> 32);
    int rank = 0, eq = 0;
#pragma unroll 16
    for (int j = 0; j < 64; j++) {
        const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)kh, j);
        rank += o > kh ? 1 : 0;
        eq += o == kh ? 1 : 0;
    }
    if (__ballot(valid && eq > 1) == 0ull) return rank;   // eq counts the lane itself
    rank = 0;
#pragma unroll 8
    for (int j = 0; j < 64; j++) rank += key_gt(readlane_f64(s, j), readlane_i64(i, j), s, i) ? 1 : 0;
    return rank;
}

// Same for two keys per lane (128 candidates): ranks of (s0,i0) and (s1,i1) among all 128.
__device__ __forceinline__ void wave_rank2(const double s[2], const int64_t i[2], int &r0, int &r1)
{
    const unsigned k0 = (unsigned)(okey(s[0]) >> 32), k1 = (unsigned)(okey(s[1]) >> 32);
    int e0 = 0, e1 = 0;
    r0 = r1 = 0;
#pragma unroll 16
    for (int j = 0; j < 64; j++) {
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)k0, j), b = (unsigned)__builtin_amdgcn_readlane((int)k1, j);
        r0 += (a > k0 ? 1 : 0) + (b > k0 ? 1 : 0);
        e0 += (a == k0 ? 1 : 0) + (b == k0 ? 1 : 0);
        r1 += (a > k1 ? 1 : 0) + (b > k1 ? 1 : 0);
        e1 += (a == k1 ? 1 : 0) + (b == k1 ? 1 : 0);
    }
    if (__ballot((i[0] >= 0 && e0 > 1) || (i[1] >= 0 && e1 > 1)) == 0ull) return;
    r0 = r1 = 0;
#pragma unroll 8
    for (int j = 0; j < 64; j++) {
        const double a0 = readlane_f64(s[0], j), a1 = readlane_f64(s[1], j);
        const int64_t b0 = readlane_i64(i[0], j), b1 = readlane_i64(i[1], j);
        r0 += (key_gt(a0, b0, s[0], i[0]) ? 1 : 0) + (key_gt(a1, b1, s[0], i[0]) ? 1 : 0);
        r1 += (key_gt(a0, b0, s[1], i[1]) ? 1 : 0) + (key_gt(a1, b1, s[1], i[1]) ? 1 : 0);
    }
}

template <int NQ>
__device__ __forceinline__ void merge_sorted_lists(const chip_topk_entry *in, int n_lists, int list_stride, int q_off, int K, chip_topk_entry *out,
                                                   chip_tick_result *result, int64_t l, int locality, double thresh, char *smem)
{
    chip_topk_entry *candA = reinterpret_cast<chip_topk_entry *>(smem);                    // [NQ][8 waves * 16]
    chip_topk_entry *surv = candA + CHIP_MAX_NQ * 8 * CHIP_MAX_TOPK;                        // [NQ][kSurvCap]
    chip_topk_entry *T1s = surv + CHIP_MAX_NQ * kSurvCap;                                   // [NQ]
    chip_topk_entry *tops = T1s + CHIP_MAX_NQ;                                              // [NQ]
    int *cnt = reinterpret_cast<int *>(tops + CHIP_MAX_NQ);                                 // [NQ]
    int *failed = cnt + CHIP_MAX_NQ;                                                        // a list carries the failure mark
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6, nw = blockDim.x >> 6;

    // ---- 1. heads ----
    double hs[NQ];
    int64_t hi[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        if (t < n_lists) { const chip_topk_entry x = in[((int64_t)t * list_stride + q_off + q) * K]; hs[q] = x.score; hi[q] = x.idx; }
        else { hs[q] = -INFINITY; hi[q] = -1; }
    }
    if (t < NQ) cnt[t] = 0;
    if (t == NQ) *failed = 0;
    // ---- 2. per-wave rank of the heads; the K best of each wave go to candA[q][w*K + rank] ----
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const bool valid = hi[q] >= 0;
        const int rank = wave_rank(hs[q], hi[q], valid);
        const int nvalid = __popcll(__ballot(valid));
        chip_topk_entry *dst = candA + (q * 8 + w) * CHIP_MAX_TOPK;
        if (valid && rank < K) { dst[rank].score = hs[q]; dst[rank].idx = hi[q]; }
        if (lane < K && lane >= nvalid) { dst[lane].score = -INFINITY; dst[lane].idx = -1; }
    }
    __syncthreads();
    // a shard that could not take part in this exchange sent the marked neutral list (kFailedShardIdx in every entry): the merge
    // ignores its entries like any unused slot and reports the mark, so that EVERY rank fails the call alike (chip_multi.hip)
    if (t < n_lists && hi[0] == kFailedShardIdx) *failed = 1;
    // ---- 3. T1[q] = K-th best head overall (or "everything survives" when fewer than K lists are non-empty) ----
    if (w < NQ) {
        const int q = w;
        double cs[2];
        int64_t ci[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c = lane + 64 * h;  // candidate (wave c / K, slot c % K)
            if (c < nw * K) { const chip_topk_entry x = candA[(q * 8 + c / K) * CHIP_MAX_TOPK + (c % K)]; cs[h] = x.score; ci[h] = x.idx; }
            else { cs[h] = -INFINITY; ci[h] = -1; }
        }
        int r0, r1;
        wave_rank2(cs, ci, r0, r1);
        const int nv = __popcll(__ballot(ci[0] >= 0)) + __popcll(__ballot(ci[1] >= 0));
        if (nv >= K) {
            if (ci[0] >= 0 && r0 == K - 1) { T1s[q].score = cs[0]; T1s[q].idx = ci[0]; }
            if (ci[1] >= 0 && r1 == K - 1) { T1s[q].score = cs[1]; T1s[q].idx = ci[1]; }
        } else if (lane == 0) { T1s[q].score = -INFINITY; T1s[q].idx = -1; }
    }
    __syncthreads();
    // ---- 4. survivors: the prefix of every list whose head >= T1 ----
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const double ts = T1s[q].score;
        const int64_t ti = T1s[q].idx;
        if (t < n_lists && hi[q] >= 0 && !key_gt(ts, ti, hs[q], hi[q])) {
            double es[CHIP_MAX_TOPK];
            int64_t ei[CHIP_MAX_TOPK];
            es[0] = hs[q]; ei[0] = hi[q];
#pragma unroll
            for (int j = 1; j < CHIP_MAX_TOPK; j++) {
                if (j < K) { const chip_topk_entry x = in[((int64_t)t * list_stride + q_off + q) * K + j]; es[j] = x.score; ei[j] = x.idx; }
                else { es[j] = -INFINITY; ei[j] = -1; }
            }
            int c = 1;
#pragma unroll
            for (int j = 1; j < CHIP_MAX_TOPK; j++)
                if (c == j && ei[j] >= 0 && !key_gt(ts, ti, es[j], ei[j])) c = j + 1;   // sorted: survivors are a prefix
            const int base = atomicAdd(&cnt[q], c);
#pragma unroll
            for (int j = 0; j < CHIP_MAX_TOPK; j++)
                if (j < c && base + j < kSurvCap) { surv[q * kSurvCap + base + j].score = es[j]; surv[q * kSurvCap + base + j].idx = ei[j]; }
        }
    }
    __syncthreads();
    // ---- 5. rank the survivors; rank r < K -> out[r] ----
    if (w < NQ) {
        const int q = w;
        int n = cnt[q];
        if (n > kSurvCap) n = kSurvCap;  // cannot happen (<= K*K), keeps the loops bounded
        double cs[4];
        int64_t ci[4];
        int rk[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int c = lane + 64 * h;
            if (c < n) { cs[h] = surv[q * kSurvCap + c].score; ci[h] = surv[q * kSurvCap + c].idx; }
            else { cs[h] = -INFINITY; ci[h] = -1; }
            rk[h] = 0;
        }
        for (int j = 0; j < n; j++) {
            const chip_topk_entry x = surv[q * kSurvCap + j];  // LDS broadcast
#pragma unroll
            for (int h = 0; h < 4; h++) rk[h] += key_gt(x.score, x.idx, cs[h], ci[h]) ? 1 : 0;
        }
#pragma unroll
        for (int h = 0; h < 4; h++) {
            if (ci[h] >= 0 && rk[h] < K) {
                if (out) { out[q * K + rk[h]].score = cs[h]; out[q * K + rk[h]].idx = ci[h]; }
                if (rk[h] == 0) { tops[q].score = cs[h]; tops[q].idx = ci[h]; }
            }
        }
        if (lane < K && lane >= n) {  // fewer than K candidates in total: pad
            if (out) { out[q * K + lane].score = -INFINITY; out[q * K + lane].idx = -1; }
            if (lane == 0) { tops[q].score = -INFINITY; tops[q].idx = -1; }
        }
    }
    __syncthreads();
    if (out != nullptr && t == 0 && *failed) out[0].idx = kFailedShardIdx;
    if (result != nullptr && t == 0) {
        chip_tick_result res;
        res.status = *failed ? CHIP_TICK_FAILED : CHIP_TICK_SCANNED;
        res.found = 0;
        res.idx_curr = -1;
        res.idx_prev = -1;
        res.score = 0.0;
        for (int q = 0; q < 3; q++) {
            res.argmax[q] = q < NQ ? tops[q].idx : -1;
            res.maxv[q] = q < NQ ? tops[q].score : -INFINITY;
        }
        if (NQ >= 3 && res.argmax[0] >= 0 && res.argmax[1] >= 0 && res.argmax[2] >= 0) {
            // Cerebro.cpp:1056  abs(u_argmax-um_argmax) < LOCALITY && abs(u_argmax-umm_argmax) < LOCALITY && u_max > THRESH
            int64_t d1 = res.argmax[0] - res.argmax[1];
            int64_t d2 = res.argmax[0] - res.argmax[2];
            if (d1 < 0) d1 = -d1;
            if (d2 < 0) d2 = -d2;
            if (d1 < locality && d2 < locality && res.maxv[0] > thresh) {
                res.found = 1;
                res.idx_curr = l - 1;  // Cerebro.cpp:1080
                res.idx_prev = res.argmax[0];
                res.score = res.maxv[0];
            }
        }
        *result = res;
    }
}


}  // namespace chip
