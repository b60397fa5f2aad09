// ransac_common.h -- pieces shared by the batched RANSAC legs (pnp.hip, icp.hip): the counter-based RNG + sampler
// (device) and theia::Ransac's sequential selection rule replayed on the host (K7).
#pragma once
#include "chip_internal.h"
#include <cfloat>
#include <cmath>

namespace chip {

constexpr int kSampleMax = 16;   // DlsPnpWithRansac.h:45 uses 15, :118 uses 10

__device__ __forceinline__ uint64_t splitmix64_d(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t rng_draw(uint64_t seed, uint32_t hyp, uint32_t draw)
{
    return splitmix64_d(seed ^ ((uint64_t)hyp << 32) ^ (uint64_t)draw);
}

// theia::RandomSampler restated: partial Fisher-Yates over a VIRTUAL identity permutation of N: step i swaps positions i and
// j_i = i + draw_i % (N - i); only the <= 2S touched positions are materialised.  Executed by a WHOLE WAVE: the S draws and
// their 64-bit modulo are computed by lanes 0..S-1 at once, and the sparse permutation map lives in registers (lane e holds
// entry e) so that every lookup is a __ballot + v_readlane instead of a serial scan.  Returns sample i in lane i (< S).
__device__ __forceinline__ int ransac_sample_wave(uint64_t seed, int hyp, int N, int S, int lane)
{
    int jv = 0;
    if (lane < S) {
        const uint64_t x = rng_draw(seed, (uint32_t)hyp, (uint32_t)lane);
        jv = lane + (int)(x % (uint64_t)(N - lane));
    }
    int key = -1, val = 0, used = 0, mine = 0;
    for (int i = 0; i < S; i++) {
        const int j = __builtin_amdgcn_readlane(jv, i);
        const unsigned long long mi = __ballot(lane < used && key == i), mj = __ballot(lane < used && key == j);
        int vi = i, vj = j, pi = -1, pj = -1;
        if (mi) { pi = __builtin_ctzll(mi); vi = __builtin_amdgcn_readlane(val, pi); }
        if (mj) { pj = __builtin_ctzll(mj); vj = __builtin_amdgcn_readlane(val, pj); }
        if (pi < 0) { pi = used++; if (lane == pi) key = i; }   // idx[i] <- vj ; idx[j] <- vi
        if (lane == pi) val = vj;
        if (j != i) {
            if (pj < 0) { pj = used++; if (lane == pj) key = j; }
            if (lane == pj) val = vi;
        }
        if (lane == i) mine = vj;
    }
    return mine;
}

inline uint64_t splitmix64_h(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// CHIP_SAMPLER_THEIA_PERSISTENT: theia::RandomSampler as written -- Initialize() once (0..N-1), every Sample() continues on the
// permutation the previous one left (oracle/pnp_ransac.c orc_ransac_sample_persistent).  Inherently sequential over the
// hypotheses, S swaps each: done on the host, the kernels read the table (out[h * stride + i]).  `perm` is scratch (>= N).
inline void ransac_sample_table_persistent(uint64_t seed, int32_t H, int32_t N, int32_t S, int32_t stride, int32_t *perm, int32_t *out)
{
    for (int32_t i = 0; i < N; i++) perm[i] = i;
    for (int32_t h = 0; h < H; h++)
        for (int32_t i = 0; i < S; i++) {
            const uint64_t x = splitmix64_h(seed ^ ((uint64_t)(uint32_t)h << 32) ^ (uint64_t)(uint32_t)i);
            const int32_t j = i + (int32_t)(x % (uint64_t)(N - i));
            const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
            out[(size_t)h * stride + i] = perm[i];
        }
}

// theia::SampleConsensusEstimator::ComputeMaxIterations
inline int32_t ransac_max_iterations(int32_t S, double ratio, double log_fail, int32_t min_it, int32_t max_it)
{
    if (ratio == 1.0) return min_it;
    const double log_prob = std::log(1.0 - std::pow(ratio, (double)S)) - DBL_EPSILON;
    const double itf = std::floor(log_fail / log_prob) + 1.0;
    int32_t it = (itf > 2.0e9) ? 2000000000 : (int32_t)itf;
    if (it < min_it) it = min_it;
    if (it > max_it) it = max_it;
    return it;
}

inline int32_t ransac_initial_iterations(const chip_ransac_params *p)
{
    if (p->n_hypotheses > 0) return p->n_hypotheses;
    int32_t max_it = p->max_iterations;
    if (p->min_inlier_ratio > 0)
        max_it = ransac_max_iterations(p->sample_size, p->min_inlier_ratio, std::log(p->failure_probability), p->min_iterations, p->max_iterations);
    return max_it;
}

// K7: theia::Ransac::Estimate's sequential rule (strict '<': first best wins; early termination unless benchmark mode)
// replayed over per-hypothesis results.  Returns the winner (-1: none); *num_it = iterations the reference would run.
inline int32_t ransac_select(const chip_ransac_params *p, int32_t N, int32_t H, const int32_t *valid, const double *cost,
                             const int32_t *nin, int32_t *num_it_out, int32_t *n_models_out, double *best_cost_out)
{
    const bool bench = p->n_hypotheses > 0;
    const int32_t S = p->sample_size;
    const double log_fail = std::log(p->failure_probability);
    double best_cost = DBL_MAX;
    int32_t best_h = -1, n_models = 0, num_it = 0, max_it = H;
    for (num_it = 0; num_it < max_it; num_it++) {
        if (!valid[num_it]) continue;   // EstimateModel returned false
        n_models++;
        if (cost[num_it] < best_cost) {
            best_cost = cost[num_it];
            best_h = num_it;
            if (!bench) {
                const double ratio = (double)nin[num_it] / (double)N;
                if (ratio < (double)S / (double)N) continue;
                const int32_t mi = ransac_max_iterations(S, ratio, log_fail, p->min_iterations, p->max_iterations);
                if (mi < max_it) max_it = mi;
            }
        }
    }
    *num_it_out = num_it;
    *n_models_out = n_models;
    *best_cost_out = best_cost;
    return best_h;
}

}  // namespace chip
