// Dependent-chain latencies of the instruction kinds the PnP factor wave's critical path is made of, one wave on an otherwise idle
// CU (gfx950).  Prints cycles per dependent step (s_memtime around 64 unrolled steps, best of 20).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/chain_latency scripts/ubench/chain_latency.hip && scripts/ubench/chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND> __global__ void chain(unsigned long long *out, double *sink, double seed)
{
    const int lane = threadIdx.x;
    double d = seed + lane, e = seed * 0.5 + lane;
    unsigned u = (unsigned)lane * 2654435761u + (unsigned)seed;
    int lsel = (lane * 7 + 3) & 63;
    unsigned long long best = ~0ull;
    for (int it = 0; it < 20; it++) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (KIND == 0) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (KIND == 1) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(lsel));) }
        if (KIND == 2) { REP64(asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u));) }
        if (KIND == 3) { REP64(asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(u));) }
        if (KIND == 4) {   // 64-bit max step: two DPP moves + v_max_f64 (compiler-scheduled: v_mov_b32_dpp x2, v_max_f64)
            REP64({ const long long b = __double_as_longlong(d);
                    const int lo2 = __builtin_amdgcn_update_dpp(0, (int)b, 0x111, 0xf, 0xf, false);
                    const int hi2 = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x111, 0xf, 0xf, false);
                    d = __builtin_fmax(d, __longlong_as_double(((long long)hi2 << 32) | (unsigned)lo2)); asm volatile("" : "+v"(d)); })
        }
        if (KIND == 5) {   // readlane (SGPR lane select) -> VALU consumer -> next readlane
            REP64({ const unsigned s = __builtin_amdgcn_readlane(u, 5); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "s"(s)); })
        }
        if (KIND == 6) {   // v_cmp -> SGPR mask -> v_cndmask (VALU -> SGPR -> VALU)
            REP64(asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u) : "v"(lsel) : "vcc");)
        }
        if (KIND == 7) {   // v_cmp -> s_bcnt1 (VALU -> SALU) -> v_add (SALU -> VALU)
            REP64(asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\ts_bcnt1_i32_b64 s20, vcc\n\tv_add_u32 %0, %0, s20" : "+v"(u) : "v"(lsel) : "vcc", "s20");)
        }
        if (KIND == 8) { REP64(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(lsel));) }
        if (KIND == 9) { REP64(asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(u) : "v"(lsel << 2));) }
        if (KIND == 10) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(e));) }
        if (KIND == 11) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(d));) }
        if (KIND == 12) {  // ds_write + ds_read round trip (same address)
            __shared__ unsigned buf[64];
            REP64({ buf[lane] = u; __builtin_amdgcn_s_waitcnt(0xc07f); u = buf[lsel] + 1; __builtin_amdgcn_s_waitcnt(0xc07f); })
        }
        if (KIND == 13) { REP64(asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %1" : "+v"(d) : "v"(e));) }
        if (KIND == 14) { REP64(asm volatile("v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2" : "+v"(d), "+v"(e) : "v"(seed));) }   // two independent chains
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (t1 - t0 < best) best = t1 - t0;
    }
    if (lane == 0) out[KIND] = best;
    sink[lane] = d + e + u + lsel;
}

int main()
{
    unsigned long long *out, h[16] = {0};
    double *sink;
    (void)hipMalloc(&out, sizeof h); (void)hipMalloc(&sink, 64 * sizeof(double));
    (void)hipMemset(out, 0, sizeof h);
#define RUN(k) hipLaunchKernelGGL(chain<k>, dim3(1), dim3(64), 0, 0, out, sink, 1.25);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14)
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const char *name[] = {"v_add_f64 (dependent)", "v_add_u32 (dependent)", "s_nop 1 + v_max_u32_dpp row_shr", "s_nop 1 + v_max_u32_dpp row_bcast31",
                          "2 x v_mov_b32_dpp + v_max_f64", "v_readlane -> v_add (SGPR operand)", "v_cmp -> vcc -> v_cndmask", "v_cmp -> s_bcnt1 -> v_add",
                          "v_permlane32_swap", "ds_bpermute + wait", "v_fma_f64 (dependent)", "v_rcp_f64 (dependent)", "ds_write, wait, ds_read, wait",
                          "v_mul_f64 + v_add_f64 (dependent pair)", "2 independent v_add_f64 chains (per pair)"};
    for (int k = 0; k < 15; k++) printf("%-44s %7.1f cycles per step\n", name[k], (double)h[k] / 64.0);
    return 0;
}
