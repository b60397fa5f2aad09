// Dense fp64 VECTOR peak of the chip (gfx950), measured: every lane runs NCHAIN independent v_fma_f64 chains, enough waves per SIMD
// that the pipe never waits (the PnP roofline's denominator: MI355X_MICROARCH.md quotes 78.6 TFLOP/s = 256 CUs x 64 lanes x 2 flop
// x 2.4 GHz; this probe says what THIS box sustains).  Also prints the same with v_mul_f64 + v_add_f64 pairs (the PnP kernels are
// built with -ffp-contract=off: a mul + add pair is two issue slots for two flops, i.e. HALF the fma peak).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/fp64_peak scripts/ubench/fp64_peak.hip && scripts/ubench/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int KIND> __global__ void __launch_bounds__(256) dense(double *sink, double x, double y, int iters)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                               "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        } else {
            REP16(asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                               "v_add_f64 %0, %0, %9\n\tv_add_f64 %1, %1, %9\n\tv_add_f64 %2, %2, %9\n\tv_add_f64 %3, %3, %9\n\t"
                               "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8\n\t"
                               "v_add_f64 %4, %4, %9\n\tv_add_f64 %5, %5, %9\n\tv_add_f64 %6, %6, %9\n\tv_add_f64 %7, %7, %9"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));)
        }
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

template <int KIND> static double run(int blocks, int iters, double *sink)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(dense<KIND>, dim3(blocks), dim3(256), 0, 0, sink, 0.999999, 1e-9, 16);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(dense<KIND>, dim3(blocks), dim3(256), 0, 0, sink, 0.999999, 1e-9, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 2.0 * 8 * 16 * (double)iters * 256.0 * blocks;   // 8 chains x 16 reps x (1 fma | mul + add) x 2 flop
    return flops / (best * 1e-3) / 1e12;
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int wg_per_cu = 8;                     // 8 x 256 threads = 32 waves per CU = 8 per SIMD
    const int blocks = cus * wg_per_cu, iters = argc > 1 ? atoi(argv[1]) : 4000;
    double *sink; (void)hipMalloc(&sink, sizeof(double) * 256 * blocks);
    const double fma = run<0>(blocks, iters, sink), ma = run<1>(blocks, iters, sink);
    const double clk_ghz = p.clockRate / 1e6, spec = cus * 64 * 2.0 * clk_ghz / 1e3;
    printf("device %s  CUs %d  clockRate %.3f GHz  -> fp64 vector spec at that clock %.1f TFLOP/s (MI355X_MICROARCH.md: 78.6)\n", p.gcnArchName, cus, clk_ghz, spec);
    printf("v_fma_f64 dense (8 chains/lane, 8 waves/SIMD)        %7.2f TFLOP/s  (%.3f of 78.6)\n", fma, fma / 78.6);
    printf("v_mul_f64 + v_add_f64 dense (-ffp-contract=off form) %7.2f TFLOP/s  (%.3f of 78.6)\n", ma, ma / 78.6);
    printf("{\"fp64_fma_tflops\": %.3f, \"fp64_mul_add_tflops\": %.3f, \"spec_tflops\": 78.6, \"cus\": %d, \"clock_ghz\": %.3f}\n", fma, ma, cus, clk_ghz);
    return 0;
}
