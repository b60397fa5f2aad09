// Microbenchmark: what limits v_mfma_f32_32x32x2_f32 streams shaped like db_gemm_topk's inner loop?
//   mode 0: pure MFMA, 4 accumulators, operands in registers
//   mode 1: + fragments read from LDS (ds_read2_b32, register double buffer) as in the kernel
//   mode 2: mode 1 + 32 ds_write_b32 per 64 MFMAs + one barrier per 64 MFMAs (4-wave workgroup)
//   mode 3: fragments as ds_read_b128 + one v_cndmask per operand (the LDS-DMA layout of db_gemm_topk), no loads, no barrier
//   mode 4: mode 3 + one barrier per 64 MFMAs
//   mode 5: mode 4 + 8 global_load_lds_dwordx4 per wave per 64 MFMAs from a small (cache-resident) buffer, counted vmcnt
// build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void k3(float *out, int iters, const float *src)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S = reinterpret_cast<float *>(smem);   // 4 stages x (A 128x32 + B 128x32) floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 2 * 128 * 32; i += 256) S[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fk = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const unsigned fmask = fk ? 0xffffffffu : 0u;
    int f_off[4], f_sw[4];
    const int rows[4] = {wm * 64 + fr, wm * 64 + 32 + fr, wn * 64 + fr, wn * 64 + 32 + fr};
    for (int o = 0; o < 4; o++) { f_off[o] = (o >= 2 ? 4096 : 0) + rows[o] * 32; f_sw[o] = (rows[o] >> 1) & 7; }
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void *)S);
    const float *g = src + (size_t)(blockIdx.x % 64) * 8192 + tid * 4;
    for (int it = 0; it < iters; it++) {
        const float *St = S + (it & 3) * 8192;
        if (MODE == 5) {
            const unsigned st = lds_base + (unsigned)(((it + 3) & 3) * 8192 + wave * 256) * 4u;
#pragma unroll
            for (int u = 0; u < 4; u++) { glds16(g + u * 1024, st + u * 4096u); glds16(g + 4096 + u * 1024, st + 16384u + u * 4096u); }
        }
        u32x4 f[2][4];
#pragma unroll
        for (int o = 0; o < 4; o++) f[0][o] = *reinterpret_cast<const u32x4 *>(St + f_off[o] + ((0 ^ f_sw[o]) << 2));
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            if (k4 + 1 < 8) {
#pragma unroll
                for (int o = 0; o < 4; o++) f[(k4 + 1) & 1][o] = *reinterpret_cast<const u32x4 *>(St + f_off[o] + (((k4 + 1) ^ f_sw[o]) << 2));
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                float v[4];
#pragma unroll
                for (int o = 0; o < 4; o++) { const u32x4 &w = f[k4 & 1][o]; v[o] = __uint_as_float((w[2 * t + 1] & fmask) | (w[2 * t] & ~fmask)); }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[2], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[3], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], v[2], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[1], v[3], acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == 5) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

// mode 6: A from a [k][row] image (conflict-free b32 reads, no VALU), B row-major + swizzle read as two b32 at +fk*4 (4-way
//         bank conflicts, no VALU), no barrier;  mode 7: + barrier + LDS-DMA as mode 5
template <int MODE>
__global__ __launch_bounds__(256) void k6(float *out, int iters, const float *src)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 2 * 128 * 32; i += 256) S[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fk = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int a_off = fk * 128 + wm * 64 + fr;                       // + kk * 256 (+32 for the second row block)
    const int rb0 = wn * 64 + fr, rb1 = rb0 + 32;
    const int b_off0 = 4096 + rb0 * 32 + fk, b_off1 = 4096 + rb1 * 32 + fk, sw0 = (rb0 >> 1) & 7, sw1 = (rb1 >> 1) & 7;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void *)S);
    const float *g = src + (size_t)(blockIdx.x % 64) * 8192 + tid * 4;
    for (int it = 0; it < iters; it++) {
        const float *St = S + (it & 3) * 8192;
        if (MODE == 7) {
            const unsigned st = lds_base + (unsigned)(((it + 3) & 3) * 8192 + wave * 256) * 4u;
#pragma unroll
            for (int u = 0; u < 4; u++) { glds16(g + u * 1024, st + u * 4096u); glds16(g + 4096 + u * 1024, st + 16384u + u * 4096u); }
        }
        float f[2][8];   // [buf][a0(2) a1(2) b0(2) b1(2)] for k-steps (2 k4, 2 k4 + 1)
        auto rd = [&](int k4, float *d) {
            d[0] = St[a_off + (2 * k4) * 256];      d[1] = St[a_off + (2 * k4 + 1) * 256];
            d[2] = St[a_off + 32 + (2 * k4) * 256]; d[3] = St[a_off + 32 + (2 * k4 + 1) * 256];
            const int p0 = (k4 ^ sw0) << 2, p1 = (k4 ^ sw1) << 2;
            d[4] = St[b_off0 + p0]; d[5] = St[b_off0 + p0 + 2];
            d[6] = St[b_off1 + p1]; d[7] = St[b_off1 + p1 + 2];
        };
        rd(0, f[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            if (k4 + 1 < 8) rd(k4 + 1, f[(k4 + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float a0 = f[k4 & 1][t], a1 = f[k4 & 1][2 + t], b0 = f[k4 & 1][4 + t], b1 = f[k4 & 1][6 + t];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        if (MODE == 7) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run6(int blocks, int iters, const char *name)
{
    float *out, *src;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipMalloc(&src, sizeof(float) * 64 * 8192 + 65536);
    hipMemset(src, 0, sizeof(float) * 64 * 8192 + 65536);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k6<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k6<MODE>, dim3(blocks), dim3(256), 131072, 0, out, 16, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k6<MODE>, dim3(blocks), dim3(256), 131072, 0, out, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64.0 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks=%4d: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    hipFree(out); hipFree(src);
}

// mode 8/9: per-wave tile 128 x 64 (4 x 2 MFMA tiles, 128 accumulator registers), workgroup tile 256 x 128, one workgroup per CU:
//           per k-pair 4 conflict-free A reads (ds_read2st64) + 2 conflicted B reads feed 16 MFMAs; mode 9 adds 12 LDS-DMA per wave
//           and one counted barrier per 128 MFMAs (two stages of 48 KiB)
template <int MODE>
__global__ __launch_bounds__(256) void k8(float *out, int iters, const float *src)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S = reinterpret_cast<float *>(smem);   // 2 stages x (A [32 k][256 rows] + B [128 rows][32]) = 2 x 12288 floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 12288; i += 256) S[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4][2];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fk = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int a_off = fk * 256 + wm * 128 + fr;                      // + kk * 512, + 32 i for row block i
    const int rb0 = wn * 64 + fr, rb1 = rb0 + 32;
    const int b_off0 = 8192 + rb0 * 32 + fk, b_off1 = 8192 + rb1 * 32 + fk, sw0 = (rb0 >> 1) & 7, sw1 = (rb1 >> 1) & 7;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void *)S);
    const float *g = src + (size_t)(blockIdx.x % 32) * 12288 + tid * 4;
    for (int it = 0; it < iters; it++) {
        const float *St = S + (it & 1) * 12288;
        if (MODE == 9) {
            const unsigned st = lds_base + (unsigned)(((it + 1) & 1) * 12288 + wave * 256) * 4u;
#pragma unroll
            for (int u = 0; u < 12; u++) glds16(g + u * 1024, st + u * 4096u);
        }
        float f[2][12];   // [buf][a0..a3 (2 each), b0, b1 (2 each)]
        auto rd = [&](int k4, float *d) {
#pragma unroll
            for (int i = 0; i < 4; i++) { d[2 * i] = St[a_off + 32 * i + (2 * k4) * 512]; d[2 * i + 1] = St[a_off + 32 * i + (2 * k4 + 1) * 512]; }
            const int p0 = (k4 ^ sw0) << 2, p1 = (k4 ^ sw1) << 2;
            d[8] = St[b_off0 + p0]; d[9] = St[b_off0 + p0 + 2];
            d[10] = St[b_off1 + p1]; d[11] = St[b_off1 + p1 + 2];
        };
        rd(0, f[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            if (k4 + 1 < 8) rd(k4 + 1, f[(k4 + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float b0 = f[k4 & 1][8 + t], b1 = f[k4 & 1][10 + t];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float a = f[k4 & 1][2 * i + t];
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[i][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        if (MODE == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run8(int blocks, int iters, const char *name)
{
    float *out, *src;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipMalloc(&src, sizeof(float) * 32 * 12288 + 65536);
    hipMemset(src, 0, sizeof(float) * 32 * 12288 + 65536);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k8<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k8<MODE>, dim3(blocks), dim3(256), 98304, 0, out, 16, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k8<MODE>, dim3(blocks), dim3(256), 98304, 0, out, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 128.0 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks=%4d: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)  %s\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(src);
}


// mode 10/11: the same 128 x 64 per-wave tile with EIGHT waves per workgroup (workgroup tile 256 x 256, two waves per SIMD that
//             meet at the same barrier), two stages of 64 KiB, 8 LDS-DMA per wave and one counted barrier per 128 MFMAs
template <int MODE>
__global__ __launch_bounds__(512) void k10(float *out, int iters, const float *src)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *S = reinterpret_cast<float *>(smem);   // 2 stages x (A [32 k][256 rows] + B [256 rows][32]) = 2 x 16384 floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 16384; i += 512) S[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4][2];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fk = lane >> 5, wm = wave >> 2, wn = wave & 3;
    const int a_off = fk * 256 + wm * 128 + fr;
    const int rb0 = wn * 64 + fr, rb1 = rb0 + 32;
    const int b_off0 = 8192 + rb0 * 32 + fk, b_off1 = 8192 + rb1 * 32 + fk, sw0 = (rb0 >> 1) & 7, sw1 = (rb1 >> 1) & 7;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void *)S);
    const float *g = src + (size_t)(blockIdx.x % 16) * 16384 + (tid & 255) * 4;
    for (int it = 0; it < iters; it++) {
        const float *St = S + (it & 1) * 16384;
        if (MODE == 11) {
            const unsigned st = lds_base + (unsigned)(((it + 1) & 1) * 16384 + wave * 256) * 4u;
#pragma unroll
            for (int u = 0; u < 8; u++) glds16(g + u * 2048, st + u * 8192u);
        }
        float f[2][12];
        auto rd = [&](int k4, float *d) {
#pragma unroll
            for (int i = 0; i < 4; i++) { d[2 * i] = St[a_off + 32 * i + (2 * k4) * 512]; d[2 * i + 1] = St[a_off + 32 * i + (2 * k4 + 1) * 512]; }
            const int p0 = (k4 ^ sw0) << 2, p1 = (k4 ^ sw1) << 2;
            d[8] = St[b_off0 + p0]; d[9] = St[b_off0 + p0 + 2];
            d[10] = St[b_off1 + p1]; d[11] = St[b_off1 + p1 + 2];
        };
        rd(0, f[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int k4 = 0; k4 < 8; k4++) {
            if (k4 + 1 < 8) rd(k4 + 1, f[(k4 + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const float b0 = f[k4 & 1][8 + t], b1 = f[k4 & 1][10 + t];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float a = f[k4 & 1][2 * i + t];
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[i][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        if (MODE == 11) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static void run10(int blocks, int iters, const char *name)
{
    float *out, *src;
    hipMalloc(&out, sizeof(float) * 512 * blocks);
    hipMalloc(&src, sizeof(float) * 16 * 16384 + 131072);
    hipMemset(src, 0, sizeof(float) * 16 * 16384 + 131072);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k10<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k10<MODE>, dim3(blocks), dim3(512), 131072, 0, out, 16, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k10<MODE>, dim3(blocks), dim3(512), 131072, 0, out, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * iters * 128.0 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks=%4d: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)  %s\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(src);
}

template <int MODE>
static void run3(int blocks, int iters, const char *name)
{
    float *out, *src;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipMalloc(&src, sizeof(float) * 64 * 8192 + 65536);
    hipMemset(src, 0, sizeof(float) * 64 * 8192 + 65536);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k3<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3<MODE>, dim3(blocks), dim3(256), 131072, 0, out, 16, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k3<MODE>, dim3(blocks), dim3(256), 131072, 0, out, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64.0 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks=%4d: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)  err=%s\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(src);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float S[2 * 2 * 128 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 2 * 128 * 33; i += 256) S[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    const int fr = lane & 31, fk = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int fa = (wm * 64 + fr) * 33 + fk, fb = 128 * 33 + (wn * 64 + fr) * 33 + fk;
    float x = (float)lane * 1e-3f, y = 1.0f + x;
    for (int it = 0; it < iters; it++) {
        const float *St = S + (it & 1) * (2 * 128 * 33);
        if (MODE == 2) {
            float *Sw = S + ((it + 1) & 1) * (2 * 128 * 33);
            const int ld_c = (tid % 8) * 4, ld_r = tid / 8;
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    Sw[(ld_r + 32 * u) * 33 + ld_c + c] = x + c;
                    Sw[128 * 33 + (ld_r + 32 * u) * 33 + ld_c + c] = y + u;
                }
        }
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 16; g++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, acc[1][1], 0, 0, 0);
            }
        } else {
            float f[2][8];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                f[0][4 * t + 0] = St[fa + 2 * t]; f[0][4 * t + 1] = St[fa + 32 * 33 + 2 * t];
                f[0][4 * t + 2] = St[fb + 2 * t]; f[0][4 * t + 3] = St[fb + 32 * 33 + 2 * t];
            }
#pragma unroll
            for (int g = 0; g < 8; g++) {
                if (g + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int kk = 2 * (g + 1) + t;
                        f[(g + 1) & 1][4 * t + 0] = St[fa + 2 * kk]; f[(g + 1) & 1][4 * t + 1] = St[fa + 32 * 33 + 2 * kk];
                        f[(g + 1) & 1][4 * t + 2] = St[fb + 2 * kk]; f[(g + 1) & 1][4 * t + 3] = St[fb + 32 * 33 + 2 * kk];
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const float a0 = f[g & 1][4 * t + 0], a1 = f[g & 1][4 * t + 1], b0 = f[g & 1][4 * t + 2], b1 = f[g & 1][4 * t + 3];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int e = 0; e < 16; e++) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(int blocks, int iters, const char *name)
{
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 64.0 /*mfma*/ * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks=%4d: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    hipFree(out);
}

int main()
{
    const int iters = 20000;
    for (int blocks : {256, 512}) {
        run<0>(blocks, iters, "pure MFMA, 4 accumulators");
        run<1>(blocks, iters, "+ LDS fragment reads (double-buffered)");
        run<2>(blocks, iters, "+ 32 ds_write_b32 + barrier per 64 MFMA");
    }
    run3<3>(256, iters, "b128 fragments + cndmask (1 WG/CU)");
    run3<4>(256, iters, "  + barrier per 64 MFMA");
    run3<5>(256, iters, "  + 8 LDS-DMA per wave per 64 MFMA, vmcnt(16)");
    run6<6>(256, iters, "A [k][row] b32 + B swizzled 2 x b32, no VALU");
    run6<7>(256, iters, "  + LDS-DMA + counted barrier per 64 MFMA");
    run8<8>(256, iters / 2, "128x64 per wave (16 MFMA per 6 reads), no DMA");
    run8<9>(256, iters / 2, "  + 12 LDS-DMA + barrier per 128 MFMA");
    run10<10>(256, iters / 4, "8 waves x 128x64 (two waves per SIMD), no DMA");
    run10<11>(256, iters / 4, "  + 8 LDS-DMA + barrier per 128 MFMA");
    return 0;
}
