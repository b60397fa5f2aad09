// Does global_load_lds_dwordx4 honour an LDS destination base (M0) that is only 4-byte aligned?
// If it does, the DB tile of db_gemm_topk could be staged with a per-8-row rotation of 1..3 floats, which would make its
// ds_read_b32 fragment reads bank-conflict free (today: 4-way).  Prints where the 64 lanes' 16-byte payloads landed.
// build: hipcc -O3 --offload-arch=gfx950 glds_align_probe.hip -o glds_align_probe.bin
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

__global__ void probe(const float *src, float *out, int misalign_bytes)
{
    __shared__ __attribute__((aligned(16))) float S[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) S[i] = -1.f;
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void *)S);
    glds16(src + threadIdx.x * 4, base + 64 + misalign_bytes);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = S[i];
}

int main()
{
    float h[256], *src, *out, r[1024];
    for (int i = 0; i < 256; i++) h[i] = (float)i;
    hipMalloc(&src, sizeof h); hipMalloc(&out, sizeof r);
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    for (int mis : {0, 4, 8, 12}) {
        probe<<<1, 64>>>(src, out, mis);
        if (hipDeviceSynchronize() != hipSuccess) { printf("misalign %d: device error\n", mis); return 1; }
        hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost);
        int first = -1, n = 0, ordered = 1;
        for (int i = 0; i < 1024; i++) if (r[i] >= 0.f) { if (first < 0) first = i; n++; }
        for (int i = 0; i < 256 && first >= 0 && first + i < 1024; i++) if (r[first + i] != (float)i) ordered = 0;
        printf("misalign %2d B: first written float index %d (expected %d), %d floats written, contiguous 0..255 in order: %s\n",
               mis, first, 16 + mis / 4, n, ordered ? "yes" : "no");
    }
    return 0;
}
