// bar_pingpong.hip -- round 5 probe: how a resident kernel should be handed a command.  Ping-pong between a host thread and ONE polling
// wave, two ways for the "ping":
//   A  the host writes a word in PINNED HOST memory, the wave polls it with system-scope loads over PCIe (what db_scan_resident does);
//   B  the host writes the word straight into DEVICE memory through the PCIe BAR (fine-grained allocation), the wave polls it locally.
// The "pong" is the same both ways: a system-scope store to pinned host memory that the host polls.  Prints the round trips.
//   hipcc -O2 --offload-arch=gfx950 scripts/probes/bar_pingpong.hip -o scripts/probes/bar_pingpong.bin && scripts/probes/bar_pingpong.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void echo(const unsigned *ping, unsigned *pong, unsigned last, int inflight)
{
    unsigned seen = 0;
    for (long spin = 0; spin < (1L << 23); spin++) {   // (bounded: a few seconds)
        const unsigned v = __hip_atomic_load(ping, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v != seen) {
            seen = v;
            __hip_atomic_store(pong, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == last) return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static int run(const char *name, volatile unsigned *ping_host_view, unsigned *ping_dev_view, volatile unsigned *pong_h, unsigned *pong_d, int n)
{
    *ping_host_view = 0;
    *pong_h = 0;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, s, ping_dev_view, pong_d, (unsigned)n, 1);
    std::vector<double> t(n);
    for (int i = 1; i <= n; i++) {
        const auto a = std::chrono::steady_clock::now();
        __atomic_store_n((unsigned *)ping_host_view, (unsigned)i, __ATOMIC_RELEASE);
        __builtin_ia32_sfence();     // the BAR mapping is write-combining: without the fence the store may sit in the CPU's WC buffer
        long spins = 0;
        while (__atomic_load_n((unsigned *)pong_h, __ATOMIC_ACQUIRE) != (unsigned)i)
            if (++spins > 400000000L) { std::printf("%s: no echo of ping %d (pong = %u)\n", name, i, *pong_h); return 1; }
        t[i - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
    }
    CK(hipStreamSynchronize(s));
    std::sort(t.begin(), t.end());
    std::printf("%-52s round trip: p50 %.2f us  min %.2f  p99 %.2f   (n = %d)\n", name, t[n / 2], t[0], t[n * 99 / 100], n);
    CK(hipStreamDestroy(s));
    return 0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    int large_bar = -1;
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    std::printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    unsigned *h = nullptr, *hd = nullptr;
    CK(hipHostMalloc((void **)&h, 4096, hipHostMallocDefault));
    CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    const int n = 5000;
    if (run("A  ping in pinned host memory (GPU polls over PCIe)", h, hd, h + 64, hd + 64, n)) return 1;
    // B: device memory written by the host
    unsigned *d = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&d, 4096, hipDeviceMallocFinegrained);
    std::printf("hipExtMallocWithFlags(fine-grained) -> %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { CK(hipMalloc((void **)&d, 4096)); std::printf("(falling back to hipMalloc)\n"); }
    CK(hipMemset(d, 0, 4096));
    CK(hipDeviceSynchronize());
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1) == 0) {
        *(volatile unsigned *)d = 0;       // faults unless the allocation is mapped for the host
        std::printf("host store to device memory: ok\n");
        if (run("B  ping in device memory (host writes through the BAR)", d, d, h + 64, hd + 64, n)) return 1;
    } else {
        std::printf("host store to device memory: FAULT (not host-accessible)\n");
    }
    unsigned *d2 = nullptr;
    CK(hipMalloc((void **)&d2, 4096));
    CK(hipMemset(d2, 0, 4096));
    CK(hipDeviceSynchronize());
    if (sigsetjmp(jb, 1) == 0) {
        *(volatile unsigned *)d2 = 0;
        std::printf("host store to plain hipMalloc memory: ok\n");
        if (run("B' ping in plain hipMalloc memory", d2, d2, h + 64, hd + 64, n)) return 1;
    } else {
        std::printf("host store to plain hipMalloc memory: FAULT\n");
    }
    return 0;
}
