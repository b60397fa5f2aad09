// Microbenchmark: what is the best a pure READ stream reaches on this part?  db_scan_topk is priced against the 8.0 TB/s HBM3E
// spec peak; this probe gives the practical ceiling of a kernel that does nothing but load the same 16.4 GB (1M rows x 16 KiB)
// and fold it into one register, in the access shapes the scan could use:
//   shape 0: grid-stride over 16-B vectors, U independent loads per lane in flight        (the classic copy-kernel read side)
//   shape 1: one wave per 16 KiB row, rows dealt round-robin to waves (db_scan_topk's shape: lane l reads bytes 16l + 1024 j)
//   shape 2: one workgroup per contiguous slab of rows (each wave walks its own contiguous run of rows)
// each with plain and non-temporal loads, 256 / 512 / 1024 threads per workgroup, 1..8 workgroups per CU.
// The buffer holds pseudo-random bits by default ("zero" as 2nd argument fills it with zeros instead: an all-zero stream toggles
// no data lines, which a power-managed part can turn into clock -- not representative of a descriptor DB).
// build: hipcc -O3 --offload-arch=gfx950 hbm_read_probe.hip -o hbm_read_probe.bin ; run: ./hbm_read_probe.bin [GiB] [zero|rand]
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4 *gptr;

template <bool NT> __device__ __forceinline__ f32x4 ld(gptr p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

template <int SHAPE, int U, bool NT>
__global__ void read_all(const f32x4 *__restrict__ src_, size_t n_vec, float *out)
{
    gptr src = (gptr)src_;
    f32x4 acc[U];
    for (int u = 0; u < U; u++) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    if (SHAPE == 0) {
        size_t i = tid;
        for (; i + (U - 1) * nthreads < n_vec; i += U * nthreads) {
            f32x4 v[U];
            for (int u = 0; u < U; u++) v[u] = ld<NT>(src + i + u * nthreads);
            for (int u = 0; u < U; u++) acc[u] += v[u];
        }
        for (; i < n_vec; i += nthreads) acc[0] += ld<NT>(src + i);
    } else {
        // rows of 1024 vectors (16 KiB); a wave reads a row as 16 loads of 64 x 16 B
        const int lane = threadIdx.x & 63;
        const size_t wave = tid >> 6, nwaves = nthreads >> 6, n_rows = n_vec >> 10;
        size_t r0, r1, step;
        if (SHAPE == 1) { r0 = wave; r1 = n_rows; step = nwaves; }
        else { size_t per = (n_rows + nwaves - 1) / nwaves; r0 = wave * per; r1 = r0 + per < n_rows ? r0 + per : n_rows; step = 1; }
        for (size_t r = r0; r < r1; r += step) {
            gptr row = src + (r << 10) + lane;
            for (int j = 0; j < 16; j += U) {
                f32x4 v[U];
                for (int u = 0; u < U; u++) v[u] = ld<NT>(row + (size_t)(j + u) * 64);
                for (int u = 0; u < U; u++) acc[u] += v[u];
            }
        }
    }
    f32x4 s = acc[0];
    for (int u = 1; u < U; u++) s += acc[u];
    float t = s.x + s.y + s.z + s.w;
    if (t == -123456.789f) out[tid & 1023] = t;   // never true (all values are >= 0): keeps the loads alive
}

__global__ void fill_random(unsigned *dst, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        dst[i] = ((unsigned)(z >> 32) & 0x007fffffu) | 0x3c000000u;   // floats of magnitude ~0.01, like unit-norm descriptors
    }
}

template <int SHAPE, int U, bool NT>
static double run(const f32x4 *buf, size_t n_vec, float *out, int block, int grid, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    read_all<SHAPE, U, NT><<<grid, block>>>(buf, n_vec, out);
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) read_all<SHAPE, U, NT><<<grid, block>>>(buf, n_vec, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)n_vec * 16.0 * reps / (ms * 1e-3) / 1e12;
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 15.2587890625;   // 1M rows x 16 KiB
    const size_t n_vec = ((size_t)(gib * 1073741824.0) / 16) & ~(size_t)1023;
    f32x4 *buf; float *out;
    if (hipMalloc(&buf, n_vec * 16) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const bool zero = argc > 2 && argv[2][0] == 'z';
    if (zero) hipMemset(buf, 0, n_vec * 16);
    else fill_random<<<4096, 256>>>((unsigned *)buf, n_vec * 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, %.2f GB per pass, %s data\n", prop.gcnArchName, cus, n_vec * 16.0 / 1e9, zero ? "all-zero" : "pseudo-random");
    const int reps = 6;
    struct Best { double v = 0; char what[128] = ""; } best[3];
    auto note = [&](int shape, double tbs, const char *nm, int block, int per_cu) {
        printf("shape %d %-14s block %4d  %d WG/CU : %.3f TB/s (%.3f of 8.0)\n", shape, nm, block, per_cu, tbs, tbs / 8.0);
        if (tbs > best[shape].v) { best[shape].v = tbs; snprintf(best[shape].what, sizeof best[shape].what, "%s block %d, %d WG/CU", nm, block, per_cu); }
    };
    for (int block : {256, 512, 1024})
        for (int per_cu : {1, 2, 4, 8}) {
            if (block * per_cu > 2048) continue;
            const int grid = cus * per_cu;
            note(0, run<0, 4, false>(buf, n_vec, out, block, grid, reps), "U=4", block, per_cu);
            note(0, run<0, 8, false>(buf, n_vec, out, block, grid, reps), "U=8", block, per_cu);
            note(0, run<0, 8, true>(buf, n_vec, out, block, grid, reps), "U=8 nt", block, per_cu);
            note(1, run<1, 8, false>(buf, n_vec, out, block, grid, reps), "U=8", block, per_cu);
            note(1, run<1, 16, false>(buf, n_vec, out, block, grid, reps), "U=16", block, per_cu);
            note(1, run<1, 16, true>(buf, n_vec, out, block, grid, reps), "U=16 nt", block, per_cu);
            note(2, run<2, 8, true>(buf, n_vec, out, block, grid, reps), "U=8 nt", block, per_cu);
            note(1, run<1, 8, true>(buf, n_vec, out, block, grid, reps), "U=8 nt", block, per_cu);
            note(2, run<2, 16, false>(buf, n_vec, out, block, grid, reps), "U=16", block, per_cu);
            note(2, run<2, 16, true>(buf, n_vec, out, block, grid, reps), "U=16 nt", block, per_cu);
        }
    for (int s = 0; s < 3; s++) printf("BEST shape %d: %.3f TB/s (%.3f of 8.0)  %s\n", s, best[s].v, best[s].v / 8.0, best[s].what);
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
    return 0;
}
