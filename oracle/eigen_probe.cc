// eigen_probe.cc -- bench.py's cpu_baseline with the REAL Eigen, when the machine has one.
// Test / baseline infrastructure, never on the product path.  Built at run time by bench.py only if <Eigen/Dense> is found
// (it is absent from the build image): g++ -O3 -DNDEBUG, no -march -- the reference's flags (CMakeLists.txt:42-44, Release).
// Times the literal statements of /root/reference/src/Cerebro.cpp:1026-1043 on a D x cols MatrixXd filled with float32-valued
// unit-norm-ish columns: three separate products, three maxCoeff, the last-index loop.  Prints one line:
//   eigen <EIGEN_WORLD>.<MAJOR>.<MINOR> cols <cols> D <D> ticks <n> seconds <s> checksum <x>
#include <Eigen/Dense>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
using namespace Eigen;

// --dump <in.bin> <out.bin>: pins the Eigen-order restatement (oracle/dot_scan.c orc_ref_scan_f64_eigen_gemv3) against the REAL Eigen
// once.  in.bin: i32 D, i32 k, then (k + 3) x D float64 column-major (the columns of M; the last three are the
// queries v, vm, vmm as at Cerebro.cpp:987-989).  out.bin: the literal u, um, umm of :1026-1028, 3 x k float64.
static int dump(const char *in, const char *out)
{
    FILE *f = std::fopen(in, "rb");
    if (!f) return 2;
    int32_t D = 0, k = 0;
    if (std::fread(&D, 4, 1, f) != 1 || std::fread(&k, 4, 1, f) != 1 || D < 1 || k < 1) return 2;
    MatrixXd M(D, k + 3);
    if (std::fread(M.data(), sizeof(double), (size_t)D * (k + 3), f) != (size_t)D * (k + 3)) return 2;
    std::fclose(f);
    const VectorXd v = M.col(k + 2), vm = M.col(k + 1), vmm = M.col(k);
    VectorXd u = v.transpose() * M.leftCols(k);                     // :1026
    VectorXd um = vm.transpose() * M.leftCols(k);                   // :1027
    VectorXd umm = vmm.transpose() * M.leftCols(k);                 // :1028
    FILE *o = std::fopen(out, "wb");
    if (!o) return 2;
    std::fwrite(u.data(), sizeof(double), (size_t)k, o);
    std::fwrite(um.data(), sizeof(double), (size_t)k, o);
    std::fwrite(umm.data(), sizeof(double), (size_t)k, o);
    std::fclose(o);
    std::printf("eigen %d.%d.%d dumped u, um, umm of %d columns x %d\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION, k, D);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 3 && std::string(argv[1]) == "--dump") return dump(argv[2], argv[3]);
    const int D = argc > 1 ? std::atoi(argv[1]) : 4096;
    const int cols = argc > 2 ? std::atoi(argv[2]) : 20000;
    const double budget = argc > 3 ? std::atof(argv[3]) : 10.0;
    MatrixXd M = MatrixXd::Zero(D, cols + 3);                       // :946 (capacity = what is scanned here)
    uint64_t x = 0x9E3779B97F4A7C15ull;
    const double scale = 1.0 / std::sqrt((double)D / 3.0);
    for (int c = 0; c < cols + 3; c++)
        for (int r = 0; r < D; r++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            M(r, c) = (double)(float)(((double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0) * scale);
        }
    const VectorXd v = M.col(cols + 2), vm = M.col(cols + 1), vmm = M.col(cols);   // :987-989
    const int k = cols;
    double check = 0.0;
    int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    double dt = 0.0;
    do {
        VectorXd u = v.transpose() * M.leftCols(k);                 // :1026
        VectorXd um = vm.transpose() * M.leftCols(k);               // :1027
        VectorXd umm = vmm.transpose() * M.leftCols(k);             // :1028
        const double u_max = u.maxCoeff(), um_max = um.maxCoeff(), umm_max = umm.maxCoeff();   // :1035-1037
        int u_argmax = -1, um_argmax = -1, umm_argmax = -1;
        for (int ii = 0; ii < u.size(); ii++) {                     // :1038-1043
            if (u(ii) == u_max) u_argmax = ii;
            if (um(ii) == um_max) um_argmax = ii;
            if (umm(ii) == umm_max) umm_argmax = ii;
        }
        check += u_max + um_max + umm_max + u_argmax + um_argmax + umm_argmax;
        n++;
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (dt < budget);
    std::printf("eigen %d.%d.%d cols %d D %d ticks %d seconds %.6f checksum %.17g\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION,
                EIGEN_MINOR_VERSION, cols, D, n, dt, check);
    return 0;
}
