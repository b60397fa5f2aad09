// examples/minimal_loop_detector.cc -- the INTEGRATION.md call sequence as a stand-alone program (no ROS, no Eigen): what the
// patched Cerebro::descrip_N__dot__descrip_0_N loop and StaticTheiaPoseCompute::PNP body do, against libcerebro_hip.so only.
//
//   g++ -O2 -std=c++17 -Iinclude examples/minimal_loop_detector.cc -Lcerebro_amd/lib -lcerebro_hip -Wl,-rpath,$PWD/cerebro_amd/lib -o /tmp/minimal_loop_detector
//
// Feeds a synthetic descriptor stream in which keyframes 400..411 revisit keyframes 100..111, ticks every 3 keyframes like the
// 10 Hz dot-product thread, prints the loop candidates, then verifies one candidate pose with PnP-RANSAC on synthetic
// correspondences.  Exit code 0 iff the revisit was detected and the pose recovered.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "cerebro_hip.h"

#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        const int st_ = (call);                                                                            \
        if (st_ != CHIP_OK) { std::fprintf(stderr, "%s -> %s\n", #call, chip_strerror(st_)); return 2; }   \
    } while (0)

// Usage: minimal_loop_detector [device list]   e.g. "0,1,2,3" = one handle over four GPUs (chip_create_multi; row i of the DB on
// device i % 4, the per-device top-k lists exchanged by RCCL inside the library); a repeated device ("0,0") runs the sharded code
// path on one GPU.  Without an argument: chip_create on device 0.  Everything after the create line is the same either way.
int main(int argc, char **argv)
{
    const int D = 4096, N = 600;
    std::vector<int32_t> devices;
    if (argc > 1)
        for (const char *p = argv[1]; *p;) {
            devices.push_back(std::atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
    std::mt19937 rng(7);
    std::normal_distribution<float> gauss(0.f, 1.f);
    std::vector<std::vector<double>> desc(N, std::vector<double>(D));   // the .srv wire type is float64[] holding float32 values
    for (int i = 0; i < N; i++) {
        std::vector<float> v(D);
        double nrm = 0;
        for (int j = 0; j < D; j++) {
            v[j] = (i >= 400 && i < 412) ? (float)desc[i - 300][j] + 0.003f * gauss(rng) : gauss(rng);   // revisit = noisy copy
            nrm += (double)v[j] * v[j];
        }
        const float inv = (float)(1.0 / std::sqrt(nrm));
        for (int j = 0; j < D; j++) desc[i][j] = (double)(float)(v[j] * inv);                            // unit norm, float32-valued
    }

    chip_ctx *chip = nullptr;
    if (devices.empty()) CHECK(chip_create(&chip, D, /*capacity_hint*/ 29000, /*device*/ 0, /*shard_rank*/ 0, /*shard_count*/ 1));   // Cerebro.cpp:946
    else CHECK(chip_create_multi(&chip, D, 29000, devices.data(), (int32_t)devices.size(), 0));
    chip_dot_params prm;
    chip_dot_params_default(&prm);                                                                               // :912-914

    int n_loops = 0, first_prev = -1;
    for (int l = 3; l <= N; l += 3) {                                       // l = wholeImageComputedList_size() at this iteration
        for (int s = (int)chip_db_size(chip); s < l; s++) {                 // :1005-1006
            int64_t row = -1;
            CHECK(chip_db_append_f64(chip, desc[s].data(), 1, 0, &row));
            if (row != s) return 3;
        }
        chip_tick_result r;
        CHECK(chip_loop_tick(chip, l, &prm, &r));                           // :962-1056
        if (r.status == CHIP_TICK_SCANNED && r.found) {
            std::printf("loop candidate: keyframe %lld <-> %lld  score %.6f\n", (long long)r.idx_curr, (long long)r.idx_prev, r.score);
            if (n_loops++ == 0) first_prev = (int)r.idx_prev;
        }
    }

    // pose verification (DlsPnpWithRansac.cpp:192-240): 300 correspondences, 20 % outliers, known pose
    const int M = 300;
    std::vector<double> X(3 * M), uv(2 * M);
    const double yaw = 0.2, cy = std::cos(yaw), sy = std::sin(yaw), t[3] = {0.3, -0.1, 0.2};
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int i = 0; i < M; i++) {
        const double x = 2 * U(rng), y = 1.5 * U(rng), z = 6 + 4 * U(rng);
        X[3 * i] = x; X[3 * i + 1] = y; X[3 * i + 2] = z;
        const double px = cy * x + sy * z + t[0], py = y + t[1], pz = -sy * x + cy * z + t[2];
        uv[2 * i] = px / pz; uv[2 * i + 1] = py / pz;
        if (i % 5 == 0) { uv[2 * i] = 0.5 * U(rng); uv[2 * i + 1] = 0.4 * U(rng); }
    }
    chip_ransac_params rp;
    chip_ransac_params_default(&rp);                                        // .03 / .7 / 50 / 5 / use_mle, 15-point samples
    rp.seed = 12345;
    double T[16];
    float confidence = 0;
    chip_ransac_summary sum;
    std::vector<uint8_t> mask(M);
    CHECK(chip_pnp_ransac(chip, X.data(), uv.data(), M, &rp, T, &confidence, mask.data(), &sum));
    std::printf("PnP: %d iterations, %d inliers of %d, confidence %.3f, t = (%.3f %.3f %.3f), R00 = %.4f (cos yaw = %.4f)\n",
                sum.n_iterations, sum.n_inliers, M, confidence, T[12], T[13], T[14], T[0], cy);
    chip_destroy(chip);

    const bool pose_ok = std::fabs(T[0] - cy) < 1e-3 && std::fabs(T[12] - t[0]) < 1e-2 && sum.n_inliers > 200;
    return (n_loops > 0 && first_prev >= 100 && first_prev < 112 && pose_ok) ? 0 : 1;
}
