// examples/sync_tick_latency.cc -- what ONE synchronous chip_loop_tick costs a C / C++ caller (the live system's mode: the reference's
// dot-product thread runs one tick at a time at 10 Hz, Cerebro.cpp:916), without Python's ctypes in the measurement: mean / median /
// minimum over n ticks, and the same tick split into its enqueue and collect halves.  Links libcerebro_hip.so only.
//   sync_tick_latency [rows = 10000] [n = 2000] [device = 0] [pause_ms = 0] [sleep | spin]
// pause_ms > 0: the host sleeps that long between ticks (100 = the reference's 10 Hz producer) -- what a tick costs when the GPU has
// been left alone since the last one, not back to back.  "spin": the calling thread BUSY-WAITS through the pause instead of sleeping --
// the GPU idles just as long, but the CPU core stays awake and warm: separates what the idle GPU costs from what the sleeping host costs.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "cerebro_hip.h"

#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        const int st_ = (call);                                                                            \
        if (st_ != CHIP_OK) { std::fprintf(stderr, "%s -> %s\n", #call, chip_strerror(st_)); return 2; }   \
    } while (0)

static double us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b)
{
    return std::chrono::duration<double, std::micro>(b - a).count();
}

int main(int argc, char **argv)
{
    const int64_t rows = argc > 1 ? std::atoll(argv[1]) : 10000;
    const int n = argc > 2 ? std::atoi(argv[2]) : 2000, device = argc > 3 ? std::atoi(argv[3]) : 0;
    const int pause_ms = argc > 4 ? std::atoi(argv[4]) : 0;
    const bool spin = argc > 5 && argv[5][0] == 's' && argv[5][1] == 'p';
    auto pause = [&] {
        if (pause_ms <= 0) return;
        if (!spin) { std::this_thread::sleep_for(std::chrono::milliseconds(pause_ms)); return; }
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(pause_ms);
        while (std::chrono::steady_clock::now() < until) {}
    };
    const int D = 4096, span = 150;
    chip_ctx *chip = nullptr;
    CHECK(chip_create(&chip, D, rows + 3 * span + 100, device, 0, 1));
    CHECK(chip_db_append_synthetic(chip, rows + 3 * span + 60, 20190412ull, nullptr, nullptr, nullptr, 0));
    chip_dot_params prm;
    chip_dot_params_default(&prm);
    prm.min_new = -(1 << 30);                    // every call scans, whatever the previous l was
    chip_tick_result r;
    for (int i = 0; i < 50; i++) CHECK(chip_loop_tick(chip, rows + 50 + 3 * (i % span), &prm, &r));
    std::vector<double> t(n), te(n), tc(n);
    for (int i = 0; i < n; i++) {
        const int64_t l = rows + 50 + 3 * (i % span);
        pause();
        const auto a = std::chrono::steady_clock::now();
        CHECK(chip_loop_tick(chip, l, &prm, &r));
        t[i] = us(a, std::chrono::steady_clock::now());
    }
    const int n_split = pause_ms > 0 ? (n < 40 ? n : 40) : n;   // (paced: forty ticks are enough for the split)
    te.resize(n_split); tc.resize(n_split);
    for (int i = 0; i < n_split; i++) {
        const int64_t l = rows + 50 + 3 * (i % span);
        pause();
        const auto a = std::chrono::steady_clock::now();
        CHECK(chip_loop_tick_enqueue(chip, l, &prm, 0));
        const auto b = std::chrono::steady_clock::now();
        CHECK(chip_loop_tick_collect(chip, 0, &r));
        te[i] = us(a, b);
        tc[i] = us(b, std::chrono::steady_clock::now());
    }
    auto stat = [&](std::vector<double> v, const char *name) {
        std::sort(v.begin(), v.end());
        double s = 0;
        for (double x : v) s += x;
        std::printf("\"%s\": {\"mean_us\": %.2f, \"p50_us\": %.2f, \"min_us\": %.2f, \"p99_us\": %.2f}", name, s / v.size(), v[v.size() / 2], v[0], v[v.size() * 99 / 100]);
    };
    std::printf("{\"rows\": %lld, \"n\": %d, \"pause_ms\": %d, \"pause\": \"%s\", ", (long long)rows, n, pause_ms, spin ? "spin" : "sleep");
    stat(t, "sync_tick"); std::printf(", "); stat(te, "enqueue"); std::printf(", "); stat(tc, "collect");
    std::printf(", \"status\": %d}\n", r.status);
    chip_destroy(chip);
    return r.status == CHIP_TICK_SCANNED ? 0 : 1;
}
