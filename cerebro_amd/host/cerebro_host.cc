// cerebro_host.cc -- see cerebro_host.h.  Host bookkeeping only; all arithmetic of the hot path is behind the C ABI.
#include "cerebro_host.h"
#include "state_json.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <sstream>
#include <thread>

namespace cerebro_hip {

Cerebro::Cerebro(int descriptor_size, int device, int64_t capacity_hint) : D_(descriptor_size)
{
    chip_dot_params_default(&params);
    status_ = chip_create(&ctx_, descriptor_size, capacity_hint, device, 0, 1);
    if (status_ != CHIP_OK) ctx_ = nullptr;
}

Cerebro::~Cerebro()
{
    run_thread_disable();
    if (ctx_) chip_destroy(ctx_);
}

bool Cerebro::descriptor_available(const Time &stamp, const double *desc, int n)
{
    if (!ctx_ || n != D_) return false;
    int64_t first = -1;
    status_ = chip_db_append_f64(ctx_, desc, 1, 0, &first);  // float64[] -> device row (Cerebro.cpp:268-274)
    if (status_ != CHIP_OK) return false;
    std::lock_guard<std::mutex> lk(m_wholeImageComputedList);  // Cerebro.cpp:321-326
    wholeImageComputedList.push_back(stamp);
    return (int64_t)wholeImageComputedList.size() == first + 1;
}

int64_t Cerebro::loadStateFromDisk(const std::string &path)
{
    if (!ctx_) return -1;
    StateDescriptors sd;
    if (!load_state_json(path, sd)) { error_ = sd.error; status_ = CHIP_ERR_INVALID_ARG; return -1; }
    if (sd.stampNSec.empty()) return 0;
    if (sd.D != D_) { error_ = "descriptor size in state.json differs from descriptor_size"; status_ = CHIP_ERR_INVALID_ARG; return -1; }
    int64_t first = -1;
    status_ = chip_db_append_f64(ctx_, sd.desc.data(), (int64_t)sd.stampNSec.size(), CHIP_APPEND_ALLOW_ROUNDING, &first);
    if (status_ != CHIP_OK) { error_ = chip_strerror(status_); return -1; }
    std::lock_guard<std::mutex> lk(m_wholeImageComputedList);
    for (uint64_t ns : sd.stampNSec) {
        Time t;
        t.sec = (uint32_t)(ns / 1000000000ull);   // ros::Time().fromNSec
        t.nsec = (uint32_t)(ns % 1000000000ull);
        wholeImageComputedList.push_back(t);
    }
    return (int64_t)sd.stampNSec.size();
}

int Cerebro::wholeImageComputedList_size() const
{
    std::lock_guard<std::mutex> lk(m_wholeImageComputedList);
    return (int)wholeImageComputedList.size();
}

Time Cerebro::wholeImageComputedList_at(int k) const
{
    std::lock_guard<std::mutex> lk(m_wholeImageComputedList);
    return wholeImageComputedList.at((size_t)k);
}

bool Cerebro::descrip_N__dot__descrip_0_N_once(int64_t l, chip_tick_result *detail)
{
    if (!ctx_) return false;
    if (l < 0) l = wholeImageComputedList_size();  // Cerebro.cpp:960
    chip_tick_result r;
    status_ = chip_loop_tick(ctx_, l, &params, &r);  // :962-1056 (skip rule, k = l-50, scan, argmax, accept rule)
    if (detail) *detail = r;
    if (status_ != CHIP_OK || !r.found) return false;
    const Time t_curr = wholeImageComputedList_at((int)r.idx_curr);   // wholeImageComputedList_at(l-1)
    const Time t_prev = wholeImageComputedList_at((int)r.idx_prev);   // wholeImageComputedList_at(u_argmax)
    std::lock_guard<std::mutex> lk(m_foundLoops);                     // :1078-1081
    foundLoops.push_back(std::make_tuple(t_curr, t_prev, r.score));
    return true;
}

void Cerebro::run(double rate_hz)
{
    const auto period = std::chrono::duration<double>(1.0 / rate_hz);  // ros::Rate rate(10), Cerebro.cpp:916
    while (b_run_thread) {
        descrip_N__dot__descrip_0_N_once();
        std::this_thread::sleep_for(period);
    }
}

int Cerebro::foundLoops_count() const
{
    std::lock_guard<std::mutex> lk(m_foundLoops);
    return (int)foundLoops.size();
}

std::tuple<Time, Time, double> Cerebro::foundLoops_i(int i) const
{
    std::lock_guard<std::mutex> lk(m_foundLoops);
    return foundLoops.at((size_t)i);
}

std::string Cerebro::foundLoops_as_JSON() const
{
    std::vector<std::tuple<Time, Time, double>> loops;
    {
        std::lock_guard<std::mutex> lk(m_foundLoops);
        loops = foundLoops;
    }
    std::vector<Time> stamps;
    {
        std::lock_guard<std::mutex> lk(m_wholeImageComputedList);
        stamps = wholeImageComputedList;
    }
    auto index_of = [&](const Time &t) {
        for (size_t i = 0; i < stamps.size(); i++)
            if (stamps[i] == t) return (long)i;
        return -1L;
    };
    std::ostringstream o;
    o.precision(17);
    o << "[";
    for (size_t i = 0; i < loops.size(); i++) {
        const Time &a = std::get<0>(loops[i]), &b = std::get<1>(loops[i]);
        if (i) o << ",";
        o << "{\"time_sec_a\":" << a.sec << ",\"time_nsec_a\":" << a.nsec << ",\"time_sec_b\":" << b.sec << ",\"time_nsec_b\":" << b.nsec
          << ",\"time_double_a\":" << a.toSec() << ",\"time_double_b\":" << b.toSec() << ",\"global_a\":" << index_of(a)
          << ",\"global_b\":" << index_of(b) << ",\"score\":" << std::get<2>(loops[i]) << "}";
    }
    o << "]";
    return o.str();
}

float StaticTheiaPoseCompute::PNP(chip_ctx *ctx, const std::vector<std::array<double, 3>> &w_X,
                                  const std::vector<std::array<double, 2>> &c_uv_normalized, double c_T_w[16], std::string &pnp__msg,
                                  const chip_ransac_params *params, std::vector<uint8_t> *inliers)
{
    if (w_X.size() < 20) return -1;  // DlsPnpWithRansac.cpp:136-139
    pnp__msg = "";
    if (!ctx || w_X.size() != c_uv_normalized.size()) return -1;
    chip_ransac_params p;
    if (params) p = *params; else chip_ransac_params_default(&p);  // :207-212
    float confidence = 0.f;
    chip_ransac_summary s;
    std::vector<uint8_t> mask(w_X.size());
    const auto t0 = std::chrono::steady_clock::now();
    const int st = chip_pnp_ransac(ctx, &w_X[0][0], &c_uv_normalized[0][0], (int32_t)w_X.size(), &p, c_T_w, &confidence, mask.data(), &s);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (st != CHIP_OK) { pnp__msg = std::string("chip_pnp_ransac: ") + chip_strerror(st); return -1; }
    if (inliers) inliers->swap(mask);
    char buf[512];  // :234-236 (the reference pretty-prints the pose with PoseManipUtils; ypr/xyz formatting is out of scope)
    std::snprintf(buf, sizeof buf,
                  "DlsPnpWithRansac (best_rel_pose.b_T_a): t=(%.6f,%.6f,%.6f);    num_iterations=%d  confidence=%f   elapsed_dls_pnp_ransac (ms)=%f;",
                  c_T_w[12], c_T_w[13], c_T_w[14], s.n_iterations, (double)confidence, ms);
    pnp__msg += buf;
    return confidence;  // summary.confidence (:240)
}

void matrix4_to_pose(const double T[16], double position[3], double q[4])
{
    position[0] = T[12]; position[1] = T[13]; position[2] = T[14];
    // Eigen::Quaterniond(Matrix3d): trace branch / largest-diagonal branch; T is column-major: m(r,c) = T[4*c + r]
    auto m = [&](int r, int c) { return T[4 * c + r]; };
    double w, x, y, z;
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        w = 0.5 * t;
        t = 0.5 / t;
        x = (m(2, 1) - m(1, 2)) * t; y = (m(0, 2) - m(2, 0)) * t; z = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        w = (m(k, j) - m(j, k)) * t;
        v[j] = (m(j, i) + m(i, j)) * t;
        v[k] = (m(k, i) + m(i, k)) * t;
        x = v[0]; y = v[1]; z = v[2];
    }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

LoopEdgePOD make_loop_edge(const Time &t_node_1, const Time &t_node_2, const double T[16], float ransac_confidence, int idx1, int idx2)
{
    LoopEdgePOD e;
    e.timestamp0 = t_node_1;  // ProcessedLoopCandidate.cpp:24-25
    e.timestamp1 = t_node_2;
    matrix4_to_pose(T, e.position, e.orientation_xyzw);  // :27-29
    e.weight = ransac_confidence;                         // :31
    e.description = std::to_string(idx1) + "<=>" + std::to_string(idx2) + "    this pose is: " + std::to_string(idx2) + "_T_" + std::to_string(idx1);  // :32-33
    return e;
}

}  // namespace cerebro_hip
