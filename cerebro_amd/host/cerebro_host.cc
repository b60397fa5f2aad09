// cerebro_host.cc -- see cerebro_host.h.  Host bookkeeping only; all arithmetic of the hot path is behind the C ABI.
#include "cerebro_host.h"
#include "state_json.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <thread>
#include <unordered_map>

namespace cerebro_hip {

Cerebro::Cerebro(int descriptor_size, int device, int64_t capacity_hint) : D_(descriptor_size)
{
    chip_dot_params_default(&params);
    status_ = chip_create(&ctx_, descriptor_size, capacity_hint, device, 0, 1);
    if (status_ != CHIP_OK) ctx_ = nullptr;
}

Cerebro::Cerebro(int descriptor_size, const std::vector<int> &devices, int64_t capacity_hint, uint32_t create_flags) : D_(descriptor_size)
{
    chip_dot_params_default(&params);
    std::vector<int32_t> dev(devices.begin(), devices.end());
    status_ = chip_create_multi(&ctx_, descriptor_size, capacity_hint, dev.data(), (int32_t)dev.size(), create_flags);
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
    // A checkpoint holds Eigen's FullPrecision text: 15 significant digits (RawFileIO.cpp:330-459).  Float32 descriptors (every
    // Keras model of the reference's server) come back as doubles that are NOT float32-representable but lie within the print
    // precision of one: rounding them restores the original float rows exactly, so the loaded DB selects the same candidates as
    // the live run that saved it.  Genuinely float64 descriptors (ReljaNetVLAD, whole_image_desc_compute_server.py:148-149) are
    // not near any float32: they are appended as they are and an empty DB becomes a double-row DB -- the same storage the live
    // appends of those descriptors choose (chip_create "decided by the data").  The file decides, value by value.
    bool float32_text = true;
    for (double d : sd.desc) {
        const double r = (double)(float)d;
        if (!(std::fabs(d - r) <= 1e-14 * std::fabs(d))) { float32_text = false; break; }   // 15 digits: <= 5e-15 relative; NaN fails too
    }
    status_ = chip_db_append_f64(ctx_, sd.desc.data(), (int64_t)sd.stampNSec.size(), float32_text ? CHIP_APPEND_ALLOW_ROUNDING : 0u, &first);
    if (status_ != CHIP_OK) { error_ = chip_strerror(status_); return -1; }
    for (uint64_t ns : sd.all_stampNSec) {        // every DataNode of the checkpoint is a data_map entry (DataManager.cpp:1230-1290 rebuilds it)
        Time t;
        t.sec = (uint32_t)(ns / 1000000000ull);
        t.nsec = (uint32_t)(ns % 1000000000ull);
        data_map_insert(t);
    }
    std::lock_guard<std::mutex> lk(m_wholeImageComputedList);
    for (uint64_t ns : sd.stampNSec) {
        Time t;
        t.sec = (uint32_t)(ns / 1000000000ull);   // ros::Time().fromNSec
        t.nsec = (uint32_t)(ns % 1000000000ull);
        wholeImageComputedList.push_back(t);
    }
    return (int64_t)sd.stampNSec.size();
}

void Cerebro::data_map_insert(const Time &stamp)
{
    std::lock_guard<std::mutex> lk(m_data_map);
    const uint64_t k = ((uint64_t)stamp.sec << 32) | stamp.nsec;
    if (!data_map_.empty() && k <= data_map_.back()) data_map_sorted_ = false;   // frames normally arrive in time order
    data_map_.push_back(k);
}

void Cerebro::set_frame_index_map(const std::vector<int64_t> &row_to_frame)
{
    std::lock_guard<std::mutex> lk(m_data_map);
    row_to_frame_ = row_to_frame;
}

int64_t Cerebro::data_map_size() const
{
    std::lock_guard<std::mutex> lk(m_data_map);
    if (!data_map_sorted_) {
        std::sort(data_map_.begin(), data_map_.end());
        data_map_.erase(std::unique(data_map_.begin(), data_map_.end()), data_map_.end());
        data_map_sorted_ = true;
    }
    return (int64_t)data_map_.size();
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

// index.search(1, X_raw, 5, distances, labels) for n_rows query rows against the index prefix [0, ntotal)
bool Cerebro::index_search(int64_t ntotal, const int64_t *rows, int n_rows, float *distances, int64_t *labels)
{
    constexpr int K = 5;
    for (int q0 = 0; q0 < n_rows; q0 += CHIP_MAX_NQ) {
        const int nq = n_rows - q0 < CHIP_MAX_NQ ? n_rows - q0 : CHIP_MAX_NQ;   // one pass of the prefix serves up to 4 queries
        double s[CHIP_MAX_NQ * K];
        status_ = chip_query_rows(ctx_, ntotal, rows + q0, nq, K, s, labels + (size_t)q0 * K);
        if (status_ != CHIP_OK) return false;
        for (int i = 0; i < nq * K; i++) distances[(size_t)q0 * K + i] = (float)s[i];   // faiss distances are float
    }
    return true;
}

int Cerebro::faiss__naive_loopcandidate_generator_once(int64_t l)
{
    if (!ctx_) return 0;
    constexpr int start_adding_descriptors_to_index_after = 150, LOCALITY_THRESH = 12;   // Cerebro.cpp:375-376
    const float DOT_PROD_THRESH = 0.9f;                                                   // :377
    if (l < 0) l = wholeImageComputedList_size();                                         // :401
    status_ = CHIP_OK;
    if (l - naive_last_l_ < 3) return 0;                                                  // :403-407
    if (l > start_adding_descriptors_to_index_after)                                      // :415-433 (the rows are already on the device)
        naive_l_last_added_to_index_ = l - start_adding_descriptors_to_index_after;
    const int64_t ntotal = naive_l_last_added_to_index_;
    std::vector<float> tmp_;
    std::vector<int> tmp_i;
    const int64_t n_new = l - naive_last_l_;
    if (ntotal >= 5 && n_new == 3) {                       // :451; any other count can never satisfy _n == 3 (:476)
        int64_t rows[3] = {naive_last_l_, naive_last_l_ + 1, naive_last_l_ + 2}, labels[15];
        float distances[15];
        if (!index_search(ntotal, rows, 3, distances, labels)) return 0;
        for (int i = 0; i < 3; i++) { tmp_.push_back(distances[5 * i]); tmp_i.push_back((int)labels[5 * i]); }   // :471-472
    }
    int pushed = 0;
    const int _n = (int)tmp_.size();
    if (_n == 3 && tmp_[_n - 1] > DOT_PROD_THRESH && std::abs(tmp_i[0] - tmp_i[1]) < LOCALITY_THRESH &&
        std::abs(tmp_i[0] - tmp_i[2]) < LOCALITY_THRESH) {                                // :476
        const Time a = wholeImageComputedList_at((int)l - 1), b = wholeImageComputedList_at(tmp_i[2]);
        std::lock_guard<std::mutex> lk(m_foundLoops);                                     // :483-484
        foundLoops.push_back(std::make_tuple(a, b, (double)tmp_[2]));
        pushed = 1;
    }
    naive_last_l_ = l;                                                                    // :488
    return pushed;
}

int Cerebro::faiss_clique_loopcandidate_generator_once(int64_t l)
{
    if (!ctx_) return 0;
    constexpr int start_adding_descriptors_to_index_after = 150, K_NEAREST_NEIGHBOURS = 5, LOCALITY = 7,
                  reset_accumulation_every_n_frames = 4;                                  // Cerebro.cpp:515-519
    const double DOT_PROD_THRESH = 0.85;
    if (l < 0) l = wholeImageComputedList_size();                                         // :542
    status_ = CHIP_OK;
    if (l <= clique_last_l_) return 0;                                                    // :543-547
    if (l > start_adding_descriptors_to_index_after)                                      // :558-582
        clique_l_last_added_to_index_ = l - start_adding_descriptors_to_index_after;
    const int64_t ntotal = clique_l_last_added_to_index_;
    int pushed = 0;
    if (ntotal >= K_NEAREST_NEIGHBOURS) {                                                 // :596 (ntotal is constant over the l_i loop)
        // the searches of one iteration are independent of `retained`: run them first, up to 4 per pass of the prefix
        const int n_new = (int)(l - clique_last_l_);
        std::vector<int64_t> rows((size_t)n_new), labels((size_t)n_new * K_NEAREST_NEIGHBOURS);
        std::vector<float> distances((size_t)n_new * K_NEAREST_NEIGHBOURS);
        for (int i = 0; i < n_new; i++) rows[(size_t)i] = clique_last_l_ + i;
        if (!index_search(ntotal, rows.data(), n_new, distances.data(), labels.data())) return 0;
        const Time t_curr = wholeImageComputedList_at((int)l - 1);
        for (int i = 0; i < n_new; i++) {
            const int64_t l_i = rows[(size_t)i];
            for (int g = 0; g < K_NEAREST_NEIGHBOURS; g++) {                              // :625-650
                const float d = distances[(size_t)i * K_NEAREST_NEIGHBOURS + g];
                const int64_t label = labels[(size_t)i * K_NEAREST_NEIGHBOURS + g];
                if (d < DOT_PROD_THRESH) break;
                int64_t duplicate = -1;
                for (auto ity = retained_.begin(); ity != retained_.end(); ity++)
                    if ((ity->first - label) < LOCALITY) { duplicate = ity->first; break; }   // :634 (signed, as upstream)
                if (duplicate != -1) retained_[duplicate]++;
                else retained_[label] = 1;
            }
            if (retained_.size() > 0 && l_i % reset_accumulation_every_n_frames == 0) {   // :653
                auto push = [&](int64_t prev) {
                    const Time b = wholeImageComputedList_at((int)prev);
                    std::lock_guard<std::mutex> lk(m_foundLoops);
                    foundLoops.push_back(std::make_tuple(t_curr, b, 0.9));                // :680-683 / :695-697
                    pushed++;
                };
                if (retained_.size() == 1) push(retained_.begin()->first);                // :672
                if (retained_.size() > 1) {                                               // :686
                    const int percent = (int)(100. / retained_.size());
                    for (auto ity = retained_.begin(); ity != retained_.end(); ity++)
                        if ((rand_source ? rand_source() : std::rand()) % 100 < percent) push(ity->first);
                }
                retained_.clear();                                                        // :703
            }
        }
    }
    clique_last_l_ = l;                                                                   // :710
    return pushed;
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
    // stamp -> first row carrying it (one pass; a 1M-row DB made the former linear search per loop quadratic)
    std::unordered_map<uint64_t, long> first_row;
    first_row.reserve(stamps.size());
    for (size_t i = 0; i < stamps.size(); i++) first_row.emplace(((uint64_t)stamps[i].sec << 32) | stamps[i].nsec, (long)i);
    // global_a / global_b (Cerebro.cpp:1142-1143): the frame's rank in data_map when the frames are known, else the DB row
    (void)data_map_size();   // sorts + dedups the registered frame stamps
    std::vector<uint64_t> frames;
    std::vector<int64_t> r2f;
    {
        std::lock_guard<std::mutex> lk(m_data_map);
        frames = data_map_;
        r2f = row_to_frame_;
    }
    auto index_of = [&](const Time &t) {
        const uint64_t k = ((uint64_t)t.sec << 32) | t.nsec;
        auto it = first_row.find(k);
        const long row = it == first_row.end() ? -1L : it->second;
        if (row >= 0 && (size_t)row < r2f.size()) return (long)r2f[(size_t)row];
        if (!frames.empty()) {
            auto f = std::lower_bound(frames.begin(), frames.end(), k);
            if (f != frames.end() && *f == k) return (long)(f - frames.begin());   // std::distance(begin, find(t))
        }
        return row;
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

float StaticTheiaPoseComputeICP::P3P_ICP(chip_ctx *ctx, const std::vector<std::array<double, 3>> &uv_X, const std::vector<std::array<double, 3>> &uvd_Y,
                                         double uvd_T_uv[16], std::string &p3p__msg, const chip_ransac_params *params)
{
    if (uv_X.size() < 20) return -1;  // DlsPnpWithRansac.cpp:19-22
    p3p__msg = "";
    if (!ctx || uv_X.size() != uvd_Y.size()) return -1;
    chip_ransac_params p;
    if (params) p = *params; else chip_icp_params_default(&p);  // :88-93
    float confidence = 0.f;
    chip_ransac_summary s;
    const int st = chip_icp_ransac(ctx, &uv_X[0][0], &uvd_Y[0][0], (int32_t)uv_X.size(), &p, uvd_T_uv, &confidence, nullptr, &s);
    if (st != CHIP_OK) { p3p__msg = std::string("chip_icp_ransac: ") + chip_strerror(st); return -1; }
    p3p__msg += "ICP Ransac;     #iterations=" + std::to_string(s.n_iterations) + "    confidence=" + std::to_string(confidence);  // :115-118
    return confidence;  // :121
}

void matrix4_inverse_rigid(const double T[16], double I[16])
{
    // [R t; 0 1]^-1 = [R^T  -R^T t; 0 1]   (the reference calls Eigen's general Matrix4d::inverse(); identical up to rounding
    // for the proper rigid transforms that reach this code)
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) I[4 * c + r] = T[4 * r + c];
    for (int r = 0; r < 3; r++) I[12 + r] = -(I[r] * T[12] + I[4 + r] * T[13] + I[8 + r] * T[14]);
    I[3] = I[7] = I[11] = 0.0;
    I[15] = 1.0;
}

void matrix4_mul(const double A[16], const double B[16], double C[16])
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += A[4 * k + r] * B[4 * c + k];
            C[4 * c + r] = s;
        }
}

void matrix4_to_rawyprt(const double T[16], double ypr[3], double t[3])
{
    // n, o, a = columns 0, 1, 2 of R (PoseManipUtils.cpp:150-157)
    const double n0 = T[0], n1 = T[1], n2 = T[2], o0 = T[4], o1 = T[5], a0 = T[8], a1 = T[9];
    const double y = std::atan2(n1, n0);
    const double p = std::atan2(-n2, n0 * std::cos(y) + n1 * std::sin(y));
    const double r = std::atan2(a0 * std::sin(y) - a1 * std::cos(y), -o0 * std::sin(y) + o1 * std::cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;  // :162
    t[0] = T[12]; t[1] = T[13]; t[2] = T[14];
}

bool ProcessedLoopCandidate::makeLoopEdgeMsg(LoopEdgePOD &msg) const
{
    if (!isSet_3d2d__2T1) return false;  // ProcessedLoopCandidate.cpp:18-23
    msg = make_loop_edge(t_node_1, t_node_2, _3d2d__2T1.data(), _3d2d__2T1__ransac_confidence, idx_from_datamanager_1, idx_from_datamanager_2);
    return true;
}

static double inf_norm3(const double v[3])
{
    double m = std::fabs(v[0]);
    if (std::fabs(v[1]) > m) m = std::fabs(v[1]);
    if (std::fabs(v[2]) > m) m = std::fabs(v[2]);
    return m;
}

bool ProcessedLoopCandidate::makeLoopEdgeMsgWithConsistencyCheck(LoopEdgePOD &msg)
{
    if (opX_b_T_a.size() != 3) return false;  // :42-46
    // ros::Duration diff = node_1->getT() - node_2->getT(); if( abs(diff.sec) < 10 ) return false;  (:49-56)
    // ros::Duration is normalised to sec = floor(seconds), 0 <= nsec < 1e9
    int64_t dns = ((int64_t)t_node_1.sec - (int64_t)t_node_2.sec) * 1000000000ll + ((int64_t)t_node_1.nsec - (int64_t)t_node_2.nsec);
    int64_t dsec = dns / 1000000000ll;
    if (dns % 1000000000ll < 0) dsec -= 1;
    if ((dsec < 0 ? -dsec : dsec) < 10) return false;

    const double *op1 = opX_b_T_a[0].data(), *op2 = opX_b_T_a[1].data(), *icp = opX_b_T_a[2].data();
    double op1_inv[16], op2_inv[16], d12[16], d1i[16], d2i[16];
    matrix4_inverse_rigid(op1, op1_inv);
    matrix4_inverse_rigid(op2, op2_inv);
    matrix4_mul(op1_inv, op2, d12);   // op1_m_op2 (:64)
    matrix4_mul(op1_inv, icp, d1i);   // op1_m_icp (:65)
    matrix4_mul(op2_inv, icp, d2i);   // op2_m_icp (:66)
    double y12[3], t12[3], y1i[3], t1i[3], y2i[3], t2i[3];
    matrix4_to_rawyprt(d12, y12, t12);
    matrix4_to_rawyprt(d1i, y1i, t1i);
    matrix4_to_rawyprt(d2i, y2i, t2i);
    const bool is_consistent_ypr = inf_norm3(y12) < 5.0 && inf_norm3(y1i) < 5.0 && inf_norm3(y2i) < 5.0;  // :75-79
    const bool is_consistent_tr = inf_norm3(t1i) < .2 && inf_norm3(t1i) < .2 && inf_norm3(t2i) < .2;       // :81-85 (sic: op1_m_icp twice)
    if (pf_matches > 800 && (is_consistent_ypr && is_consistent_tr)) {  // :112
        _3d2d__2T1 = opX_b_T_a[0];
        isSet_3d2d__2T1 = true;
        float g = opX_goodness[0];
        if (opX_goodness[1] > g) g = opX_goodness[1];
        if (opX_goodness[2] > g) g = opX_goodness[2];
        _3d2d__2T1__ransac_confidence = g;  // :116
        return makeLoopEdgeMsg(msg);
    }
    return false;
}

bool compute_three_way_pose(chip_ctx *ctx, const PosePairInput &in, ProcessedLoopCandidate &pc, uint64_t seed)
{
    chip_ransac_params pp, pi;
    chip_ransac_params_default(&pp);
    chip_icp_params_default(&pi);
    if (seed) { pp.seed = seed; pi.seed = seed ^ 0x9E3779B97F4A7C15ull; }
    std::array<double, 16> op1{}, op2_a_T_b{}, op2{}, icp{};
    // The three estimations of an image pair are independent.  P3P_ICP (:1629) is enqueued first on the ICP stream; the two PNP
    // calls (:1518 a->b, :1572 b->a; seeds seed, seed+1) share one pair of launches; the ICP result is collected afterwards --
    // its kernel has long finished underneath the PnP kernels.
    float g1 = -1.f, g2 = -1.f, g3 = -1.f;
    const bool icp_ok = in.uv_X.size() >= 20 && in.uv_X.size() == in.uvd_Y.size() &&
                        chip_icp_ransac_enqueue(ctx, &in.uv_X[0][0], &in.uvd_Y[0][0], (int32_t)in.uv_X.size(), &pi) == CHIP_OK;
    if (in.world_point_uv.size() >= 20 && in.world_point_uv_d.size() >= 20 && in.world_point_uv.size() == in.feature_position_uv_d.size() &&
        in.world_point_uv_d.size() == in.feature_position_uv.size()) {   // the < 20 guard of PNP (DlsPnpWithRansac.cpp:136-139)
        const double *Xs[2] = {&in.world_point_uv[0][0], &in.world_point_uv_d[0][0]};
        const double *uvs[2] = {&in.feature_position_uv_d[0][0], &in.feature_position_uv[0][0]};
        const int32_t Ns[2] = {(int32_t)in.world_point_uv.size(), (int32_t)in.world_point_uv_d.size()};
        const uint64_t seeds[2] = {pp.seed, pp.seed + 1};
        double T2[32];
        float conf[2] = {0.f, 0.f};
        if (chip_pnp_ransac_batch(ctx, 2, Xs, uvs, Ns, &pp, seeds, T2, conf, nullptr, nullptr) == CHIP_OK) {
            for (int i = 0; i < 16; i++) { op1[i] = T2[i]; op2_a_T_b[i] = T2[16 + i]; }
            g1 = conf[0];
            g2 = conf[1];
        }
    }
    matrix4_inverse_rigid(op2_a_T_b.data(), op2.data());                                                                          // :1582
    if (icp_ok) {
        float c3 = 0.f;
        if (chip_icp_ransac_collect(ctx, icp.data(), &c3, nullptr, nullptr) == CHIP_OK) g3 = c3;                                   // :1629
    }
    for (int i = 0; i < 16; i++)  // :1678  op != op  <=> any NaN
        if (op1[i] != op1[i] || op2[i] != op2[i] || icp[i] != icp[i]) return false;
    if (g1 < 0 || g2 < 0 || g3 < 0) return false;  // a too-small set leaves the pose untouched in the reference; treat as no pose
    pc.opX_b_T_a = {op1, op2, icp};       // :1706-1719
    pc.opX_goodness = {g1, g2, g3};
    return true;
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
