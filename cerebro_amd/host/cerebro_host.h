// cerebro_host.h -- ROS-free C++ host side above the C ABI (include/cerebro_hip.h).
//
// Mirrors, name for name, the part of the reference's `class Cerebro` that touches the hot path
// (/root/reference/src/Cerebro.h:89-174) and `StaticTheiaPoseCompute::PNP`
// (/root/reference/src/DlsPnpWithRansac.h:175-177), so that the reference's call sites keep reading the same:
//   wholeImageComputedList_size/_at   Cerebro.h:105-106, Cerebro.cpp:309-326
//   foundLoops_count/_i/_as_JSON      Cerebro.h:152-154, Cerebro.cpp:1113-1164
//   descrip_N__dot__descrip_0_N       Cerebro.cpp:903-1103  (run() :350-357 calls it)
// ros::Time is replaced by the POD `Time`; Eigen types by plain arrays (column-major Matrix4d layout).
// Nothing here computes: every dot product, top-k, accept decision and PnP hypothesis runs in libcerebro_hip.so.
#pragma once
#include <array>
#include <atomic>
#include <cstdint>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/cerebro_hip.h"

namespace cerebro_hip {

struct Time {  // ros::Time
    uint32_t sec = 0, nsec = 0;
    double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
    bool operator==(const Time &o) const { return sec == o.sec && nsec == o.nsec; }
};

// POD mirror of cerebro/LoopEdge (msg/LoopEdge.msg:1-5); quaternion as geometry_msgs/Pose (x,y,z,w)
struct LoopEdgePOD {
    Time timestamp0, timestamp1;
    double position[3];
    double orientation_xyzw[4];
    float weight;
    std::string description;
};

class Cerebro {
public:
    // descriptor_size: learnt by the reference from a zero-image service call (Cerebro.cpp:75-120)
    explicit Cerebro(int descriptor_size, int device = 0, int64_t capacity_hint = 29000 /* Cerebro.cpp:946 */);
    // The same object over several GPUs of this node (chip_create_multi): the DB is row-sharded over `devices`, every method
    // below is unchanged for the caller -- the dot-product thread stays ONE thread of ONE process as in the reference
    // (src/cerebro_node.cpp:499); the per-shard top-k lists are exchanged inside the library (RCCL all-gather).  A device
    // named twice selects the device-copy exchange (lets a 1-GPU machine run the sharded code path).
    Cerebro(int descriptor_size, const std::vector<int> &devices, int64_t capacity_hint = 29000, uint32_t create_flags = 0);
    ~Cerebro();
    Cerebro(const Cerebro &) = delete;
    Cerebro &operator=(const Cerebro &) = delete;
    bool ok() const { return ctx_ != nullptr; }
    int last_status() const { return status_; }
    chip_ctx *ctx() { return ctx_; }

    // ---- producer side (desc_th): Cerebro.cpp:268-275 = setWholeImageDescriptor + wholeImageComputedList_pushback
    bool descriptor_available(const Time &stamp, const double *desc, int n);

    // Cold start from a checkpoint written by DataManager::saveStateToDisk: every node that carries a descriptor is
    // appended in file (= time-stamp) order, as Cerebro rebuilds wholeImageComputedList after loadStateFromDisk
    // (Cerebro.cpp:133-161).  Eigen's FullPrecision text keeps 15 significant digits, so a float32-valued descriptor
    // comes back perturbed at ~1e-16 relative; the append therefore rounds to the nearest float32, which recovers
    // the original value exactly.  Returns the number of descriptors loaded, or -1 (see last_status()/last_error()).
    int64_t loadStateFromDisk(const std::string &state_json_path);
    const std::string &last_error() const { return error_; }

    // ---- what foundLoops_as_JSON's global_a / global_b refer to.  The reference reports std::distance(data_map->begin(),
    // data_map->find(t)) (Cerebro.cpp:1142-1143): the rank of the time stamp among ALL camera frames DataManager holds (every frame
    // gets a node, DataManager.cpp image callbacks; only keyframes get descriptors), not the row of the descriptor DB.  Two ways
    // to give this mirror the same information, both optional:
    //   data_map_insert(stamp)        call it for every frame, as DataManager does (any order; duplicates ignored);
    //   set_frame_index_map(r2f)      r2f[row] = index of that row's frame in data_map (e.g. from a recorded run).
    // With neither, global_a / global_b fall back to the DB row (= wholeImageComputedList index).
    void data_map_insert(const Time &stamp);
    void set_frame_index_map(const std::vector<int64_t> &row_to_frame);
    int64_t data_map_size() const;

    // ---- thread-safe accessors (Cerebro.h:105-106)
    int wholeImageComputedList_size() const;
    Time wholeImageComputedList_at(int k) const;

    // ---- consumer side (dot_product_th)
    // One iteration of the while-loop body of descrip_N__dot__descrip_0_N (Cerebro.cpp:956-1100) for the current
    // l = wholeImageComputedList_size() (or an explicit l when replaying a recorded schedule).  Returns true iff a
    // loop candidate was pushed to foundLoops.
    bool descrip_N__dot__descrip_0_N_once(int64_t l = -1, chip_tick_result *detail = nullptr);
    // The thread main (Cerebro.cpp:350-357): polls at `rate_hz` until run_thread_disable().
    void run(double rate_hz = 10.0);
    void run_thread_enable() { b_run_thread = true; }
    void run_thread_disable() { b_run_thread = false; }

    // ---- top-k candidate policies on a flat inner-product index (upstream: faiss::IndexFlatIP, HAVE_FAISS builds only).
    // The "index" is the prefix [0, l-150) of the device DB; index.search(1, x, 5) is chip_query_rows(k, row, topk=5)
    // with the fp64 score rounded to float.  One call = one iteration of the respective while-loop body.
    //   faiss__naive_loopcandidate_generator   Cerebro.cpp:366-492: 3 consecutive top-1 labels within 12, last score > 0.9f
    //   faiss_clique_loopcandidate_generator   Cerebro.cpp:506-722: neighbours > 0.85 accumulated in `retained` (LOCALITY 7),
    //                                          flushed when l_i % 4 == 0, rand()-thinned when more than one key
    // Return the number of candidates pushed to foundLoops.
    int faiss__naive_loopcandidate_generator_once(int64_t l = -1);
    int faiss_clique_loopcandidate_generator_once(int64_t l = -1);
    std::function<int()> rand_source;  // rand() of Cerebro.cpp:692; defaults to std::rand

    // ---- foundLoops (Cerebro.h:152-158)
    int foundLoops_count() const;
    std::tuple<Time, Time, double> foundLoops_i(int i) const;
    // Same keys as Cerebro.cpp:1149-1159.  global_a / global_b: the reference's data_map index when the frames were registered
    // (data_map_insert / set_frame_index_map / a state.json cold start, which lists every node), else the DB row.
    std::string foundLoops_as_JSON() const;

    chip_dot_params params;  // LOCALITY_THRESH / DOT_PROD_THRESH / lag  (Cerebro.cpp:912-914)

private:
    chip_ctx *ctx_ = nullptr;
    int status_ = 0;
    int D_ = 0;
    std::string error_;
    mutable std::mutex m_wholeImageComputedList;
    std::vector<Time> wholeImageComputedList;
    mutable std::mutex m_foundLoops;
    std::vector<std::tuple<Time, Time, double>> foundLoops;
    mutable std::mutex m_data_map;
    mutable std::vector<uint64_t> data_map_;          // all frame stamps (sec << 32 | nsec); sorted lazily
    mutable bool data_map_sorted_ = true;
    std::vector<int64_t> row_to_frame_;
    std::atomic<bool> b_run_thread{false};
    // state of the two index policies (function-locals of the reference: Cerebro.cpp:393-394 / :535-539)
    bool index_search(int64_t ntotal, const int64_t *rows, int n_rows, float *distances, int64_t *labels);
    int64_t naive_last_l_ = 0, naive_l_last_added_to_index_ = 0;
    int64_t clique_last_l_ = 0, clique_l_last_added_to_index_ = 0;
    std::map<int64_t, int> retained_;
};

struct StaticTheiaPoseCompute {
    // float PNP(w_X, c_uv_normalized, c_T_w, pnp__msg)  (DlsPnpWithRansac.cpp:132-245).  c_T_w: column-major 4x4.
    // Returns summary.confidence, or -1 for fewer than 20 points (:136-139) / on a library error.
    static float PNP(chip_ctx *ctx, const std::vector<std::array<double, 3>> &w_X,
                     const std::vector<std::array<double, 2>> &c_uv_normalized, double c_T_w[16], std::string &pnp__msg,
                     const chip_ransac_params *params = nullptr, std::vector<uint8_t> *inliers = nullptr);
};

struct StaticTheiaPoseComputeICP {
    // float P3P_ICP(uv_X, uvd_Y, uvd_T_uv, p3p__msg)  (DlsPnpWithRansac.cpp:16-122, RANSAC branch).  Same conventions as PNP.
    static float P3P_ICP(chip_ctx *ctx, const std::vector<std::array<double, 3>> &uv_X, const std::vector<std::array<double, 3>> &uvd_Y,
                         double uvd_T_uv[16], std::string &p3p__msg, const chip_ransac_params *params = nullptr);
};

// ---------------------------------------------------------------------------------------------------------------
// Row N3: what happens right after the path -- the three-way consistency gate and the LoopEdge message.
// Mirrors ProcessedLoopCandidate (src/ProcessedLoopCandidate.{h,cpp}) and the pose part of
// Cerebro::process_loop_candidate_imagepair_consistent_pose_compute (src/Cerebro.cpp:1512-1719).
struct ProcessedLoopCandidate {
    Time t_node_1, t_node_2;                       // node_1->getT(), node_2->getT()
    int idx_from_datamanager_1 = -1, idx_from_datamanager_2 = -1;
    int pf_matches = 0;                            // number of GMS point-feature matches (Cerebro.cpp:1505)
    std::vector<std::array<double, 16>> opX_b_T_a; // op1__b_T_a, op2__b_T_a, icp_b_T_a (column-major)
    std::vector<float> opX_goodness;
    // results of the gate
    bool isSet_3d2d__2T1 = false;
    std::array<double, 16> _3d2d__2T1{};
    float _3d2d__2T1__ransac_confidence = 0.f;

    // ProcessedLoopCandidate.cpp:16-36
    bool makeLoopEdgeMsg(LoopEdgePOD &msg) const;
    // ProcessedLoopCandidate.cpp:40-125: 3 candidates; |dt| >= 10 s; pairwise |ypr|_inf < 5 deg and |t|_inf < 0.2 m
    // (the reference tests op1_m_icp_tr twice instead of op1_m_op2_tr -- kept); pf_matches > 800.
    bool makeLoopEdgeMsgWithConsistencyCheck(LoopEdgePOD &msg);
};

// 3-D/2-D and 3-D/3-D correspondence sets of an image pair, in the layout StaticPointFeatureMatching produces
// (src/utils/PointFeatureMatching.cpp:95-195): points of frame a with their normalized projections in b, and vice versa.
struct PosePairInput {
    std::vector<std::array<double, 3>> world_point_uv;       // a_X       (Cerebro.cpp:1512)
    std::vector<std::array<double, 2>> feature_position_uv_d; // uv in b
    std::vector<std::array<double, 3>> world_point_uv_d;     // b_X       (Cerebro.cpp:1566)
    std::vector<std::array<double, 2>> feature_position_uv;   // uv in a
    std::vector<std::array<double, 3>> uv_X, uvd_Y;          // 3d-3d     (Cerebro.cpp:1629)
};
// PNP(a->b) (:1518), PNP(b->a) inverted (:1572,:1582), P3P_ICP (:1629), NaN gate (:1678), push of the 3 poses (:1706-1719).
// Returns false when one of the three poses has NaN/Inf (the candidate is skipped).
bool compute_three_way_pose(chip_ctx *ctx, const PosePairInput &in, ProcessedLoopCandidate &proc_candi, uint64_t seed = 0);

// PoseManipUtils::R2ypr (src/utils/PoseManipUtils.cpp:148-163), degrees, from a column-major 4x4
void matrix4_to_rawyprt(const double T_colmajor[16], double ypr_deg[3], double t[3]);
void matrix4_inverse_rigid(const double T[16], double Tinv[16]);
void matrix4_mul(const double A[16], const double B[16], double C[16]);

// geometry_msgs::Pose from a column-major 4x4 (PoseManipUtils::eigenmat_to_geometry_msgs_Pose,
// src/utils/PoseManipUtils.cpp:31-45: position = T(0..2,3), orientation = Quaterniond(T.topLeftCorner<3,3>()))
void matrix4_to_pose(const double T_colmajor[16], double position[3], double orientation_xyzw[4]);

// ProcessedLoopCandidate::makeLoopEdgeMsg (src/ProcessedLoopCandidate.cpp:16-36)
LoopEdgePOD make_loop_edge(const Time &t_node_1, const Time &t_node_2, const double _3d2d__2T1[16], float ransac_confidence,
                           int idx_from_datamanager_1, int idx_from_datamanager_2);

}  // namespace cerebro_hip
