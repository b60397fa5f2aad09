#include "cerebro_hip_ros/adapter.h"

namespace cerebro_hip_ros {

LoopEdge to_msg(const cerebro_hip::LoopEdgePOD &e)
{
    LoopEdge m;                                   // ProcessedLoopCandidate.cpp:24-33
    m.timestamp0 = to_ros(e.timestamp0);
    m.timestamp1 = to_ros(e.timestamp1);
    m.pose_1T0.position.x = e.position[0];
    m.pose_1T0.position.y = e.position[1];
    m.pose_1T0.position.z = e.position[2];
    m.pose_1T0.orientation.x = e.orientation_xyzw[0];
    m.pose_1T0.orientation.y = e.orientation_xyzw[1];
    m.pose_1T0.orientation.z = e.orientation_xyzw[2];
    m.pose_1T0.orientation.w = e.orientation_xyzw[3];
    m.weight = e.weight;
    m.description = e.description;
    return m;
}

bool descriptor_from_response(cerebro_hip::Cerebro &cer, const ros::Time &stamp, const WholeImageDescriptorCompute::Response &res)
{
    if (res.desc.empty()) return false;           // Cerebro.cpp:262: the reference asserts desc.size() > 0
    cer.data_map_insert(from_ros(stamp));
    return cer.descriptor_available(from_ros(stamp), res.desc.data(), (int)res.desc.size());
}

}  // namespace cerebro_hip_ros
