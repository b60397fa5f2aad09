// adapter.h -- the ROS edge of the drop-in: converts between the ROS types of the reference's surface and the
// ROS-free host mirror (cerebro_amd/host/cerebro_host.h).  Built only where catkin exists (ros_adapter/CMakeLists.txt); NOT
// compiled in the repository's own build image (no ROS there) -- kept to a dozen statements for that reason.
//   reference call sites:  Cerebro.cpp:94-105,260-275 (service call + setWholeImageDescriptor + wholeImageComputedList_pushback)
//                          ProcessedLoopCandidate.cpp:16-36 (makeLoopEdgeMsg), cerebro_node.cpp (pub_loopedge.publish)
#pragma once
#include <ros/ros.h>
#include <geometry_msgs/Pose.h>

#include "cerebro_host.h"

#if !defined(CEREBRO_HIP_MSG_NS)
#define CEREBRO_HIP_MSG_NS cerebro
#endif
#define CEREBRO_HIP_STR2(x) #x
#define CEREBRO_HIP_STR(x) CEREBRO_HIP_STR2(x)
#include CEREBRO_HIP_STR(CEREBRO_HIP_MSG_NS/LoopEdge.h)
#include CEREBRO_HIP_STR(CEREBRO_HIP_MSG_NS/WholeImageDescriptorCompute.h)

namespace cerebro_hip_ros {

using LoopEdge = CEREBRO_HIP_MSG_NS::LoopEdge;
using WholeImageDescriptorCompute = CEREBRO_HIP_MSG_NS::WholeImageDescriptorCompute;

inline cerebro_hip::Time from_ros(const ros::Time &t) { cerebro_hip::Time r; r.sec = t.sec; r.nsec = t.nsec; return r; }
inline ros::Time to_ros(const cerebro_hip::Time &t) { return ros::Time(t.sec, t.nsec); }

// LoopEdgePOD (make_loop_edge / makeLoopEdgeMsgWithConsistencyCheck) -> the message the reference publishes
LoopEdge to_msg(const cerebro_hip::LoopEdgePOD &e);

// The descriptor thread's step (Cerebro.cpp:260-275): hand the service response's float64[] to the device DB.
// Also registers the frame in the data_map mirror, so that foundLoops_as_JSON reports the reference's global_a / global_b.
bool descriptor_from_response(cerebro_hip::Cerebro &cer, const ros::Time &stamp, const WholeImageDescriptorCompute::Response &res);

// Every camera frame (DataManager's image callback inserts a node per frame): keeps global_a / global_b reference-compatible.
inline void frame_seen(cerebro_hip::Cerebro &cer, const ros::Time &stamp) { cer.data_map_insert(from_ros(stamp)); }

}  // namespace cerebro_hip_ros
