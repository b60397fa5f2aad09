// tests/ros_stub: stand-in for <geometry_msgs/Pose.h> (geometry_msgs/Pose = Point position + Quaternion orientation, float64 members).
#pragma once
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
}  // namespace geometry_msgs
