// tests/ros_stub: stand-in for <ros/ros.h> -- ONLY what ros_adapter/ uses (ros::Time with sec / nsec).  Test infrastructure, not ROS.
#pragma once
#include <cstdint>
namespace ros {
struct Time {
    uint32_t sec = 0, nsec = 0;
    Time() = default;
    Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
};
}  // namespace ros
