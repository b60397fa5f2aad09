// tests/ros_stub: stand-in for <sensor_msgs/Image.h> -- the request side of WholeImageDescriptorCompute.srv carries one; the adapter never looks inside.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
namespace sensor_msgs {
struct Image { uint32_t height = 0, width = 0; std::string encoding; uint8_t is_bigendian = 0; uint32_t step = 0; std::vector<uint8_t> data; };
}  // namespace sensor_msgs
