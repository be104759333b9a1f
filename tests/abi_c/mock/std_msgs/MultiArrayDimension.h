// MOCK (tests/abi_c): std_msgs/MultiArrayDimension as far as generateROSMsg (rbp_planner.hpp:269-291) uses it
#pragma once
#include <cstdint>
#include <string>
namespace std_msgs {
struct MultiArrayDimension {
    std::string label;
    uint32_t size = 0, stride = 0;
};
}  // namespace std_msgs
