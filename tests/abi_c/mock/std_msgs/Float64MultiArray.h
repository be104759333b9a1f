// MOCK (tests/abi_c): std_msgs/Float64MultiArray as far as generateROSMsg (rbp_planner.hpp:269-291) uses it
#pragma once
#include <std_msgs/MultiArrayDimension.h>
#include <vector>
namespace std_msgs {
struct MultiArrayLayout {
    std::vector<MultiArrayDimension> dim;
    uint32_t data_offset = 0;
};
struct Float64MultiArray {
    MultiArrayLayout layout;
    std::vector<double> data;
};
}  // namespace std_msgs
