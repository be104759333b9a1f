// MOCK (tests/abi_c): the two logging macros the adapter uses
#pragma once
#include <iostream>
#define ROS_ERROR_STREAM(x) (std::cerr << "[ERROR] " << x << std::endl)
#define ROS_WARN_STREAM(x) (std::cerr << "[WARN] " << x << std::endl)
