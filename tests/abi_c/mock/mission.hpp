// MOCK (tests/abi_c): the data members of Mission (reference: swarm_planner/include/mission.hpp:13-15); no JSON, no ROS
#pragma once
#include <ros/ros.h>
#include <vector>
namespace SwarmPlanning {
class Mission {
public:
    int qn = 0;
    std::vector<std::vector<double>> startState, goalState, max_vel, max_acc;
    std::vector<double> quad_size, quad_speed;
};
}  // namespace SwarmPlanning
