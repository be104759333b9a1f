// MOCK (tests/abi_c): the shape of PlanResult and its typedefs (reference: swarm_planner/include/sp_const.hpp:16-28), so that the adapter
// of INTEGRATION.md is seen by a compiler without ROS / octomap.  Test infrastructure.
#pragma once
#include <octomap/OcTree.h>
#include <std_msgs/Float64MultiArray.h>
#include <std_msgs/MultiArrayDimension.h>
typedef std::vector<std::vector<octomap::point3d>> initTraj_t;
typedef std::vector<std::vector<std::pair<std::vector<double>, double>>> SFC_t;
typedef std::vector<std::vector<std::vector<std::pair<octomap::point3d, double>>>> RSFC_t;
namespace SwarmPlanning {
struct PlanResult {
    initTraj_t initTraj;
    std::vector<double> T;
    SFC_t SFC;
    RSFC_t RSFC;
    std_msgs::Float64MultiArray msgs_traj_info;
    std::vector<std_msgs::Float64MultiArray> msgs_traj_coef;
};
}  // namespace SwarmPlanning
