// MOCK (tests/abi_c): the data members of Param (reference: swarm_planner/include/param.hpp:9-38); no ROS parameter server
#pragma once
#include <sp_const.hpp>
#include <string>
namespace SwarmPlanning {
class Param {
public:
    bool log = false;
    std::string package_path;
    double world_x_min = -5, world_y_min = -5, world_z_min = 0, world_x_max = 5, world_y_max = 5, world_z_max = 2.5;
    double ecbs_w = 1.3, grid_xy_res = 0.3, grid_z_res = 0.6, grid_margin = 0.2;
    double box_xy_res = 0.1, box_z_res = 0.1;
    bool time_scale = true;
    double time_step = 1, downwash = 2.0;
    int iteration = 1;
    bool sequential = false;
    int batch_size = 4, batch_iter = 0, n = 5, phi = 3;
    std::vector<std::vector<double>> color;
};
}  // namespace SwarmPlanning
