// MOCK (tests/abi_c): a DynamicEDTOctomap that serves getDistance() from a flat grid, with the lookup rule SURVEY.md App. B states
// (voxel key = floor(coord / res), -1 outside the box).  Only what the adapter calls.
#pragma once
#include <octomap/OcTree.h>
class DynamicEDTOctomap {
public:
    DynamicEDTOctomap(const int dim_[3], const int key_min_[3], double res_, std::vector<float> grid_)
        : res(res_), grid(std::move(grid_)) {
        for (int a = 0; a < 3; ++a) dim[a] = dim_[a], key_min[a] = key_min_[a];
    }
    float getDistance(const octomap::point3d& p) const {
        const float c[3] = {p.x(), p.y(), p.z()};
        int k[3];
        for (int a = 0; a < 3; ++a) {
            k[a] = (int)std::floor((1.0 / res) * (double)c[a]) - key_min[a];
            if (k[a] < 0 || k[a] >= dim[a]) return -1.0f;
        }
        return grid[((size_t)k[0] * dim[1] + k[1]) * dim[2] + k[2]];
    }
private:
    int dim[3], key_min[3];
    double res;
    std::vector<float> grid;
};
