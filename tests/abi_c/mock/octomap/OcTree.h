// MOCK (tests/abi_c, test infrastructure): the three things of octomap the adapter of INTEGRATION.md touches -- octomap::point3d =
// octomath::Vector3, three floats with x() / y() / z().  Written for the compile check; not octomap.
#pragma once
#include <cmath>
#include <memory>
#include <utility>
#include <vector>
namespace octomath {
class Vector3 {
public:
    Vector3() : d{0, 0, 0} {}
    Vector3(float x, float y, float z) : d{x, y, z} {}
    float& x() { return d[0]; }
    float& y() { return d[1]; }
    float& z() { return d[2]; }
    const float& x() const { return d[0]; }
    const float& y() const { return d[1]; }
    const float& z() const { return d[2]; }
private:
    float d[3];
};
}  // namespace octomath
namespace octomap {
typedef octomath::Vector3 point3d;
}
