// edt.hip — the distance grid of a world on gfx950 ("GPU EDT, later").
//
// Replaces, for the Corridor stage's input,
//     DynamicEDTOctomap distmap(maxDist, tree, world_min, world_max, false); distmap.update();
// (reference: swarm_planner/src/swarm_traj_planner_rbp_test_all.cpp:57-63, src/swarm_traj_planner_rbp.cpp:73-80) followed by
// getDistance() on every voxel centre of the bounding box, i.e. the float grid rbp_world describes.  The host library has the
// same function on the CPU (csrc/host/octomap_edt.cpp, rbp_world_build); the two are bit-identical (tests/test_gpu_edt.py).
//
// dynamicEDT3D clamps: maxDist_squared = ((int)(maxDist / res + 1))^2 cells, every voxel at or beyond keeps sqrt(maxDist_squared).
// So only obstacles closer than md = (int)(maxDist / res + 1) cells along EVERY axis can matter, and the exact squared Euclidean
// distance transform separates into three min-plus passes with a window of 2 md - 1 cells:
//     d2(x,y,z) = min_x' (x-x')^2 + [ min_y' (y-y')^2 + [ min_z' (z-z')^2 + occ(x',y',z') ] ]      (occ = 0 on obstacles, "infinity" elsewhere)
// in int32 (values <= 3 md^2): exact, no lower-envelope bookkeeping, one thread per voxel and pass, z fastest so the x and y passes
// read coalesced.  Wherever the true d2 is < md^2 the windowed minimum equals it (each component of the minimiser is < md); elsewhere
// both are >= md^2 and clamp to the same value.  11-cell windows on the 101 x 101 x 23 grids of the benchmark: 0.23 M voxels x 63 reads.
#include "rbp_dev.h"

#include <string>
#include <vector>

namespace {

constexpr int EDT_INF = 0x3f3f3f3f;

__global__ __launch_bounds__(256) void edt_raster_kernel(const int* __restrict__ keys, long long n_leaves, int kx0, int ky0, int kz0, int nx, int ny,
                                                         int nz, int* __restrict__ g) {
    // one thread per occupied leaf: a cube of `s` voxels per edge, clipped to the box (DynamicEDTOctomap::initializeOcTree)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    const int s = keys[4 * i + 3];
    const int x0 = max(keys[4 * i] - kx0, 0), x1 = min(keys[4 * i] - kx0 + s - 1, nx - 1);
    const int y0 = max(keys[4 * i + 1] - ky0, 0), y1 = min(keys[4 * i + 1] - ky0 + s - 1, ny - 1);
    const int z0 = max(keys[4 * i + 2] - kz0, 0), z1 = min(keys[4 * i + 2] - kz0 + s - 1, nz - 1);
    for (int x = x0; x <= x1; ++x)
        for (int y = y0; y <= y1; ++y)
            for (int z = z0; z <= z1; ++z) g[((size_t)x * ny + y) * nz + z] = 0;
}

// out(c) = min over |d| < md of d^2 + in(c + d * stride) along one axis (n cells, position p of c along it)
__global__ __launch_bounds__(256) void edt_pass_kernel(const int* __restrict__ in, int* __restrict__ out, long long ncell, int n, long long stride, int md) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const int p = (int)((c / stride) % n);
    const int lo = max(-(md - 1), -p), hi = min(md - 1, n - 1 - p);
    int best = EDT_INF;
    for (int d = lo; d <= hi; ++d) {
        const int v = in[c + d * stride];
        best = min(best, v >= EDT_INF ? EDT_INF : v + d * d);
    }
    out[c] = best;
}

__global__ __launch_bounds__(256) void edt_finish_kernel(const int* __restrict__ d2, float* __restrict__ dist, long long ncell, int md2, double res) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const int sq = d2[c];
    const float cells = (float)__dsqrt_rn((double)(sq < md2 ? sq : md2));  // dynamicEDT3D keeps float distances in cells
    dist[c] = (float)((double)cells * res);                                 // getDistance: float * double treeResolution -> float
}

int dims(double res, const double* bmin, const double* bmax, int* dim, int* kmin) {
    if (!(res > 0)) return 1;
    const double rf = 1.0 / res;  // octomap resolution_factor
    for (int a = 0; a < 3; ++a) {
        // octomap::point3d is float32; coordToKey(c) = (int)floor(resolution_factor * c) (+32768)
        kmin[a] = (int)floor(rf * (double)(float)bmin[a]);
        const int kmax = (int)floor(rf * (double)(float)bmax[a]);
        if (kmax < kmin[a]) return 1;
        dim[a] = kmax - kmin[a] + 1;
        if (dim[a] > 4096) return 1;
    }
    return 0;
}

}  // namespace

extern "C" int rbp_edt_dims(double res, const double bbx_min[3], const double bbx_max[3], int32_t dim[3], int32_t key_min[3]) {
    if (!bbx_min || !bbx_max || !dim || !key_min || dims(res, bbx_min, bbx_max, dim, key_min))
        return rbp_set_error(RBP_ERR_BAD_ARGUMENT, "rbp_edt_dims: need res > 0, bbx_min <= bbx_max, at most 4096 voxels per axis");
    return RBP_OK;
}

extern "C" int rbp_edt_build(const int32_t* leaf_keys, int64_t n_leaves, double res, const double bbx_min[3], const double bbx_max[3],
                             double max_dist, float* dist) {
    int dim[3], kmin[3];
    if (!bbx_min || !bbx_max || !dist || n_leaves < 0 || (n_leaves > 0 && !leaf_keys) || !(max_dist > 0) || dims(res, bbx_min, bbx_max, dim, kmin))
        return rbp_set_error(RBP_ERR_BAD_ARGUMENT, "rbp_edt_build: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return rbp_set_error(RBP_ERR_NO_DEVICE, "no HIP device: the RBP path has no CPU fallback");
    const long long ncell = (long long)dim[0] * dim[1] * dim[2];
    const int md = (int)(max_dist / res + 1);
    if (md < 1 || md > 4096) return rbp_set_error(RBP_ERR_BAD_ARGUMENT, "rbp_edt_build: max_dist / res out of range");
    int *d_keys = nullptr, *d_a = nullptr, *d_b = nullptr;
    float* d_dist = nullptr;
    hipError_t e = hipSuccess;
    auto done = [&](int rc, const std::string& msg) {
        (void)hipFree(d_keys), (void)hipFree(d_a), (void)hipFree(d_b), (void)hipFree(d_dist);
        return rc == RBP_OK ? RBP_OK : rbp_set_error(rc, msg.c_str());
    };
#define EDT_TRY(expr)                                                                                  \
    do {                                                                                               \
        e = (expr);                                                                                    \
        if (e != hipSuccess) return done(RBP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e)); \
    } while (0)
    EDT_TRY(hipMalloc((void**)&d_a, sizeof(int) * ncell));
    EDT_TRY(hipMalloc((void**)&d_b, sizeof(int) * ncell));
    EDT_TRY(hipMalloc((void**)&d_dist, sizeof(float) * ncell));
    EDT_TRY(hipMemset(d_a, 0x3f, sizeof(int) * ncell));
    const unsigned nb = (unsigned)((ncell + 255) / 256);
    if (n_leaves > 0) {
        EDT_TRY(hipMalloc((void**)&d_keys, sizeof(int) * 4 * n_leaves));
        EDT_TRY(hipMemcpy(d_keys, leaf_keys, sizeof(int) * 4 * n_leaves, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(edt_raster_kernel, dim3((unsigned)((n_leaves + 255) / 256)), dim3(256), 0, 0, d_keys, (long long)n_leaves, kmin[0], kmin[1],
                           kmin[2], dim[0], dim[1], dim[2], d_a);
    }
    hipLaunchKernelGGL(edt_pass_kernel, dim3(nb), dim3(256), 0, 0, d_a, d_b, ncell, dim[2], 1LL, md);                          // z
    hipLaunchKernelGGL(edt_pass_kernel, dim3(nb), dim3(256), 0, 0, d_b, d_a, ncell, dim[1], (long long)dim[2], md);             // y
    hipLaunchKernelGGL(edt_pass_kernel, dim3(nb), dim3(256), 0, 0, d_a, d_b, ncell, dim[0], (long long)dim[1] * dim[2], md);    // x
    hipLaunchKernelGGL(edt_finish_kernel, dim3(nb), dim3(256), 0, 0, d_b, d_dist, ncell, md * md, res);
    EDT_TRY(hipGetLastError());
    EDT_TRY(hipMemcpy(dist, d_dist, sizeof(float) * ncell, hipMemcpyDeviceToHost));
#undef EDT_TRY
    return done(RBP_OK, "");
}
