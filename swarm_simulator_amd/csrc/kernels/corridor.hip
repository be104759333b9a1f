// corridor.hip — SFC box growth and RSFC normals on gfx950.
//
// Replaces Corridor::update (reference: swarm_planner/include/rbp_corridor.hpp:21-26):
//   sfc_kernel   updateObsBox :149-243 with isObstacleInBox :44-78, isBoxInBoundary :80-87,
//                isPointInBox :89-97, expand_box :99-147
//   rsfc_kernel  updateRelBox :338-398
//
// SFC: ONE WAVEFRONT PER (mission, agent).  The control flow of expand_box is inherently sequential
// (round-robin axis growth, each step decided by the previous test) and is executed wave-uniformly; the
// work of each isObstacleInBox test — tens to 234 k getDistance samples — is spread over the 64 lanes in
// the reference's own sample order (x outer, z inner), 64 samples per step, with a ballot to find the first
// obstacle, so the early exit and the sample count are exactly the reference's.  Sample coordinates are
// produced by the same double accumulation / float32 rounding / floor as the CPU path (bit-exact boxes);
// per-axis voxel indices are cached in LDS so the inner loop is three ds_reads + one occupancy read.  Four agents of a
// mission share a workgroup and one LDS BITMASK of the grid (bit = dist < r - 1e-6, 29 KB for 101x101x23), built once per
// workgroup with coalesced reads + ballots; agents whose radius differs from the group's first read the float grid.
//
// RSFC: one thread per (mission, pair, segment), float32 arithmetic in octomath's operation order
// (compiled with -ffp-contract=off; HIP's float division and the f64 sqrt are correctly rounded).
#include "rbp_dev.h"

#define SFC_MAXS 512        // max samples per axis (world extent / box resolution + 2)
#define SFC_WAVES 4         // agents (wavefronts) per workgroup, sharing one occupancy bitmask in LDS
#define SFC_MASK_WORDS 8192  // 262144 cells = 32 KB; larger grids fall back to reading the float grid

namespace {

struct AxisCache {
    double lo, hi;  // extent the cached keys belong to
    int n;          // number of samples
};

// Computes the voxel indices of the samples along one axis of `box` (rbp_corridor.hpp:47-63 for that axis):
//   v = lo; c = 0; while (v < hi + 1e-6) { coord = (c == 0 && lo > world_min + 1e-6) ? lo - 1e-6 : v + 1e-6; ... v += res }
// and DynamicEDTOctomap::getDistance's key computation floor((1/res_map) * (double)(float)coord) - key_min.
// All lanes run the same scalar loop and store identical values.
__device__ __forceinline__ int axis_keys(int* keys, double lo, double hi, double step, double world_lo, double rf,
                                         int key_min, int dim) {
    int c = 0;
    for (double v = lo; v < hi + SP_EPSILON_FLOAT && c < SFC_MAXS; v += step, ++c) {
        double coord = v + SP_EPSILON_FLOAT;
        if (c == 0 && lo > world_lo + SP_EPSILON_FLOAT) coord = lo - SP_EPSILON_FLOAT;
        float cf = (float)coord;  // octomap::point3d is float32
        int k = (int)floor(rf * (double)cf) - key_min;
        keys[c] = (k >= 0 && k < dim) ? k : -1;
    }
    return c;
}

struct SfcCtx {
    const unsigned* mask;  // LDS occupancy bitmask (bit = dist < margin - 1e-6) or nullptr
    const float* grid;
    int dim[3], key_min[3];
    double rf, world_min[3], world_max[3], res[3];
    double margin_cmp;  // margin - 1e-6
    int* keys[3];       // LDS
    AxisCache cache[3];
    unsigned long long samples;
};

// rbp_corridor.hpp:44-78.  Returns true if any sample reads dist < margin - 1e-6 (or lies outside the grid: -1).
__device__ bool is_obstacle_in_box(SfcCtx& c, const double* box, int lane) {
    int n[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!(c.cache[a].lo == box[a] && c.cache[a].hi == box[a + 3])) {
            c.cache[a].n = axis_keys(c.keys[a], box[a], box[a + 3], c.res[a], c.world_min[a], c.rf, c.key_min[a], c.dim[a]);
            c.cache[a].lo = box[a];
            c.cache[a].hi = box[a + 3];
        }
        n[a] = c.cache[a].n;
    }
    __builtin_amdgcn_wave_barrier();
    const long long total = (long long)n[0] * n[1] * n[2];
    if (total == 0) return false;
    // mixed-radix decomposition of the lane id and of the stride 64 in (n0, n1, n2), z fastest
    int c2 = lane % n[2], t = lane / n[2];
    int c1 = t % n[1], c0 = t / n[1];
    const int d2 = 64 % n[2], t64 = 64 / n[2];
    const int d1 = t64 % n[1], d0 = t64 / n[1];
    const int ny = c.dim[1], nz = c.dim[2];
    for (long long base = 0; base < total; base += 64) {
        bool hit = false;
        if (base + lane < total) {
            int ix = c.keys[0][c0], iy = c.keys[1][c1], iz = c.keys[2][c2];
            if ((ix | iy | iz) < 0) {
                hit = true;  // getDistance returns -1 outside the map
            } else {
                const unsigned cell = ((unsigned)ix * ny + iy) * nz + iz;
                if (c.mask) {
                    hit = (c.mask[cell >> 5] >> (cell & 31)) & 1u;
                } else {
                    float d = c.grid[cell];
                    hit = (double)d < c.margin_cmp;
                }
            }
        }
        unsigned long long m = __ballot(hit);
        if (m) {
            c.samples += (unsigned long long)(__ffsll((long long)m));  // the reference stops at the first hit
            return true;
        }
        c2 += d2;
        if (c2 >= n[2]) c2 -= n[2], c1++;
        c1 += d1;
        if (c1 >= n[1]) c1 -= n[1], c0++;
        c0 += d0;
        long long rem = total - base;
        c.samples += (unsigned long long)(rem < 64 ? rem : 64);
    }
    return false;
}

__device__ __forceinline__ bool is_box_in_boundary(const SfcCtx& c, const double* b) {  // :80-87
    return b[0] > c.world_min[0] - SP_EPSILON && b[1] > c.world_min[1] - SP_EPSILON && b[2] > c.world_min[2] - SP_EPSILON &&
           b[3] < c.world_max[0] + SP_EPSILON && b[4] < c.world_max[1] + SP_EPSILON && b[5] < c.world_max[2] + SP_EPSILON;
}
__device__ __forceinline__ bool is_point_in_box(const float* p, const double* b) {  // :89-97
    return p[0] > b[0] - SP_EPSILON && p[1] > b[1] - SP_EPSILON && p[2] > b[2] - SP_EPSILON && p[0] < b[3] + SP_EPSILON &&
           p[1] < b[4] + SP_EPSILON && p[2] < b[5] + SP_EPSILON;
}

// rbp_corridor.hpp:99-147
__device__ void expand_box(SfcCtx& c, double* box, int lane) {
    double cand[6], upd[6];
    int axis_cand[6] = {0, 1, 2, 3, 4, 5};
    int n_cand = 6, i = -1;
    while (n_cand > 0) {
#pragma unroll
        for (int e = 0; e < 6; ++e) cand[e] = box[e], upd[e] = box[e];
        while (!is_obstacle_in_box(c, upd, lane) && is_box_in_boundary(c, upd)) {
            i++;
            if (i >= n_cand) i = 0;
            int axis = 0;
#pragma unroll
            for (int e = 0; e < 6; ++e)
                if (e == i) axis = axis_cand[e];
#pragma unroll
            for (int e = 0; e < 6; ++e) box[e] = cand[e], upd[e] = cand[e];
            // grow `cand` by one step along `axis`; `upd` = the new slab only
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                if (axis == e) {
                    upd[e + 3] = cand[e];
                    cand[e] = cand[e] - c.res[e];
                    upd[e] = cand[e];
                }
                if (axis == e + 3) {
                    upd[e] = cand[e + 3];
                    cand[e + 3] = cand[e + 3] + c.res[e];
                    upd[e + 3] = cand[e + 3];
                }
            }
        }
        if (i < 0) i = 0;  // (reference: UB if the very first test fails; unreachable, the seed was tested)
#pragma unroll
        for (int e = 0; e < 5; ++e)
            if (e >= i) axis_cand[e] = axis_cand[e + 1];
        n_cand--;
        if (i > 0)
            i--;
        else
            i = n_cand - 1;
    }
}

__global__ __launch_bounds__(64 * SFC_WAVES) void sfc_kernel(DevSession s) {
    const int groups = (s.agent_end - s.agent_begin + SFC_WAVES - 1) / SFC_WAVES;
    const int mission = blockIdx.x / groups, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = s.agent_begin + (blockIdx.x % groups) * SFC_WAVES + wave;
    const int M = s.M, P = M + 1, MB = s.max_boxes;
    __shared__ int keys_all[SFC_WAVES][3][SFC_MAXS];
    __shared__ unsigned mask[SFC_MASK_WORDS];
    extern __shared__ int box_log_all[];  // [SFC_WAVES][MB][P]
    int* box_log = box_log_all + (size_t)wave * MB * P;
    const DevWorld w = s.worlds[mission];
    // ---- occupancy bitmask of this mission's grid for the radius of the group's first agent, built once per workgroup
    // with coalesced reads + ballots.  The SFC test only needs "dist < r - 1e-6" (rbp_corridor.hpp:67), so one bit per
    // cell (29 KB for the 101x101x23 grid) replaces ~0.6 M float reads per agent from L2/MALL by LDS reads.
    const int q0 = s.agent_begin + (blockIdx.x % groups) * SFC_WAVES;
    const double radius0 = s.radius[(size_t)mission * s.N + q0];
    const unsigned ncell = (unsigned)w.dim[0] * w.dim[1] * w.dim[2];
    const bool mask_fits = ncell + 64 <= 32u * SFC_MASK_WORDS;
    if (mask_fits) {
        const double cmp0 = radius0 - SP_EPSILON_FLOAT;
        for (unsigned base = wave * 64; base < ((ncell + 63) & ~63u); base += 64 * SFC_WAVES) {
            const unsigned cell = base + lane;
            const bool occ = cell < ncell && (double)w.dist[cell] < cmp0;
            const unsigned long long b = __ballot(occ);
            if (lane == 0) {
                mask[base >> 5] = (unsigned)b;
                mask[(base >> 5) + 1] = (unsigned)(b >> 32);
            }
        }
    }
    __syncthreads();
    if (qi >= s.agent_end) return;
    SfcCtx c;
    c.grid = w.dist;
    c.rf = 1.0 / w.res;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.dim[a] = w.dim[a], c.key_min[a] = w.key_min[a];
        c.world_min[a] = s.p.world_min[a], c.world_max[a] = s.p.world_max[a];
        c.keys[a] = keys_all[wave][a];
        c.cache[a].lo = 1e300, c.cache[a].hi = -1e300, c.cache[a].n = 0;
    }
    c.res[0] = c.res[1] = s.p.box_xy_res, c.res[2] = s.p.box_z_res;
    const double radius = s.radius[(size_t)mission * s.N + qi];
    c.margin_cmp = radius - SP_EPSILON_FLOAT;
    c.mask = (mask_fits && radius == radius0) ? mask : nullptr;  // agents with another radius read the float grid
    c.samples = 0;

    const float* traj = s.init_traj + ((size_t)mission * s.N + qi) * P * 3;
    const double* T = s.T + (size_t)mission * P;
    double* boxes = s.sfc_box + ((size_t)mission * s.N + qi) * MB * 6;
    double* times = s.sfc_time + ((size_t)mission * s.N + qi) * MB;
    int nbox = 0, err = 0;
    double prev[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < P - 1 && !err; ++i) {  // :157-193
        float pt[3] = {traj[3 * i], traj[3 * i + 1], traj[3 * i + 2]};
        float pn[3] = {traj[3 * i + 3], traj[3 * i + 4], traj[3 * i + 5]};
        if (is_point_in_box(pn, prev)) continue;
        double box[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {  // :174-179
            double lo = fmin((double)pt[a], (double)pn[a]), hi = fmax((double)pt[a], (double)pn[a]);
            box[a] = round(lo / c.res[a]) * c.res[a];
            box[a + 3] = round(hi / c.res[a]) * c.res[a];
        }
        if (is_obstacle_in_box(c, box, lane)) {
            err = RBP_ERR_OBSTACLE_IN_INIT_TRAJ;
            break;
        }
        expand_box(c, box, lane);
        if (nbox >= MB) {
            err = RBP_ERR_SFC_OVERFLOW;
            break;
        }
        if (lane < 6) boxes[6 * nbox + lane] = box[lane];
#pragma unroll
        for (int e = 0; e < 6; ++e) prev[e] = box[e];
        nbox++;
    }
    if (err) {
        if (lane == 0) atomicCAS(&s.status[mission], 0, err);
        return;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    // box_log(i, j): running count of consecutive waypoints j inside box i  (:198-210); lanes over boxes x waypoints
    for (int b = 0; b < nbox; ++b) {
        double bx[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) bx[e] = boxes[6 * b + e];
        for (int j = lane; j < P; j += 64) {
            float pj[3] = {traj[3 * j], traj[3 * j + 1], traj[3 * j + 2]};
            box_log[b * P + j] = is_point_in_box(pj, bx) ? 1 : 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // box_log is private to this wave
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        for (int b = 0; b < nbox; ++b)
            for (int j = 1; j < P; ++j)
                if (box_log[b * P + j]) box_log[b * P + j] = box_log[b * P + j - 1] + 1;
        // the time walk :212-237 (sequential)
        for (int b = 0; b < nbox; ++b) times[b] = -1;
        int box_iter = 0;
        const int box_max = nbox, path_max = P;
        for (int path_iter = 0; path_iter < path_max; path_iter++) {
            if (box_iter == box_max - 1) {
                if (box_log[box_iter * P + path_iter] > 0)
                    continue;
                else
                    box_iter--;
            }
            if (box_iter < 0 || path_iter < 0) break;  // undefined behaviour in the reference; guarded like the oracle
            if (box_log[box_iter * P + path_iter] > 0 && box_log[(box_iter + 1) * P + path_iter] > 0) {
                int count = 1;
                while (path_iter + count < path_max && box_log[box_iter * P + path_iter + count] > 0 &&
                       box_log[(box_iter + 1) * P + path_iter + count] > 0)
                    count++;
                times[box_iter] = T[path_iter + count / 2];
                path_iter = path_iter + count / 2;
                box_iter++;
            } else if (box_log[box_iter * P + path_iter] == 0) {
                box_iter--;
                path_iter--;
            }
        }
        if (box_max > 0) times[box_max - 1] = T[M];  // makespan :237
        s.sfc_count[(size_t)mission * s.N + qi] = nbox;
        atomicAdd(&s.counters[(size_t)mission * CT_N + CT_SFC_SAMPLES], c.samples);
    }
}

// ---- RSFC -----------------------------------------------------------------------------------------------
// octomath::Vector3 semantics: float32 components; dot()/norm_sq() evaluate the float expression left to right
// and widen; norm() = sqrt(double); normalize() divides by (float)norm when norm > 0.
__device__ __forceinline__ double v_dot(const float* a, const float* b) { return (double)(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]); }
__device__ __forceinline__ double v_norm(const float* a) { return __dsqrt_rn((double)(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])); }
__device__ __forceinline__ void v_normalize(float* a) {
    double len = v_norm(a);
    if (len > 0) {
        float l = (float)len;
        a[0] = a[0] / l, a[1] = a[1] / l, a[2] = a[2] / l;
    }
}

__global__ __launch_bounds__(256) void rsfc_kernel(DevSession s) {
    const int M = s.M, P = M + 1, N = s.N;
    const long long per_mission = (long long)s.npair * M;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= per_mission * s.K) return;
    const int mission = (int)(gid / per_mission);
    const long long r = gid % per_mission;
    const int pair = (int)(r / M), seg = (int)(r % M);
    // invert pair index -> (qi, qj), qi < qj, qi-major (rbp_corridor.hpp:342-344)
    int qi = 0, rem = pair;
    while (rem >= N - 1 - qi) rem -= N - 1 - qi, qi++;
    const int qj = qi + 1 + rem;
    if (pair == 0) s.rsfc_time[(size_t)mission * M + seg] = s.T[(size_t)mission * P + seg + 1];  // :390 (every shard)
    if (qi < s.agent_begin || qi >= s.agent_end) return;  // pair rows of another shard
    const float* ti = s.init_traj + ((size_t)mission * N + qi) * P * 3 + 3 * seg;
    const float* tj = s.init_traj + ((size_t)mission * N + qj) * P * 3 + 3 * seg;
    const double dw = s.p.downwash;
    float a[3], b[3], c[3], n[3], m[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a[k] = tj[k] - ti[k];          // :354
        b[k] = tj[3 + k] - ti[3 + k];  // :355
    }
    a[2] = (float)((double)a[2] / dw);  // :358-359
    b[2] = (float)((double)b[2] / dw);
    if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) {
        m[0] = a[0], m[1] = a[1], m[2] = a[2];
    } else {
        m[0] = a[0], m[1] = a[1], m[2] = a[2];
        double dist_min = v_norm(a);
        double dist = v_norm(b);
        if (dist_min > dist) {
            m[0] = b[0], m[1] = b[1], m[2] = b[2];
            dist_min = dist;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) n[k] = b[k] - a[k];
        v_normalize(n);
        float adn = (float)v_dot(a, n);
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = a[k] - n[k] * adn;
        dist = v_norm(c);
        float ca[3], cb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) ca[k] = c[k] - a[k], cb[k] = c[k] - b[k];
        if (v_dot(ca, cb) < 0 && dist_min > dist) m[0] = c[0], m[1] = c[1], m[2] = c[2];
    }
    v_normalize(m);
    m[2] = (float)((double)m[2] / dw);  // :384
    float* out = s.rsfc_normal + (((size_t)mission * s.npair + pair) * M + seg) * 3;
    out[0] = m[0], out[1] = m[1], out[2] = m[2];
    if (v_norm(m) == 0) atomicCAS(&s.status[mission], 0, (int)RBP_ERR_INIT_TRAJ_COLLIDE);  // :385-388
}

}  // namespace

void launch_corridor(const DevSession& s, hipStream_t st) {
    // updateObsBox() && updateRelBox() (:25): RSFC results are only meaningful if SFC succeeded; status keeps the
    // first error, with SFC errors taking precedence because sfc_kernel is enqueued first.
    const size_t lds = sizeof(int) * (size_t)SFC_WAVES * s.max_boxes * (s.M + 1);
    const int groups = (s.agent_end - s.agent_begin + SFC_WAVES - 1) / SFC_WAVES;
    (void)hipFuncSetAttribute((const void*)sfc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (groups > 0) hipLaunchKernelGGL(sfc_kernel, dim3(s.K * groups), dim3(64 * SFC_WAVES), lds, st, s);
    const long long total = (long long)s.K * s.npair * s.M;
    if (total > 0) hipLaunchKernelGGL(rsfc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s);
}
