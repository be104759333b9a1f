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

// SFC_MAXS (max samples per axis) is defined in rbp_dev.h: rbp_session_create rejects worlds/resolutions that exceed it
#ifndef SFC_WAVES
#define SFC_WAVES 8         // wavefronts per workgroup (one agent each at a time), sharing one occupancy bitmask in LDS
#endif
#ifndef SFC_AGENT_ROUNDS
#define SFC_AGENT_ROUNDS 8  // large sessions: a workgroup's waves work through SFC_WAVES * SFC_AGENT_ROUNDS agents (see sfc_kernel)
#endif
typedef short sfc_key_t;             // a voxel index along one axis (< 4096), -1 outside the grid
typedef unsigned short sfc_log_t;    // box_log entries: a run length of waypoints (<= M + 1)

namespace {

struct AxisCache {
    double lo, hi;  // extent the cached keys belong to
    int n;          // number of samples
    unsigned zmask; // OR of 1 << key over the samples inside the grid (the column test wants this of the z axis)
    int zneg;       // first sample outside the grid, or -1
};

// Computes the voxel indices of the samples along one axis of `box` (rbp_corridor.hpp:47-63 for that axis):
//   v = lo; c = 0; while (v < hi + 1e-6) { coord = (c == 0 && lo > world_min + 1e-6) ? lo - 1e-6 : v + 1e-6; ... v += res }
// and DynamicEDTOctomap::getDistance's key computation k(v) = floor((1/res_map) * (double)(float)coord) - key_min.
//
// The reference's coordinates are the partial sums of a chain of double additions -- sequential, one lane working while 63 wait --
// and a list of ~100 samples is rebuilt every time the lower end of the box moves.  Instead lane c evaluates
// the closed form t = lo + c*res, which differs from the c-th partial sum by at most d = (c + 4) * 2^-52 * max|v| (c roundings
// of at most half an ulp each in the chain, two in the closed form, doubled), and because k(.) and the loop test are
// MONOTONE in v, k(t - d) == k(t + d) proves that the chain value has that key too (and t + d < lim / t - d >= lim decide
// the loop test).  The sample coordinates sit 1e-6 away from the cell boundaries and d is ~1e-12, so this practically always
// succeeds; if any lane cannot prove its key, the whole list is rebuilt by the chain of additions itself (axis_keys_chain).
// Either way the keys are exactly the reference's.  c0 > 0 continues a list whose upper end grew (same lo).
// zmask / zneg (used for the z axis): OR of 1 << key over the valid samples, index of the first sample outside the grid.
__device__ __noinline__ int axis_keys_chain(sfc_key_t* keys, int cap, double lo, double hi, double step, double world_lo, double rf, int key_min,
                                            int dim, AxisCache* out) {
    int c = 0;
    unsigned zm = 0;
    int zn = -1;
    for (double v = lo; v < hi + SP_EPSILON_FLOAT && c < cap; v += step, ++c) {  // every lane runs the same loop
        double coord = v + SP_EPSILON_FLOAT;
        if (c == 0 && lo > world_lo + SP_EPSILON_FLOAT) coord = lo - SP_EPSILON_FLOAT;
        const float cf = (float)coord;  // octomap::point3d is float32
        const int k = (int)floor(rf * (double)cf) - key_min;
        const bool in = k >= 0 && k < dim;
        keys[c] = (sfc_key_t)(in ? k : -1);
        if (in && k < 32) zm |= 1u << k;
        if (!in && zn < 0) zn = c;
    }
    out->zmask = zm, out->zneg = zn;  // (LDS: every lane writes the same values)
    return c;
}

// `out` (LDS) receives zmask / zneg; c0 > 0: they continue from out's values
__device__ __forceinline__ int axis_keys(const bool WANT_Z, sfc_key_t* keys, int cap, double lo, double hi, double step, double world_lo,
                                         double rf, int key_min, int dim, int c0, AxisCache* out, int lane) {
    const double lim = hi + SP_EPSILON_FLOAT;
    const double vmax = fmax(fabs(lo), fabs(lim) + step);
    const bool nudge_down = lo > world_lo + SP_EPSILON_FLOAT;
    int c = c0, zn = c0 > 0 ? out->zneg : -1;
    unsigned zm = 0;
    bool unproven = false;
#ifndef SFC_FORCE_CHAIN  // (test builds: always take the chain, to check that both routes give the reference's keys)
    while (c < cap) {
        const int ci = c + lane;
        const double t = lo + (double)ci * step;
        const double dl = ci == 0 ? 0.0 : (double)(ci + 4) * 0x1p-52 * vmax;  // the 0-th partial sum is lo itself
        const double ta = t - dl, tb = t + dl;
        const bool acc = ci < cap && tb < lim;                   // certainly inside the loop
        bool open = ci < cap && !(tb < lim) && !(ta >= lim);     // cannot tell
        bool in = false;
        int k = -1;
        if (acc) {
            const double ca = (ci == 0 && nudge_down) ? lo - SP_EPSILON_FLOAT : ta + SP_EPSILON_FLOAT;
            const double cb = (ci == 0 && nudge_down) ? lo - SP_EPSILON_FLOAT : tb + SP_EPSILON_FLOAT;
            const int ka = (int)floor(rf * (double)(float)ca), kb = (int)floor(rf * (double)(float)cb);
            open = ka != kb;
            k = ka - key_min;
            in = k >= 0 && k < dim;
        }
        if (__ballot(open)) {  // (a lane beyond the end of the list is never open: the acceptance test is monotone in c)
            unproven = true;
            break;
        }
        const int nacc = __popcll(__ballot(acc));  // a prefix of the lanes
        if (acc) {
            keys[ci] = (sfc_key_t)(in ? k : -1);
            if (WANT_Z && in && k < 32) zm |= 1u << k;
        }
        if (WANT_Z) {
            const unsigned long long nb = __ballot(acc && !in);
            if (nb && zn < 0) zn = c + __ffsll((long long)nb) - 1;
        }
        c += nacc;
        if (nacc < 64) break;
    }
#else
    unproven = true;
#endif
    if (unproven) return axis_keys_chain(keys, cap, lo, hi, step, world_lo, rf, key_min, dim, out);
    if (WANT_Z) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) zm |= __shfl_xor(zm, o);
        out->zmask = (c0 > 0 ? out->zmask : 0u) | zm, out->zneg = zn;
    }
    return c;
}

#define SFC_SLAB 8  // samples of the short axis of a slab test (one box_res step: two or three)

struct SfcCtx {
    const unsigned* mask;  // LDS occupancy bitmask (bit = dist < margin - 1e-6) or nullptr
    const float* grid;
    int dim[3], key_min[3];
    double rf, world_min[3], world_max[3], res[3];
    double margin_cmp;  // margin - 1e-6
    sfc_key_t* keys[3];   // LDS, full extent of the box along each axis
    sfc_key_t* skeys[3];  // LDS, the short axis of the slab under test
    int cap[3];           // capacity of keys[a]
    AxisCache* cache;     // LDS (wave-private): extents and sizes of the key lists keys[] / skeys[] hold
    AxisCache* slab;
    unsigned long long samples;
#ifdef SFC_PROFILE
    long long t_keys, t_samp;
#endif
};

// rbp_corridor.hpp:44-78.  Returns true if any sample reads dist < margin - 1e-6 (or lies outside the grid: -1).
__device__ bool is_obstacle_in_box(SfcCtx& c, const double* box, int lane) {
    // Two key lists per axis: the full extent of the current box and the short axis of the slab under test.  expand_box
    // alternates "slab along e, full along the others", so with one list per axis every test recomputed a full axis; now
    // a full list is rebuilt only when its lower end moves, and extended by one sample when its upper end grows.
#ifdef SFC_PROFILE
    const long long pt0 = wall_clock64();
#endif
    int n[3];
    const sfc_key_t* kp[3];
    unsigned zmask = 0;  // of the list kp[2] points at: the z cells the box touches
    int zneg = -1;       // ... first z sample outside the grid (getDistance = -1 there): every column "hits" at that sample
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = box[a], hi = box[a + 3];
        AxisCache& f = c.cache[a];
        AxisCache& sl = c.slab[a];
        bool slab = false;
        if (f.lo == lo && f.hi == hi) {
        } else if (sl.lo == lo && sl.hi == hi) {
            slab = true;
        } else if (f.lo == lo && hi > f.hi && f.n > 0) {  // upper end grew: the samples so far stay, append the new ones
            f.n = axis_keys(a == 2, c.keys[a], c.cap[a], lo, hi, c.res[a], c.world_min[a], c.rf, c.key_min[a], c.dim[a], f.n, &f, lane);
            f.hi = hi;
        } else if (hi - lo < (SFC_SLAB - 3) * c.res[a]) {
            sl.n = axis_keys(a == 2, c.skeys[a], SFC_SLAB, lo, hi, c.res[a], c.world_min[a], c.rf, c.key_min[a], c.dim[a], 0, &sl, lane);
            sl.lo = lo, sl.hi = hi;
            slab = true;
        } else {
            f.n = axis_keys(a == 2, c.keys[a], c.cap[a], lo, hi, c.res[a], c.world_min[a], c.rf, c.key_min[a], c.dim[a], 0, &f, lane);
            f.lo = lo, f.hi = hi;
        }
        kp[a] = slab ? c.skeys[a] : c.keys[a], n[a] = slab ? sl.n : f.n;
        if (a == 2) zmask = slab ? sl.zmask : f.zmask, zneg = slab ? sl.zneg : f.zneg;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the key lists are written by one lane each and read by all
    __builtin_amdgcn_wave_barrier();
#ifdef SFC_PROFILE
    const long long pt1 = wall_clock64();
    c.t_keys += pt1 - pt0;
    struct Tm { SfcCtx& c; long long t; __device__ ~Tm() { c.t_samp += wall_clock64() - t; } } tm_{c, pt1};
#endif
    const long long total = (long long)n[0] * n[1] * n[2];
    if (total == 0) return false;
    if (c.mask && c.dim[2] <= 32) {
        // COLUMN TEST: z is the fastest dimension of the occupancy bitmask, so the n2 samples of one (x, y) column are
        // nz consecutive bits.  One lane tests a whole column: window of the bitmask at the column's bit offset, ANDed with
        // the set of z cells the box touches.  The columns are visited in the reference's (x outer, y inner) order, 64 per
        // step, and on a hit the first occupied z sample of the first occupied column is located, so the boolean AND the
        // number of getDistance calls the reference would have made (early exit at the first obstacle) are unchanged --
        // with n2 times fewer steps.
        const int n2 = n[2], n1 = n[1], nyz = c.dim[1] * c.dim[2], nzz = c.dim[2];
        const long long ncol = (long long)n[0] * n1;
        int c1 = lane % n1, c0 = lane / n1;
        const int d1 = 64 % n1, d0 = 64 / n1;
        for (long long base = 0; base < ncol; base += 64) {
            bool hit = false;
            int ix = 0, iy = 0;
            unsigned o = 0;
            if (base + lane < ncol) {
                ix = kp[0][c0], iy = kp[1][c1];
                if ((ix | iy) < 0) {
                    hit = true;
                } else {
                    o = (unsigned)ix * nyz + (unsigned)iy * nzz;
                    const unsigned long long w2 = ((unsigned long long)c.mask[(o >> 5) + 1] << 32) | c.mask[o >> 5];
                    hit = (((unsigned)(w2 >> (o & 31))) & zmask) != 0 || zneg >= 0;
                }
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                const int L = __ffsll((long long)m) - 1;
                int zf = 0;
                if (lane == L && (ix | iy) >= 0) {
                    zf = n2 - 1;
                    for (int q = 0; q < n2; ++q) {
                        const int kz = kp[2][q];
                        const bool h = kz < 0 || ((c.mask[(o + kz) >> 5] >> ((o + kz) & 31)) & 1u);
                        if (h) {
                            zf = q;
                            break;
                        }
                    }
                }
                zf = __shfl(zf, L);
                c.samples += (unsigned long long)((base + L) * n2 + zf + 1);  // the reference stops at the first hit
                return true;
            }
            c1 += d1;
            if (c1 >= n1) c1 -= n1, c0++;
            c0 += d0;
        }
        c.samples += (unsigned long long)total;
        return false;
    }
    // mixed-radix decomposition of the lane id and of the stride 64 in (n0, n1, n2), z fastest
    int c2 = lane % n[2], t = lane / n[2];
    int c1 = t % n[1], c0 = t / n[1];
    const int d2 = 64 % n[2], t64 = 64 / n[2];
    const int d1 = t64 % n[1], d0 = t64 / n[1];
    const int ny = c.dim[1], nz = c.dim[2];
    // Four 64-sample steps per trip: the dependent LDS reads (key -> occupancy word) of the four steps overlap, the ballots
    // are then examined in the reference's order, so the early exit and the sample count stay exactly the reference's.
    constexpr int SU = 4;
    for (long long base = 0; base < total; base += 64 * SU) {
        int kx[SU], ky[SU], kz[SU];
        bool live[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            live[u] = base + 64 * u + lane < total;
            kx[u] = live[u] ? kp[0][c0] : 0, ky[u] = live[u] ? kp[1][c1] : 0, kz[u] = live[u] ? kp[2][c2] : 0;
            c2 += d2;
            if (c2 >= n[2]) c2 -= n[2], c1++;
            c1 += d1;
            if (c1 >= n[1]) c1 -= n[1], c0++;
            c0 += d0;
        }
        bool hit[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            hit[u] = false;
            if (live[u]) {
                if ((kx[u] | ky[u] | kz[u]) < 0) {
                    hit[u] = true;  // getDistance returns -1 outside the map
                } else {
                    const unsigned cell = ((unsigned)kx[u] * ny + ky[u]) * nz + kz[u];
                    if (c.mask) {
                        hit[u] = (c.mask[cell >> 5] >> (cell & 31)) & 1u;
                    } else {
                        float d = c.grid[cell];
                        hit[u] = (double)d < c.margin_cmp;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const long long rem = total - (base + 64 * u);
            if (rem <= 0) break;
            const unsigned long long m = __ballot(hit[u]);
            if (m) {
                c.samples += (unsigned long long)(__ffsll((long long)m));  // the reference stops at the first hit
                return true;
            }
            c.samples += (unsigned long long)(rem < 64 ? rem : 64);
        }
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

// One bit per grid cell: dist < radius - 1e-6 for the radius of the mission's first agent (agents with another radius read the
// float grid).  Built once per mission and launch -- the 16 workgroups of a mission used to rebuild it from the 0.94 MB grid each.
// Coalesced reads, one ballot per 64 cells; grid = (SFC_MASK_BLOCKS, K).
#define SFC_MASK_BLOCKS 8
__global__ __launch_bounds__(256) void mask_kernel(DevSession s) {
    const int mission = blockIdx.y, lane = threadIdx.x & 63;
    const DevWorld w = s.worlds[mission];
    const unsigned ncell = (unsigned)w.dim[0] * w.dim[1] * w.dim[2];
    if (ncell + 64 > 32u * SFC_MASK_WORDS) return;
    const double cmp0 = s.radius[(size_t)mission * s.N] - SP_EPSILON_FLOAT;
    unsigned* mask = s.sfc_mask + (size_t)mission * SFC_MASK_WORDS;
    const unsigned nchunk = (ncell + 63) >> 6, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = SFC_MASK_BLOCKS * 4;
    for (unsigned c0 = wave; c0 < nchunk; c0 += 4 * nwave) {
        bool occ[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned cell = (c0 + u * nwave) * 64 + lane;
            occ[u] = cell < ncell && (double)w.dist[cell] < cmp0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned c = c0 + u * nwave;
            const unsigned long long b = __ballot(occ[u]);
            if (lane == 0 && c < nchunk) mask[2 * c] = (unsigned)b, mask[2 * c + 1] = (unsigned)(b >> 32);
        }
    }
}

#ifndef SFC_WAVES_PER_EU
#ifndef SFC_WAVES_PER_EU
#define SFC_WAVES_PER_EU 4
#endif
#endif
// apw = agents per workgroup (a multiple of SFC_WAVES).  apw == SFC_WAVES: one agent per wave, as many workgroups in flight as possible (small
// sessions: a lone mission is eight workgroups on eight CUs).  apw > SFC_WAVES (round 6, sessions of several rounds of workgroups): the
// wavefronts of a workgroup TAKE the agents of its range one after the other from a counter in LDS -- an agent's corridor takes anything from
// 40 to 200 us, and with one agent per wave a workgroup (its LDS, its wave slots) lived as long as its slowest agent while the other seven
// slots idled; now a wave that is done starts the next agent.  Which wave grows which agent's boxes changes nothing about them.
__global__ __launch_bounds__(64 * SFC_WAVES, SFC_WAVES_PER_EU) void sfc_kernel(DevSession s, int apw) {
#ifdef SFC_PROFILE
    const long long t_start = wall_clock64();
#endif
    const int groups = (s.agent_end - s.agent_begin + apw - 1) / apw;
    // the wave index is wave-uniform by construction; as a plain threadIdx expression the compiler must treat it (and the agent index, every
    // pointer and the whole box state derived from it) as divergent, i.e. keep them in vector registers
    const int mission = blockIdx.x / groups, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a_lo = s.agent_begin + (blockIdx.x % groups) * apw, a_hi = a_lo + apw < s.agent_end ? a_lo + apw : s.agent_end;
    __shared__ int next_agent;  // next agent of [a_lo, a_hi) nobody has taken yet (the first SFC_WAVES go to the waves by index)
    if (threadIdx.x == 0) next_agent = SFC_WAVES;
    const int M = s.Mk[mission], P = M + 1, PS = s.M + 1, MB = s.max_boxes, MBcap = s.MBk[mission];  // PS, MB: slot strides
    // LDS: [occupancy mask: s.sfc_mask_words words][per wave: key lists x | y | z, three slab lists][per wave: box_log [MB][P]]
    extern __shared__ __attribute__((aligned(16))) unsigned sfc_lds[];
    unsigned* mask = sfc_lds;
    const int kw = (s.sfc_cap[0] + s.sfc_cap[1] + s.sfc_cap[2] + 3 * SFC_SLAB + 1) & ~1;
    sfc_key_t* kbase = (sfc_key_t*)(sfc_lds + s.sfc_mask_words) + (size_t)wave * kw;
    sfc_log_t* box_log = (sfc_log_t*)((sfc_key_t*)(sfc_lds + s.sfc_mask_words) + (size_t)SFC_WAVES * kw) + (size_t)wave * MB * PS;
    const DevWorld w = s.worlds[mission];
    // ---- occupancy bitmask of this mission's grid (mask_kernel below; 29 KB for the 101x101x23 grid) into LDS.  The SFC test
    // only needs "dist < r - 1e-6" (rbp_corridor.hpp:67), so one bit per cell replaces ~0.6 M float reads per agent.
    const double radius0 = s.radius[(size_t)mission * s.N];
    const unsigned ncell = (unsigned)w.dim[0] * w.dim[1] * w.dim[2];
    const unsigned nq = (((ncell + 63) >> 6) * 2 + 2 + 3) >> 2;  // 16-byte quads; the column test reads one word past the last cell
    const bool mask_fits = 4 * nq <= (unsigned)s.sfc_mask_words;
    if (mask_fits) {
        const uint4* gm = (const uint4*)(s.sfc_mask + (size_t)mission * SFC_MASK_WORDS);
        for (unsigned i = threadIdx.x; i < nq; i += 64 * SFC_WAVES) ((uint4*)mask)[i] = gm[i];
    }
    __syncthreads();
    // what the key lists hold (extent, count, z summary) lives in LDS beside them: wave-uniform state that is read once per box test
    // has no business occupying 42 vector registers for the whole kernel
    __shared__ AxisCache caches[SFC_WAVES][6];
    for (int take = wave;; ) {
    const int qi = a_lo + take;
    if (qi >= a_hi) return;
    {  // the agent after this one (taken now: the atomic's round trip hides behind the set-up below)
        int nx = 0;
        if (lane == 0) nx = atomicAdd(&next_agent, 1);
        take = __builtin_amdgcn_readfirstlane(nx);
    }
    SfcCtx c;
    c.cache = caches[wave], c.slab = caches[wave] + 3;
    c.grid = w.dist;
    c.rf = 1.0 / w.res;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c.dim[a] = w.dim[a], c.key_min[a] = w.key_min[a];
        c.world_min[a] = s.p.world_min[a], c.world_max[a] = s.p.world_max[a];
        c.keys[a] = kbase + (a == 0 ? 0 : a == 1 ? s.sfc_cap[0] : s.sfc_cap[0] + s.sfc_cap[1]);
        c.skeys[a] = kbase + s.sfc_cap[0] + s.sfc_cap[1] + s.sfc_cap[2] + a * SFC_SLAB;
        c.cap[a] = s.sfc_cap[a];
        c.cache[a].lo = 1e300, c.cache[a].hi = -1e300, c.cache[a].n = 0, c.cache[a].zmask = 0, c.cache[a].zneg = -1;
        c.slab[a] = c.cache[a];
    }
    c.res[0] = c.res[1] = s.p.box_xy_res, c.res[2] = s.p.box_z_res;
    const double radius = s.radius[(size_t)mission * s.N + qi];
    c.margin_cmp = radius - SP_EPSILON_FLOAT;
    c.mask = (mask_fits && radius == radius0) ? mask : nullptr;  // agents with another radius read the float grid
    c.samples = 0;
#ifdef SFC_PROFILE
    c.t_keys = c.t_samp = 0;
    const long long t_after_mask = wall_clock64();
#endif

    const float* traj = s.init_traj + (size_t)mission * s.N * PS * 3 + (size_t)qi * P * 3;
    const double* T = s.T + (size_t)mission * PS;
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
        if (nbox >= MBcap) {
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
        continue;
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
    for (int b = lane; b < nbox; b += 64)  // the running counts: one box per lane
        for (int j = 1; j < P; ++j)
            if (box_log[b * P + j]) box_log[b * P + j] = box_log[b * P + j - 1] + 1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
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
#ifdef SFC_PROFILE
        atomicAdd(&s.scalars[(size_t)mission * SC_N + 20], (double)c.t_keys);
        atomicAdd(&s.scalars[(size_t)mission * SC_N + 21], (double)c.t_samp);
        atomicAdd(&s.scalars[(size_t)mission * SC_N + 22], (double)(wall_clock64() - t_after_mask));
        if (qi == a_lo + wave) atomicAdd(&s.scalars[(size_t)mission * SC_N + 23], (double)(t_after_mask - t_start));
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the next agent reuses this wave's key lists and box_log
    __builtin_amdgcn_wave_barrier();
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
    const int MS = s.M, PS = MS + 1, N = s.N;  // slot strides
    const long long per_mission = (long long)s.npair * MS;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= per_mission * s.K) return;
    const int mission = (int)(gid / per_mission);
    const int M = s.Mk[mission], P = M + 1;
    const long long r = gid % per_mission;
    const int pair = (int)(r / MS), seg = (int)(r % MS);
    if (seg >= M) return;
    // invert pair index -> (qi, qj), qi < qj, qi-major (rbp_corridor.hpp:342-344)
    int qi = 0, rem = pair;
    while (rem >= N - 1 - qi) rem -= N - 1 - qi, qi++;
    const int qj = qi + 1 + rem;
    if (pair == 0) s.rsfc_time[(size_t)mission * MS + seg] = s.T[(size_t)mission * PS + seg + 1];  // :390 (every shard)
    if (qi < s.agent_begin || qi >= s.agent_end) return;  // pair rows of another shard
    const float* ti = s.init_traj + (size_t)mission * N * PS * 3 + (size_t)qi * P * 3 + 3 * seg;
    const float* tj = s.init_traj + (size_t)mission * N * PS * 3 + (size_t)qj * P * 3 + 3 * seg;
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
    float* out = s.rsfc_normal + (size_t)mission * s.npair * MS * 3 + ((size_t)pair * M + seg) * 3;
    out[0] = m[0], out[1] = m[1], out[2] = m[2];
    if (v_norm(m) == 0) atomicCAS(&s.status[mission], 0, (int)RBP_ERR_INIT_TRAJ_COLLIDE);  // :385-388
}

}  // namespace

// dynamic LDS of sfc_kernel: occupancy mask + per wave the key lists and box_log [max_boxes][M + 1] (checked against the CU's LDS when a
// session is created: abi/session.hip)
size_t corridor_lds_bytes(const DevSession& s) {
    const int kw = (s.sfc_cap[0] + s.sfc_cap[1] + s.sfc_cap[2] + 3 * SFC_SLAB + 1) & ~1;
    return sizeof(unsigned) * s.sfc_mask_words + sizeof(sfc_key_t) * (size_t)SFC_WAVES * kw +
           sizeof(sfc_log_t) * (size_t)SFC_WAVES * s.max_boxes * (s.M + 1) + 16;
}

int launch_corridor(const DevSession& s, hipStream_t st) {
    // updateObsBox() && updateRelBox() (:25): RSFC results are only meaningful if SFC succeeded; status keeps the
    // first error, with SFC errors taking precedence because sfc_kernel is enqueued first.
    const size_t lds = corridor_lds_bytes(s);
    // agents per workgroup: sessions whose one-agent-per-wave grid would be more than two rounds of workgroups on the chip let the waves
    // take agents from a counter (sfc_kernel); smaller ones keep every agent on a wave of its own, all in flight at once
    const int nag = s.agent_end - s.agent_begin;
    const long long wg8 = (long long)s.K * ((nag + SFC_WAVES - 1) / SFC_WAVES);
    const int apw = wg8 > 1024 ? SFC_WAVES * SFC_AGENT_ROUNDS : SFC_WAVES;
    const int groups = (nag + apw - 1) / apw;
    if (hipFuncSetAttribute((const void*)sfc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return rbp_set_error(RBP_ERR_HIP, "sfc_kernel: the dynamic LDS it needs was refused by the device");
    if (groups > 0) hipLaunchKernelGGL(mask_kernel, dim3(SFC_MASK_BLOCKS, s.K), dim3(256), 0, st, s);
    if (groups > 0) hipLaunchKernelGGL(sfc_kernel, dim3(s.K * groups), dim3(64 * SFC_WAVES), lds, st, s, apw);
    const long long total = (long long)s.K * s.npair * s.M;
    if (total > 0) hipLaunchKernelGGL(rsfc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s);
    return RBP_OK;
}
