// rbp_dev.h — device-side layout of one session (K missions with common N, per-mission M) and kernel entry points.
//
// Every mission has its own segment count M_k = makespan + 2 (ecbs_planner.hpp:41-43, rbp_planner.hpp:35).  A mission's
// slot in every array is sized for the session's largest M (DevSession::M, the STRIDE), but inside its slot the data is
// laid out compactly with the mission's own M_k = Mk[mission] -- exactly the layout a one-mission session would have.
//
// HBM layout (all arrays mission-major, SoA inside a mission; sizes for the headline N=64, M=36):
//   dist      [K] float grids, x-major / z-fastest (0.94 MB each)        — read by the SFC kernel
//   init_traj [K][N][M+1][3] f32 (28 KB)  T [K][M+1] f64
//   start/goal[K][N][9] f64   radius [K][N]   max_vel/max_acc [K][N][3]
//   sfc_count [K][N] i32  sfc_box [K][N][MB][6] f64  sfc_time [K][N][MB] f64
//   rsfc_normal [K][N(N-1)/2][M][3] f32 (871 KB)   rsfc_time [K][M] f64
//   ctrl / coef [K][N][3][6M] f64 (332 KB each)
//   QP workspace per mission (see qp.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rbp.h"

#define SP_EPSILON 1e-9        /* reference: swarm_planner/include/sp_const.hpp:3 */
#define SP_EPSILON_FLOAT 1e-6  /* sp_const.hpp:4 */
#define QP_MAX_NB 64           /* widest batch (agents) one workgroup factorises: nk = 9 * 64 = 576 */
#define QP_MAX_M 128           /* most segments per mission the QP kernel takes: 64 LDS step counters for (M - 1) / 2 chain steps, see twisted_factor */
inline int planner_max_batch() { return QP_MAX_NB; }
#define SFC_MAXS 512           /* sfc_kernel: max samples per axis (world extent / box resolution + 3); checked at session create */
#define SFC_MASK_WORDS 8192    /* occupancy bitmask of a grid: 262144 cells = 32 KB; larger grids are read as floats */

struct DevWorld {
    int dim[3];
    int key_min[3];
    double res;
    const float* dist;  // device
};

struct DevParam {
    double world_min[3], world_max[3];
    double box_xy_res, box_z_res, downwash;
    int sequential, batch_size, batch_iter, iteration, time_scale;
    int timescale_rule;  // rbp_param.timescale_rule (RBP_TIMESCALE_*)
    int polish;  // 1: active-set polish after the interior-point solve (default)
    double far_slack;  // rbp_solver_opts.qp_far_slack (kernels/qp.hip QP_FAR_SLACK); <= 0: every row near
};

// per-session pointers handed to kernels by value
struct DevSession {
    int K, N, M, max_boxes, npair;  // M, max_boxes: session maxima = slot strides; per-mission values in Mk / MBk
    const int* Mk;            // [K] segments of mission k (<= M)
    const int* MBk;           // [K] box capacity of mission k (<= max_boxes): plan.max_boxes of the caller
    int sfc_cap[3];           // sfc_kernel: capacity of the per-axis key lists = world extent / box resolution + 4 (<= SFC_MAXS)
    int sfc_mask_words;       // sfc_kernel: words of LDS reserved for the occupancy mask (largest grid of the session that fits, multiple of 4)
    int agent_begin, agent_end;  // corridor stage only: agents (and pair rows i) of this shard, [0, N) when not sharded
    DevParam p;
    const DevWorld* worlds;   // [K]
    const float* init_traj;   // [K][N][M+1][3]
    double* T;                // [K][M+1]
    const double* start;      // [K][N][9]
    const double* goal;       // [K][N][9]
    const double* radius;     // [K][N]
    const double* max_vel;    // [K][N][3]
    const double* max_acc;    // [K][N][3]
    unsigned* sfc_mask;       // [K][SFC_MASK_WORDS] bit = dist < radius(agent 0) - 1e-6, written by mask_kernel, read by sfc_kernel
    int* sfc_count;           // [K][N]
    double* sfc_box;          // [K][N][MB][6]
    double* sfc_time;         // [K][N][MB]
    float* rsfc_normal;       // [K][npair][M][3]
    double* rsfc_time;        // [K][M]
    double* ctrl;             // [K][N][3][6M]
    double* coef;             // [K][N][3][6M]
    int* status;              // [K] first error per mission (0 ok)
    double* scalars;          // [K][8]: time_scale, total_cost, ipm_iters, qp_solved, polished, ...
    unsigned long long* counters;  // [K][4]: sfc samples, ...
    // block -> mission order of the QP kernel (r03): a session that is run again starts its LONGEST missions first (wall-clock cycles
    // of each mission's workgroup in the previous run, kept across rbp_session_reset); identity on the first run.  Results do not
    // depend on the order (every mission is a workgroup of its own); nullptr disables it (RBP_QP_ORDER=0).
    int* qp_order;                 // [K]
    unsigned long long* qp_cost;   // [K]
};

enum { SC_TIME_SCALE = 0, SC_TOTAL_COST = 1, SC_IPM_ITERS = 2, SC_QP_SOLVED = 3, SC_POLISHED = 4, SC_FLOPS = 5, SC_ROWS = 6,
       SC_KKT_MAX = 7,    // max over the batch QPs of the KKT residual of the accepted answer (see rbp_plan::kkt_max)
       SC_PROF0 = 8,      // SC_PROF0..SC_PROF0+15: per-phase cycle counters (QP_PROFILE builds); slot 8 otherwise: which batch was not polished
       SC_ROW_BYTES = 24, // algorithmic HBM bytes of the QP kernel (row state, row constants, knot blocks)
       SC_SWEEP_BYTES = 28, // ... the part of SC_ROW_BYTES the three row sweeps of an interior-point iteration stream (bench.py: sweep_phase_gbs)
       SC_TIME_SCALE_ALT = 32, // the factor the other rule of rbp_param.timescale_rule gives (rbp_plan::time_scale_alt)
       SC_N = 36 };
enum { CT_SFC_SAMPLES = 0, CT_N = 4 };

int rbp_set_error(int code, const char* msg);  // abi/session.hip: records the message rbp_last_error() returns, returns code

// launchers (defined in the .hip files)
int launch_corridor(const DevSession& s, hipStream_t st);
size_t corridor_lds_bytes(const DevSession& s);  // dynamic LDS of sfc_kernel for this session's shapes
// kernels/qp.hip is built twice: _w2 = 256 VGPRs, one workgroup per CU; _w4 = 128 VGPRs, two workgroups per CU
void launch_planner_w2(const DevSession& s, void* qp_ws, size_t qp_ws_bytes_per_mission, hipStream_t st);
void launch_planner_w4(const DevSession& s, void* qp_ws, size_t qp_ws_bytes_per_mission, hipStream_t st);
// the phase-split schedule of the batch QPs (kernels/qp_phase.inc; exported by the 256-thread build): gs[G] group streams, gev[1 + G] events
void launch_planner_phased(const DevSession& s, void* qp_ws, size_t qp_ws_bytes_per_mission, hipStream_t st, hipStream_t* gs, hipEvent_t* gev, int G,
                           int rounds);
bool planner_has_phase_split();  // false in the release library (kernels/qp_phase.inc is compiled by `make dev` only)
size_t planner_workspace_bytes_w2(int N, int M, int batch_size_eff);
size_t planner_workspace_bytes_w4(int N, int M, int batch_size_eff);
