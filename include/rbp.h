/* rbp.h — C ABI of the MI355X-native RBP plan path (SFC + RSFC + QP).
 *
 * This is the drop-in boundary for the two header-only C++ stages of the reference
 *   SwarmPlanning::Corridor::update(bool, PlanResult*)    swarm_planner/include/rbp_corridor.hpp:21-26
 *   SwarmPlanning::RBPPlanner::update(bool, PlanResult*)  swarm_planner/include/rbp_planner.hpp:33-84
 * The reference has no FFI; the shared state between the stages is `PlanResult`
 * (swarm_planner/include/sp_const.hpp:21-28).  Here the same state is a set of caller-owned flat
 * arrays (`rbp_plan`), the distance map is a flat float grid (`rbp_world`, what
 * DynamicEDTOctomap::getDistance serves in rbp_corridor.hpp:66) and Mission/Param are plain structs
 * (mission.hpp:13-15, param.hpp:44-70).  INTEGRATION.md shows the ~80-line adapter a maintainer of
 * the reference would add.
 *
 * All pointers are HOST pointers unless a function says otherwise.  Plain C, no torch types.
 */
#ifndef RBP_H
#define RBP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (the reference returns bool + ROS_ERROR; 0 == true) ------------------------- */
enum {
    RBP_OK = 0,
    RBP_ERR_OBSTACLE_IN_INIT_TRAJ = 1, /* rbp_corridor.hpp:181-187 "Obstacle invades initial trajectory" */
    RBP_ERR_UNEQUAL_TRAJ_LEN = 2,      /* rbp_corridor.hpp:346-349 (cannot happen with the flat layout; kept) */
    RBP_ERR_INIT_TRAJ_COLLIDE = 3,     /* rbp_corridor.hpp:385-388 "initial trajectories are collided" */
    RBP_ERR_SFC_OVERFLOW = 4,          /* more boxes than plan->max_boxes (flat-layout only) */
    RBP_ERR_QP_FAILED = 10,            /* rbp_planner.hpp:158-161 "Failed to optimize QP" (infeasible / no convergence) */
    RBP_ERR_UNSUPPORTED_DEGREE = 11,   /* rbp_planner.hpp:344-346, 375-377: only n=5, phi=3 */
    RBP_ERR_BAD_ARGUMENT = 20,
    RBP_ERR_NO_DEVICE = 30,            /* HIP device / kernel image unavailable: the product path never falls back to CPU */
    RBP_ERR_HIP = 31,
    RBP_ERR_EXCHANGE = 32              /* rbp_session_shard_joint: the caller's exchange hook reported a failure */
};

/* ---- distance map: what DynamicEDTOctomap(maxDist=1, tree, bbxMin, bbxMax, false) serves -------
 * swarm_planner/src/swarm_traj_planner_rbp.cpp:73-80.  Cell (ix,iy,iz) holds the voxel whose octomap key
 * (minus 32768) is key_min + (ix,iy,iz); a point p maps to key floor((1/res) * (double)p) per axis and
 * reads -1 outside the grid (so out-of-world samples count as obstacle in rbp_corridor.hpp:67). */
typedef struct rbp_world {
    int32_t dim[3];     /* nx, ny, nz */
    int32_t key_min[3]; /* voxel key of cell (0,0,0), relative to the octree centre */
    double res;         /* octree resolution [m] */
    const float* dist;  /* [nx][ny][nz], metres, z fastest */
} rbp_world;

/* ---- mission.hpp:13-15 ---------------------------------------------------------------------- */
typedef struct rbp_mission {
    int32_t N;             /* qn */
    const double* start;   /* [N][9]  pos(3) vel(3) acc(3)   mission.hpp:47-53 */
    const double* goal;    /* [N][9]                         mission.hpp:55-61 */
    const double* radius;  /* [N]     quad_size              mission.hpp:64 */
    const double* max_vel; /* [N][3]                         mission.hpp:70-76 */
    const double* max_acc; /* [N][3]                         mission.hpp:78-83 */
} rbp_mission;

/* ---- param.hpp:44-70 (names and defaults are the reference's) -------------------------------- */
typedef struct rbp_param {
    double world_min[3]; /* world/x_min,y_min,z_min  (-5,-5,0)   */
    double world_max[3]; /* world/x_max,y_max,z_max  (5,5,2.5)   */
    double box_xy_res;   /* box/xy_res 0.1 */
    double box_z_res;    /* box/z_res  0.1 */
    double downwash;     /* plan/downwash 2.0 */
    double time_step;    /* plan/time_step 1 */
    double ecbs_w;       /* ecbs/w 1.3        (front-end only) */
    double grid_xy_res;  /* grid/xy_res 0.3   (front-end only) */
    double grid_z_res;   /* grid/z_res 0.6    (front-end only) */
    double grid_margin;  /* grid/margin 0.2   (front-end only) */
    int32_t n;           /* plan/n 5   */
    int32_t phi;         /* plan/phi 3 */
    int32_t sequential;  /* plan/sequential false */
    int32_t batch_size;  /* plan/batch_size 4 */
    int32_t batch_iter;  /* plan/batch_iter 0 (launch files pass -1 = all batches) */
    int32_t iteration;   /* plan/iteration 1 */
    int32_t time_scale;  /* plan/time_scale true */
    int32_t log;         /* log false */
    int32_t timescale_rule; /* NOT a key of the reference (ABI 6): which candidate times timeScale's velocity check inspects -- see below */
} rbp_param;

/* rbp_param.timescale_rule.  scale_to_max_vel (rbp_planner.hpp:756-794) evaluates |velocity| at t = 0, t = dt and at the roots
 * roots_derivative(2, coef_der) returns (:727-754): eigenvalues of the cubic's companion matrix from Eigen::EigenSolver, of which the loop
 * `for (j = 0; j < i; j++)` (:746, i = 2 = the derivative order) reads only the FIRST TWO, in Eigen's internal order.
 *   RBP_TIMESCALE_ALL_REAL_ROOTS (0, the default): every real root of the cubic is a candidate.  The factor can only be >= the reference's
 *     (more candidates), is still a power of 1.1, and respects max_vel at EVERY velocity extremum -- the reference's literal rule can leave
 *     a segment above max_vel when the skipped third eigenvalue holds the peak.  Independent of any eigenvalue ordering.
 *   RBP_TIMESCALE_FIRST_EIGENVALUES (1): the loop as written -- the real ones among the first two eigenvalues, in the order of the real
 *     Schur decomposition of Eigen 3.3.x (Francis double-shift QR on the companion matrix, eigenvalues read off T from the top), restated
 *     from Eigen's published algorithm in kernels/qp.hip; Eigen is an un-pinned system dependency of the reference, so this order is a
 *     documented, deterministic choice, not a pinned one.
 * Both factors are always computed: rbp_plan.time_scale is the selected rule's (and what coef / T / corridor times are scaled by),
 * rbp_plan.time_scale_alt the other rule's -- `time_scale != time_scale_alt` tells a caller that the two rules disagree on this plan
 * (6 of 150 cases on the 50-map sweep with limits scaled by 1 / 0.5 / 0.25, each time by one step of the 1.1 ladder:
 * tests/test_timescale_rule.py). */
enum { RBP_TIMESCALE_ALL_REAL_ROOTS = 0, RBP_TIMESCALE_FIRST_EIGENVALUES = 1 };

/* Fill `p` with the defaults of Param::setROSParam (param.hpp:44-70). */
void rbp_param_defaults(rbp_param* p);

/* ---- PlanResult (sp_const.hpp:21-28) as flat caller-owned arrays ------------------------------
 * pair index of (qi<qj):  qi*N - qi*(qi+1)/2 + (qj-qi-1)   (the order RSFC[qi][qj] is filled in
 * rbp_corridor.hpp:342-344). */
typedef struct rbp_plan {
    int32_t N;              /* agents (== mission.N) */
    int32_t M;              /* segments = T.size()-1            rbp_planner.hpp:35 */
    double* T;              /* [M+1] segment times; rescaled in place by time_scale (rbp_planner.hpp:262-264) */
    const float* init_traj; /* [N][M+1][3] float32 = octomap::point3d waypoints (sp_const.hpp:16) */

    /* SFC_t (sp_const.hpp:17): per agent a list of (box[6] = xmin,ymin,zmin,xmax,ymax,zmax ; end time) */
    int32_t max_boxes;      /* capacity per agent; M always suffices */
    int32_t* sfc_count;     /* [N] */
    double* sfc_box;        /* [N][max_boxes][6] */
    double* sfc_time;       /* [N][max_boxes]     rescaled by time_scale (rbp_planner.hpp:250-252) */

    /* RSFC_t (sp_const.hpp:18): per pair and segment (float32 normal ; time T[m+1]) */
    float* rsfc_normal;     /* [N(N-1)/2][M][3] */
    double* rsfc_time;      /* [M]  (= T[1..M], identical for every pair; rescaled like rbp_planner.hpp:255-258) */

    /* RBPPlanner outputs */
    double* coef;           /* [N][3][6M]: per agent the column-major (6M x 3) matrix the reference copies into
                               msgs_traj_coef[qi].data (rbp_planner.hpp:286-289); rows m*6+i hold the coefficient of
                               (t-T_m)^(5-i), i.e. descending powers, seconds (rbp_planner.hpp:170-186) */
    double* ctrl;           /* [N][3][6M] Bernstein control points (the reference's `dummy`/`vals`), may be NULL */
    double time_scale;      /* rbp_planner.hpp:235 */
    double total_cost;      /* "QP total cost" rbp_planner.hpp:205: sum of batch objectives of the last pass */
    int32_t x_size;         /* count_x  of the last batch  rbp_planner.hpp:58 */
    int32_t eq_size;        /* count_eq                    rbp_planner.hpp:59 */
    int32_t ineq_size;      /* count_lq                    rbp_planner.hpp:60 */
    int32_t qp_iterations;  /* total interior-point iterations spent (diagnostic, not in the reference) */
    /* solver outcome (not in the reference, where cplex.solve() either returns an optimum or throws, rbp_planner.hpp:158-161):
     * every batch QP ends with an active-set polish whose answer is accepted only under a full KKT check; a QP whose polish
     * was refused keeps the interior-point answer (optimal to ~1e-5 m instead of ~1e-8 m) and is counted here */
    int32_t qp_solves;      /* batch QPs solved (passes x batches) */
    int32_t qp_unpolished;  /* of those, how many kept the interior-point answer; 0 = every answer is a certified optimum */
    double kkt_max;         /* max over the batch QPs of the accepted answer's KKT residual: polished QPs max(row violation [m],
                               -min multiplier / max(1, max multiplier)); unpolished QPs max(primal residual [m], relative dual
                               residual, complementarity mu) */
    double time_scale_alt;  /* (ABI 6) the factor the OTHER rule of rbp_param.timescale_rule gives for this plan; != time_scale: the rules disagree */
} rbp_plan;

/* ---- the two stage calls (synchronous; results on return) -------------------------------------
 * Corridor::update: reads world, mission.radius, param.{world_*,box_*,downwash}, plan.{T,init_traj};
 * writes plan.{sfc_*, rsfc_*}.  Unlike the reference (which appends, rbp_corridor.hpp:153,190) the
 * outputs are overwritten. */
int rbp_corridor_update(const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan);

/* Agent-sharded Corridor::update for ONE large mission spread over several GPUs (one process per GPU): computes the SFC
 * of agents qi in [agent_begin, agent_end) (the loop body of updateObsBox, rbp_corridor.hpp:154-239) and the RSFC rows
 * of the pairs (qi, qj), qi < qj, with qi in that range (updateRelBox, :342-392); rsfc_time is written by every shard.
 * All other entries of plan.{sfc_*, rsfc_normal} are unspecified: the caller exchanges the shards (all-gather over
 * RCCL/xGMI, swarm_simulator_amd/sharded.py) and obtains exactly what rbp_corridor_update would have written.  Runs on
 * the calling thread's current HIP device. */
int rbp_corridor_update_range(const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan,
                              int32_t agent_begin, int32_t agent_end);

/* RBPPlanner::update: reads mission, param, plan.{T,init_traj,sfc_*,rsfc_*}; writes plan.{coef,ctrl,
 * time_scale,total_cost,*_size} and rescales T / sfc_time / rsfc_time when time_scale != 1. */
int rbp_planner_update(const rbp_mission* mission, const rbp_param* param, rbp_plan* plan);

/* ---- device-resident, batched form (what bench.py times) --------------------------------------
 * A session owns the HBM copies of K independent missions (e.g. the 50-map sweep of
 * swarm_traj_planner_rbp_test_all.cpp:49-103).  `run` only enqueues kernels on `stream`
 * (a hipStream_t passed as void*; NULL = default stream) and never synchronises -- with ONE exception: the PLANNER stage of a
 * non-sequential plan (plan/sequential = false, the reference's code default param.hpp:67: one joint QP over all agents,
 * rbp_planner.hpp:857-859) with rbp_solver_opts.joint_wide_min_agents (16) agents or more runs on the grid-wide solver
 * (kernels/jqp.hip: a launch per phase of the interior-point method over all CUs), whose host loop learns once per iteration whether
 * any mission is still running: that `run` SYNCHRONISES `stream` before it returns. */
typedef struct rbp_session rbp_session;

enum { RBP_STAGE_CORRIDOR = 1, RBP_STAGE_PLANNER = 2, RBP_STAGE_ALL = 3 };

/* ---- solver options (ABI 4) --------------------------------------------------------------------
 * What a caller may legitimately choose about HOW the QPs are solved (never WHAT is computed: every setting ends in the same
 * KKT-verified optimum or the same error).  The library reads NO environment variables: these options are the only switches.
 * Fill the struct with rbp_solver_opts_defaults, change fields, hand it to a session (before its first `run`) or to a context
 * (sessions and one-shot calls made in it inherit the options; ctx == NULL: the calling thread's default context). */
typedef struct rbp_solver_opts {
    int32_t size;                  /* sizeof(rbp_solver_opts) of the caller's header (set by rbp_solver_opts_defaults; checked) */
    int32_t polish;                /* 1: every QP ends with the active-set polish (the certified optimum); 0: the interior-point answer
                                      with its reported KKT residual (diagnostics) */
    int32_t joint_wide_min_agents; /* 16: a joint QP (plan/sequential = false) of at least this many agents runs on the grid-wide
                                      solver (kernels/jqp.hip, no limit on N, `run` synchronises -- rbp_session_run_async does not); fewer agents -- or 0 = never --
                                      run on one workgroup per mission (<= 64 agents, `run` only enqueues) */
    int32_t joint_corrector;       /* 1: one centrality corrector per interior-point iteration of the grid-wide solver */
    int32_t joint_schedule;        /* 0: automatic; 1: look-ahead tile sweep (few missions); 2: bulk tile sweep (many missions); 3: bulk with
                                      two pivot tiles per pass over a knot's matrix (half the HBM traffic of the update; pays only with
                                      hundreds of resident missions: DESIGN.md 3.5) */
    int32_t qp_schedule;           /* batch QPs of the sequential schedule: 0 automatic; 1: one workgroup per mission runs everything
                                      (qp_batch_kernel); 2: phase split -- chip-wide row sweeps as kernels of their own
                                      (kernels/qp_phase.inc) */
    int32_t qp_variant;            /* qp_schedule 1: 0 automatic; 2: 512 threads, one workgroup per CU; 4: 256 threads, two per CU */
    int32_t qp_block_order;        /* qp_schedule 1: 1 = a session that is run again starts its longest missions first; 0 = plain order */
    int32_t qp_groups;             /* qp_schedule 2: streams the missions are spread over (0 automatic, at most 8) */
    int32_t qp_rounds;             /* qp_schedule 2: round budget (0 automatic: 48 per batch QP of the schedule) */
    double qp_far_slack;           /* qp_schedule 1, first Gauss-Seidel pass, polish on: 0.7 [m].  The interior-point phase of a batch QP leaves
                                      out the frozen-neighbour rows whose slack at the starting point exceeds this for a whole (agent, segment)
                                      group; the polish verifies EVERY row, and a batch QP that does not end polished on the reduced set is
                                      solved again with every row -- the answer is the same certified optimum for any value.  <= 0: off */
} rbp_solver_opts;
void rbp_solver_opts_defaults(rbp_solver_opts* o);

/* worlds/missions/plans: arrays of K structs (host side).  All plans share N; every mission keeps its own M (= ECBS makespan
 * + 2, ecbs_planner.hpp:41-43) and max_boxes, exactly as the reference's map sweep plans each map with its own M. */
int rbp_session_create(rbp_session** out, int device, int K, const rbp_world* worlds, const rbp_mission* missions,
                       const rbp_param* param, const rbp_plan* plans);
int rbp_session_run(rbp_session* s, int stages, void* stream);
/* `run` only enqueues on `stream`, with ONE exception: the PLANNER stage of a joint plan on the grid-wide solver (kernels/jqp.hip,
 * rbp_solver_opts.joint_wide_min_agents), whose host loop learns once per interior-point round whether any mission still iterates --
 * there `run` returns when the solve has ended.  rbp_session_run_async is the same call without that exception: the stages before the
 * joint solve are enqueued on `stream`, the solve itself proceeds on a thread and a stream of the session's own (ordered after `stream`'s
 * work so far by an event), and the call returns at once -- the caller overlaps it with host work (the next mission's ECBS) or with
 * other sessions.  Every later call on the session (download, counters, scalars, reset, run, destroy ...) first waits for the solve;
 * rbp_session_wait does only that and returns the solve's status.  What the caller enqueues on `stream` itself after run_async is NOT
 * ordered after the solve.  For every other session rbp_session_run_async is rbp_session_run. */
int rbp_session_run_async(rbp_session* s, int stages, void* stream);
int rbp_session_wait(rbp_session* s);
/* ---- ONE joint QP on TWO GPUs (ABI 5; BASELINE.json config 4 "agents sharded across GPUs") -------------------------
 * The pair rows of the joint QP couple every agent with every other (rbp_planner.hpp:638-684), so the QP does not split by agent; what
 * splits is its Newton system, block tridiagonal over the knots: the grid-wide solver eliminates it from both ends towards the middle
 * knot (two chains), and the factorisation is ~85 % of a 256-agent mission.  After this call the session is rank `rank` of a pair of
 * identical sessions (same missions, same options, one per GPU / process): rank 0 eliminates and substitutes along the lower chain, rank 1
 * along the upper one, both assemble the middle knot; row sweeps, control and polish are replicated and bit-identical, so both ranks
 * end with the SAME bits as an unsharded run.  Three kinds of exchange per interior-point iteration go through the caller's hook
 * (the library owns send_dev / recv_dev, in HBM; the session's stream is synchronised before the hook is called; the hook returns 0 once the
 * peer's `bytes` are in recv_dev, e.g. an all-gather over RCCL / xGMI -- swarm_simulator_amd/sharded.py): the explicit inverse of each
 * chain's last knot (81 N^2 doubles rounded up to 64-wide tiles: 42 MB at 256 agents), and two vectors per Newton solve.  The hook is
 * called on the thread that calls rbp_session_run; rbp_session_run_async is refused for a sharded session.  nranks must be 2 (1 = undo).
 * KEEPING THE RANKS MATCHED.  Only the replicated state keeps the two ranks' exchanges paired, so every exchange begins with a 64-byte
 * header (sequence number, kind, byte count, a hash of the state words the rank polled last, a poison word) that each rank compares with
 * its own after the hook has returned: ranks that have diverged -- or a peer that failed on its side and sent the poison word in place of
 * its next exchange -- end the run with RBP_ERR_EXCHANGE and a message naming the word, instead of a hang or a payload unpacked into the
 * wrong slot.  A hook that fails (non-zero return) ends the run with RBP_ERR_EXCHANGE on that rank; the peer is then blocked in ITS hook, so
 * a hook must not wait for ever: rbp_rccl_exchange (include/rbp_rccl.h) polls with a timeout and aborts its communicator, which releases the
 * peer.  After RBP_ERR_EXCHANGE on either rank BOTH ranks must give the solve up (undo the sharding, make a new pair). */
typedef int (*rbp_exchange_fn)(void* user, void* send_dev, void* recv_dev, size_t bytes);
int rbp_session_shard_joint(rbp_session* s, int32_t rank, int32_t nranks, rbp_exchange_fn exchange, void* user);
/* The same with a STREAM-ORDERED exchange (round 6): the hook only ENQUEUES the exchange of `bytes` bytes on `stream` (a hipStream_t: the stream the
 * run was given) and returns -- ncclSend + ncclRecv in one group on that stream (rbp_rccl_exchange_stream of include/rbp_rccl.h) --, and the library
 * synchronises for no exchange: it packs, writes the header, calls the hook, and enqueues the kernel that compares the peer's header with its own and
 * the kernel that unpacks (which does nothing once a header has not matched); the comparison's verdict is read with the once-per-round poll of the
 * missions' states, and a mismatch ends the run with RBP_ERR_EXCHANGE as above.  ~900 host synchronisations of a 256-agent solve become ~100.
 * A peer that is gone leaves the stream blocked in its receive: the per-round wait therefore polls with a clock, and after `timeout_s` seconds
 * (<= 0: for ever) calls `abort_peer(user)` -- may be NULL; rbp_rccl_abort aborts the communicator, which releases both ranks -- and returns
 * RBP_ERR_EXCHANGE.  Everything else as rbp_session_shard_joint (which it replaces on the session; nranks = 1 undoes either). */
typedef int (*rbp_exchange_stream_fn)(void* user, void* send_dev, void* recv_dev, size_t bytes, void* stream);
typedef int (*rbp_exchange_abort_fn)(void* user);
int rbp_session_shard_joint_stream(rbp_session* s, int32_t rank, int32_t nranks, rbp_exchange_stream_fn exchange, rbp_exchange_abort_fn abort_peer,
                                   void* user, double timeout_s);
/* solver options of this session (default: the context's, else rbp_solver_opts_defaults).  The QP workspace is reserved by the first
 * PLANNER run, for the options then in force: a session that only runs the CORRIDOR stage reserves none. */
int rbp_session_set_solver_opts(rbp_session* s, const rbp_solver_opts* o);
/* restrict the CORRIDOR stage of later `run` calls to agents / pair rows [agent_begin, agent_end) (see
 * rbp_corridor_update_range); [0, N) restores the whole mission */
int rbp_session_set_agent_range(rbp_session* s, int32_t agent_begin, int32_t agent_end);
/* blocks on `stream`, copies outputs into plans[0..K-1]; returns the first non-zero per-mission status
 * and, if `status` != NULL, every mission's status in status[0..K-1]. */
int rbp_session_download(rbp_session* s, rbp_plan* plans, int32_t* status, void* stream);
/* clears the per-run status / diagnostics so that `run` can be repeated (bench loops).  T, the SFC end times and the RSFC
 * times are never rescaled on the device -- rbp_session_download applies time_scale to the host copies (rbp_planner.hpp:
 * 250-264) -- so nothing else has to be restored; corridor inputs handed to rbp_session_create are uploaded again in case
 * a CORRIDOR stage overwrote them. */
int rbp_session_reset(rbp_session* s, void* stream);
void rbp_session_destroy(rbp_session* s);

/* Device views of one mission's corridor arrays inside a session (HIP device pointers into the session's arena, valid until
 * rbp_session_destroy): what an agent-sharded Corridor::update exchanges between GPUs WITHOUT a host round trip --
 * the shard written by a CORRIDOR run restricted with rbp_session_set_agent_range is gathered from / into these arrays on the
 * device (swarm_simulator_amd/sharded.py: torch tensors over the pointers, one all_gather_into_tensor over RCCL), and the PLANNER stage
 * of the same session then runs on the completed corridor.  Layouts are those of rbp_plan with the session's strides:
 * sfc_box [N][max_boxes][6] f64, sfc_time [N][max_boxes] f64, sfc_count [N] i32, rsfc_normal [npair][M][3] f32, rsfc_time [M] f64. */
typedef struct rbp_device_arrays {
    void* sfc_count;
    void* sfc_box;
    void* sfc_time;
    void* rsfc_normal;
    void* rsfc_time;
    void* status;                   /* [1] i32: the mission's first error so far (0 = ok), written by the kernels */
    int32_t N, M, max_boxes, npair; /* M, max_boxes: the strides of the arrays (the session's maxima) */
    int32_t device;
} rbp_device_arrays;
int rbp_session_device_arrays(rbp_session* s, int32_t mission, rbp_device_arrays* out);

/* work counters of the last `run`, for the roofline report: see DESIGN.md */
typedef struct rbp_counters {
    double sfc_samples;     /* getDistance-equivalent samples tested by the SFC kernel (summed over missions) */
    double qp_flops;        /* flops of the dense block factorisations/solves the QP kernel executed (grid-wide joint solver: the
                               ALGORITHMIC figure of the mission -- tile sweeps + substitutions -- whether or not panel rows are re-formed
                               per tile, and on each rank of a pair that shares the solve) */
    double qp_ipm_iters;    /* interior-point iterations summed over QPs */
    double qp_solves;       /* number of batch QPs solved */
    double qp_constraint_rows; /* inequality rows swept (rows x passes) */
    double qp_polished;     /* batch QPs whose active-set polish was accepted (the rest keep the interior-point answer) */
    double qp_row_bytes;    /* algorithmic HBM bytes the QP kernel moves (row state and constants per sweep, knot blocks per
                               factorisation / substitution): the numerator of the HBM-side roofline, DESIGN.md 3.3 */
    double kkt_max;         /* max of rbp_plan::kkt_max over the missions */
} rbp_counters;
int rbp_session_counters(rbp_session* s, rbp_counters* out, void* stream);

/* bytes of QP workspace one mission of this session occupies (0 until it has been reserved) */
size_t rbp_session_workspace_bytes(rbp_session* s);
/* reserve (and clear) the QP workspace for the solver options in force now, instead of leaving it to the first PLANNER run: a caller that
 * times its first plan, or wants the allocation failure before it has uploaded anything else */
int rbp_session_reserve_workspace(rbp_session* s, void* stream);

/* raw per-mission diagnostic scalars of the last run: out[K][n], n <= 36 = SC_N (layout: kernels/rbp_dev.h SC_*) */
int rbp_session_scalars(rbp_session* s, double* out, int n, void* stream);

/* ---- contexts: device memory kept across calls -------------------------------------------------
 * "device selection passed through an opaque context handle created once".  A context owns one device
 * arena that the synchronous calls below (and sessions created in it) reuse, so a caller that plans repeatedly -- the
 * reference's ROS node, one Corridor::update + RBPPlanner::update per plan -- pays hipMalloc/hipFree once, not per call.
 * rbp_corridor_update / rbp_planner_update (no context argument) use a per-thread default context on the calling thread's
 * current device.  device < 0 = the calling thread's current device.  One live session per context at a time. */
typedef struct rbp_ctx rbp_ctx;
int rbp_ctx_create(rbp_ctx** out, int device);
/* options of every later session / one-shot call in this context; ctx == NULL: the calling thread's default context (created on the
 * calling thread's current device if it does not exist yet) */
int rbp_ctx_set_solver_opts(rbp_ctx* ctx, const rbp_solver_opts* o);
void rbp_ctx_destroy(rbp_ctx* ctx);
int rbp_ctx_corridor_update(rbp_ctx* ctx, const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan);
int rbp_ctx_planner_update(rbp_ctx* ctx, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan);
/* Corridor::update && RBPPlanner::update in one call (one upload, one download): what swarm_traj_planner_rbp.cpp:96-116 does
 * back to back.  Returns the first failing stage's code. */
int rbp_ctx_plan_update(rbp_ctx* ctx, const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan);
/* a batched session whose device memory is the context's arena (released back to it by rbp_session_destroy) */
int rbp_session_create_in(rbp_ctx* ctx, rbp_session** out, int K, const rbp_world* worlds, const rbp_mission* missions,
                          const rbp_param* param, const rbp_plan* plans);

/* the calling thread's default context (see above) is released when the thread exits; a long-lived thread that is done planning
 * can give the arena back earlier */
void rbp_release_thread_context(void);

/* library/version/diagnostics.  RBP_ABI_VERSION changes whenever a struct of this header changes its layout; a binding built
 * against another header must refuse to run (rbp_plan / rbp_counters are written by the library).  rbp_sizeof lets a binding
 * that cannot see this header (ctypes, cgo) compare its own struct sizes with the library's. */
#define RBP_ABI_VERSION 6  /* 1: round 1; 2: rbp_plan.qp_solves/qp_unpolished/kkt_max, rbp_counters.qp_row_bytes/kkt_max; 3: contexts, device views;
                              4: rbp_solver_opts (the library no longer reads environment variables); 5: rbp_session_shard_joint, RBP_ERR_EXCHANGE;
                              6: rbp_param.timescale_rule, rbp_plan.time_scale_alt */
enum { RBP_SIZEOF_WORLD = 0, RBP_SIZEOF_MISSION = 1, RBP_SIZEOF_PARAM = 2, RBP_SIZEOF_PLAN = 3, RBP_SIZEOF_COUNTERS = 4, RBP_SIZEOF_DEVICE_ARRAYS = 5,
       RBP_SIZEOF_SOLVER_OPTS = 6 };
int rbp_abi_version(void);
size_t rbp_sizeof(int which);
const char* rbp_version(void);
/* ---- distance grid of a world on the GPU ---------------------------------
 * What  DynamicEDTOctomap distmap(max_dist, tree, bbx_min, bbx_max, false); distmap.update();  followed by getDistance() on every
 * voxel centre of the box yields (swarm_traj_planner_rbp_test_all.cpp:57-63, swarm_traj_planner_rbp.cpp:73-80): the float grid
 * rbp_world.dist points at, [nx][ny][nz] with z fastest, clamped at ((int)(max_dist / res + 1)) cells like dynamicEDT3D.
 * leaf_keys: [n_leaves][4] = min-corner voxel key minus 32768 (x, y, z) and edge length in voxels of every occupied leaf (what
 * rbp_octomap_load_bt of rbp_host.h returns).  rbp_edt_dims gives the grid shape (dim, key_min as in rbp_world) for a box;
 * rbp_edt_build fills dist (host buffer of dim[0]*dim[1]*dim[2] floats).  Bit-identical to the host library's rbp_world_build. */
int rbp_edt_dims(double res, const double bbx_min[3], const double bbx_max[3], int32_t dim[3], int32_t key_min[3]);
int rbp_edt_build(const int32_t* leaf_keys, int64_t n_leaves, double res, const double bbx_min[3], const double bbx_max[3],
                  double max_dist, float* dist);

const char* rbp_last_error(void);
int rbp_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RBP_H */
