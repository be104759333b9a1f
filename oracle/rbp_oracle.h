/* rbp_oracle.h — CPU oracle for the RBP plan path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (swarm_simulator_amd/) never does.  PARITY UNPINNED: the reference ships no golden vectors or
 * tests for this path and cannot be built here (CPLEX, Eigen, octomap, dynamicEDT3D, ROS absent —
 * SURVEY.md 8c); what pins the oracle instead is listed in oracle/README.md.
 *
 * The functions take the same flat structs as the product ABI (include/rbp.h) so a test can hand
 * identical buffers to both sides.
 */
#ifndef RBP_ORACLE_H
#define RBP_ORACLE_H

#include "../include/rbp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Corridor::update (rbp_corridor.hpp:21-26).  n_samples (may be NULL) receives the number of
 * getDistance() calls issued (rbp_corridor.hpp:66) = the SFC stage's algorithmic work unit. */
int oracle_corridor_update(const rbp_world* world, const rbp_mission* mission, const rbp_param* param,
                           rbp_plan* plan, int64_t* n_samples);

/* single pieces, for unit tests */
int oracle_is_obstacle_in_box(const rbp_world* world, const rbp_param* param, const double box[6], double margin);
void oracle_expand_box(const rbp_world* world, const rbp_param* param, double box[6], double margin);
int oracle_rsfc_normal(const float pi0[3], const float pi1[3], const float pj0[3], const float pj1[3], double downwash,
                       float out[3]);

typedef struct oracle_qp_options {
    int32_t linear_solver; /* 0 = null-space block-tridiagonal Cholesky (default), 1 = dense LU of the full KKT matrix */
    int32_t max_iter;      /* default 60 */
    double tol_feas;       /* max-norm of primal residuals [m]; default 1e-9 */
    double tol_gap;        /* complementarity measure mu = s'z / rows; default 1e-10 */
    int32_t verbose;
    int32_t polish;        /* 1 (default): active-set polish of the interior-point answer, accepted only if it passes
                              the full KKT check */
} oracle_qp_options;
void oracle_qp_default_options(oracle_qp_options* o);

typedef struct oracle_qp_report {
    int32_t n_qp;          /* batch QPs solved */
    int32_t iters_total;
    int32_t iters_max;
    int32_t n_polished;    /* QPs whose polished solution was accepted */
    int32_t n_loose;       /* QPs accepted with a primal residual between tol_feas and 1e-6 (inconsistent rows) */
    double kkt_stationarity; /* worst over QPs: || 2Qx + A'y + G'z ||_inf / (1 + ||2Qx||_inf) */
    double kkt_primal_eq;    /* || Ax - b ||_inf */
    double kkt_primal_ineq;  /* max (Gx - h)_+ */
    double kkt_dual_min;     /* min z (>= 0) */
    double kkt_compl;        /* max z_i * (h - Gx)_i */
    double duality_gap_rel;  /* worst (primal - dual objective) / max(1,|primal|) */
    double flops;            /* dense factor/solve flops executed (for the CPU baseline report) */
} oracle_qp_report;

/* RBPPlanner::update (rbp_planner.hpp:33-84) with an own interior-point method in place of CPLEX. */
int oracle_planner_update(const rbp_mission* mission, const rbp_param* param, rbp_plan* plan,
                          const oracle_qp_options* opt, oracle_qp_report* report);

/* constant / structural matrices, exactly as the reference builds them (for unit tests) */
void oracle_build_Q_base(double Q_base[36], double basis[36]);                 /* rbp_planner.hpp:327-347 */
void oracle_build_Aeq_base(int M, const double* T, double* Aeq /* [3(M+1)][6M] row-major */); /* :353-405 */
void oracle_build_dummy(int N, int M, const float* init_traj, double* dummy /* [N][3][6M] */);   /* :513-549 */

/* Objective (sum over agents/dims/segments of c' Q_p c, no 1/2; rbp_planner.hpp:582-605) of control points
 * ctrl[N][3][6M], and the worst violations of the reference's constraint sets for them:
 *   eq   : | Aeq c - deq |            rbp_planner.hpp:608-622
 *   box  : SFC bound violation         rbp_planner.hpp:626-635
 *   rsfc : (r_i+r_j) - n.(c_j-c_i)     rbp_planner.hpp:638-684 (all pairs)
 * Independent of any solver: used to judge ANY candidate solution (GPU or oracle). */
int oracle_evaluate_ctrl(const rbp_mission* mission, const rbp_plan* plan, const double* ctrl, double* objective,
                         double* viol_eq, double* viol_box, double* viol_rsfc);

/* Bernstein control points -> monomial coef (rbp_planner.hpp:170-196) */
void oracle_ctrl_to_coef(int N, int M, const double* T, const double* ctrl, double* coef);

/* timeScale (rbp_planner.hpp:209-266): returns time_scale and applies it to coef/T/sfc_time/rsfc_time */
double oracle_time_scale(const rbp_mission* mission, rbp_plan* plan);   /* rule RBP_TIMESCALE_ALL_REAL_ROOTS */
/* ... under either rule of rbp_param.timescale_rule; plan->time_scale_alt receives the OTHER rule's factor */
double oracle_time_scale_rule(const rbp_mission* mission, rbp_plan* plan, int rule);
/* eigenvalues of the companion matrix of c[0] t^deg + ... + c[deg] (deg <= 3, c[0] != 0) in the order of Eigen 3.3's EigenSolver as
 * restated in oracle/planner.c (roots_derivative, rbp_planner.hpp:737-751); returns deg, 0 = no convergence, -1 = bad argument */
int oracle_companion_eigenvalues(const double* c, int deg, double* re, double* im);

#ifdef __cplusplus
}
#endif
#endif
