/* rbp_host.h — host-side front-end (the callers / data formats either side of the hot path).
 *
 * Not accelerated; exists so that mission JSONs and octomap .bt worlds can be fed to rbp.h without ROS,
 * octomap, dynamicEDT3D or Boost:
 *   Mission::setMission            swarm_planner/include/mission.hpp:22-88
 *   octomap::OcTree(path) + DynamicEDTOctomap(1, tree, min, max, false).update()
 *                                  swarm_planner/src/swarm_traj_planner_rbp_test_all.cpp:51-63
 *   ECBSPlanner::update            swarm_planner/include/ecbs_planner.hpp:21-136
 * Plain C ABI; buffers returned by the _load/_build calls are owned by the library and freed by the matching _free.
 */
#ifndef RBP_HOST_H
#define RBP_HOST_H

#include "rbp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- mission JSON ({"quadrotors": {...}, "agents": [...]}), mission.hpp:22-88 ---------------- */
typedef struct rbp_mission_buf {
    int32_t N;
    double* start;   /* [N][9] */
    double* goal;    /* [N][9] */
    double* radius;  /* [N] */
    double* speed;   /* [N]  quad_speed (parsed, unused by the path) */
    double* max_vel; /* [N][3] */
    double* max_acc; /* [N][3] */
} rbp_mission_buf;
int rbp_mission_load_json(const char* path, rbp_mission_buf* out);
void rbp_mission_free(rbp_mission_buf* m);

/* ---- octomap .bt reader: occupied leaves (a leaf above depth 16 is a cube of `size` voxels per edge) */
typedef struct rbp_octomap_buf {
    double res;
    int64_t n_occupied;   /* occupied leaves */
    int32_t* keys;        /* [n_occupied][4] = min-corner voxel key minus 32768 (x,y,z), edge length in voxels */
    int64_t n_nodes;      /* node count as parsed (equals the header's `size`) */
} rbp_octomap_buf;
int rbp_octomap_load_bt(const char* path, rbp_octomap_buf* out);
void rbp_octomap_free(rbp_octomap_buf* m);

/* ---- distance map over the world bounding box (exact EDT in cells, clamped like dynamicEDT3D) -- */
typedef struct rbp_world_buf {
    int32_t dim[3];
    int32_t key_min[3];
    double res;
    float* dist;          /* [nx][ny][nz] metres */
} rbp_world_buf;
/* bbx_min/bbx_max are converted to float first (octomap::point3d), as the reference does. max_dist in metres. */
int rbp_world_build(const rbp_octomap_buf* map, const double bbx_min[3], const double bbx_max[3], double max_dist,
                    rbp_world_buf* out);
void rbp_world_free(rbp_world_buf* w);

/* ---- ECBS front-end: initTraj [N][M+1][3] float32 and T[M+1] (ecbs_planner.hpp:34-70) --------- */
typedef struct rbp_init_traj_buf {
    int32_t N, M;
    double* T;            /* [M+1] */
    float* init_traj;     /* [N][M+1][3] */
    int32_t makespan;     /* max path cost */
    int32_t sum_cost;
    int64_t high_level_expanded;
    int64_t low_level_expanded;
} rbp_init_traj_buf;
/* returns 0 ok; 1 start/goal occluded; 2 search failed / node budget exhausted */
int rbp_ecbs_plan(const rbp_world_buf* world, const rbp_mission* mission, const rbp_param* param,
                  int64_t max_high_level_nodes, rbp_init_traj_buf* out);
void rbp_init_traj_free(rbp_init_traj_buf* t);

/* ---- validation metrics of rbp_publisher.hpp:685-695, 769-798 ------------------
 * coef as in rbp_plan.coef.  dt = sampling step (reference: 0.1 s). */
int rbp_validate(const rbp_mission* mission, const rbp_param* param, int32_t M, const double* T, const double* coef,
                 double dt, double* min_safety_ratio, double* total_flight_distance);

/* crazyswarm CSV of rbp_planner.hpp:295-324 (one file per agent: <dir>/coef<qi+1>.csv) */
int rbp_write_coef_csv(const char* dir, int32_t N, int32_t M, const double* T, const double* coef);

/* The QP of batch `l` as a CPLEX LP-format file: what cplex.exportModel(".../log/QPmodel.lp") writes when the reference runs with
 * log = true (rbp_planner.hpp:150-152): variables named and ordered as in populatebyrow (:552-577), objective without 1/2,
 * equality, SFC and RSFC rows (:582-684).  plan: T / init_traj / corridor as BEFORE timeScale.  dummy: the control points frozen
 * agents are held at ([N][3][6M], the layout of rbp_plan.ctrl), or NULL for build_dummy of the initial trajectory (:513-549). */
int rbp_write_qp_lp(const char* path, const rbp_mission* mission, const rbp_param* param, const rbp_plan* plan, int32_t l,
                    const double* dummy);

#ifdef __cplusplus
}
#endif
#endif
