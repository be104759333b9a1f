/* smoke.c -- plain C11 caller of include/rbp.h (test infrastructure): proves that the header is C (not only C++ or
 * ctypes), that the structs can be filled by designated initialisers in a C translation unit, and that the two stage calls the
 * reference's call sites make (swarm_planner/src/swarm_traj_planner_rbp.cpp:96-116: Corridor::update then RBPPlanner::update) work from
 * C with host buffers only.  usage: smoke <case.flat> <result.flat>; exit 0 = both calls returned (their codes are in the result
 * file), 30 = RBP_ERR_NO_DEVICE from the first call (the product has no CPU fallback).  Compiled with -std=c11 -pedantic -Wall -Werror. */
#include <rbp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flatfile.h"

_Static_assert(sizeof(rbp_world) == 6 * sizeof(int32_t) + sizeof(double) + sizeof(void*), "rbp_world has no padding surprises");
_Static_assert(RBP_ABI_VERSION >= 5, "header ABI");

int main(int argc, char** argv) {
    flat_case c;
    if (argc < 3 || flat_case_read(argv[1], &c) != 0) {
        fprintf(stderr, "usage: smoke case.flat result.flat\n");
        return 91;
    }
    if (rbp_abi_version() != RBP_ABI_VERSION || rbp_sizeof(RBP_SIZEOF_PLAN) != sizeof(rbp_plan) ||
        rbp_sizeof(RBP_SIZEOF_PARAM) != sizeof(rbp_param) || rbp_sizeof(RBP_SIZEOF_MISSION) != sizeof(rbp_mission) ||
        rbp_sizeof(RBP_SIZEOF_WORLD) != sizeof(rbp_world) || rbp_sizeof(RBP_SIZEOF_SOLVER_OPTS) != sizeof(rbp_solver_opts)) {
        fprintf(stderr, "smoke: the library was built from another rbp.h\n");
        return 92;
    }
    const int N = c.N, M = c.M, MB = c.max_boxes;
    const size_t npair = (size_t)N * (size_t)(N - 1) / 2, ncoef = (size_t)N * 3 * 6 * (size_t)M;
    int32_t* sfc_count = calloc((size_t)N, sizeof(int32_t));
    double* sfc_box = calloc((size_t)N * (size_t)MB * 6, sizeof(double));
    double* sfc_time = calloc((size_t)N * (size_t)MB, sizeof(double));
    float* normals = calloc(npair * (size_t)M * 3 + 1, sizeof(float));
    double* rsfc_time = calloc((size_t)M, sizeof(double));
    double* coef = calloc(ncoef, sizeof(double));
    double* ctrl = calloc(ncoef, sizeof(double));

    rbp_world w = {.dim = {c.dim[0], c.dim[1], c.dim[2]}, .key_min = {c.key_min[0], c.key_min[1], c.key_min[2]}, .res = c.res, .dist = c.dist};
    rbp_mission m = {.N = N, .start = c.start, .goal = c.goal, .radius = c.radius, .max_vel = c.max_vel, .max_acc = c.max_acc};
    rbp_param p;
    rbp_param_defaults(&p);
    {
        const double* v = c.param;
        p.world_min[0] = v[0], p.world_min[1] = v[1], p.world_min[2] = v[2];
        p.world_max[0] = v[3], p.world_max[1] = v[4], p.world_max[2] = v[5];
        p.box_xy_res = v[6], p.box_z_res = v[7], p.downwash = v[8], p.time_step = v[9];
        p.ecbs_w = v[10], p.grid_xy_res = v[11], p.grid_z_res = v[12], p.grid_margin = v[13];
        p.n = (int32_t)v[14], p.phi = (int32_t)v[15], p.sequential = (int32_t)v[16], p.batch_size = (int32_t)v[17];
        p.batch_iter = (int32_t)v[18], p.iteration = (int32_t)v[19], p.time_scale = (int32_t)v[20], p.log = (int32_t)v[21];
    }
    rbp_plan pl = {.N = N, .M = M, .T = c.T, .init_traj = c.init_traj, .max_boxes = MB, .sfc_count = sfc_count, .sfc_box = sfc_box,
                   .sfc_time = sfc_time, .rsfc_normal = normals, .rsfc_time = rsfc_time, .coef = coef, .ctrl = ctrl};

    int rc_c = rbp_corridor_update(&w, &m, &p, &pl);
    if (rc_c == RBP_ERR_NO_DEVICE) {
        fprintf(stderr, "smoke: %s\n", rbp_last_error());
        return RBP_ERR_NO_DEVICE;
    }
    int rc_p = rc_c == RBP_OK ? rbp_planner_update(&m, &p, &pl) : -1;
    if (rc_c != RBP_OK || rc_p != RBP_OK) fprintf(stderr, "smoke: corridor %d planner %d: %s\n", rc_c, rc_p, rbp_last_error());

    FILE* f = fopen(argv[2], "wb");
    if (!f) return 93;
    int32_t head[8] = {FLAT_MAGIC, rc_c, rc_p, N, M, MB, pl.qp_solves, pl.qp_unpolished};
    fwrite(head, sizeof(int32_t), 8, f);
    fwrite(sfc_count, sizeof(int32_t), (size_t)N, f);
    fwrite(sfc_box, sizeof(double), (size_t)N * (size_t)MB * 6, f);
    fwrite(sfc_time, sizeof(double), (size_t)N * (size_t)MB, f);
    fwrite(normals, sizeof(float), npair * (size_t)M * 3, f);
    fwrite(rsfc_time, sizeof(double), (size_t)M, f);
    fwrite(coef, sizeof(double), ncoef, f);
    fwrite(ctrl, sizeof(double), ncoef, f);
    fwrite(c.T, sizeof(double), (size_t)M + 1, f);
    fwrite(&pl.time_scale, sizeof(double), 1, f);
    fwrite(&pl.total_cost, sizeof(double), 1, f);
    fclose(f);
    rbp_release_thread_context();
    free(sfc_count), free(sfc_box), free(sfc_time), free(normals), free(rsfc_time), free(coef), free(ctrl);
    flat_case_free(&c);
    return 0;
}
