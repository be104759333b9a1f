/* flatfile.h -- the flat binary case file tests/abi_c/flatio.py writes (test infrastructure; plain C so that smoke.c and the adapter
 * harness read the same bytes).  Layout, all little endian, no padding between items:
 *   int32  magic 0x46504252 ("RBPF"), N, M, max_boxes, dim[3], key_min[3]        (10 x int32)
 *   double res, param[22]  (the fields of rbp_param in header order, integers as doubles)
 *   float  dist[dim0*dim1*dim2]
 *   double start[N*9], goal[N*9], radius[N], max_vel[N*3], max_acc[N*3], T[M+1]
 *   float  init_traj[N*(M+1)*3]
 * Result file (written by the programs, read back by flatio.py):
 *   int32  magic, rc_corridor, rc_planner, N, M, max_boxes, qp_solves, qp_unpolished
 *   int32  sfc_count[N];  double sfc_box[N*MB*6], sfc_time[N*MB];  float rsfc_normal[npair*M*3];  double rsfc_time[M]
 *   double coef[N*3*6M], ctrl[N*3*6M], T[M+1], time_scale, total_cost
 */
#ifndef RBP_TEST_FLATFILE_H
#define RBP_TEST_FLATFILE_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define FLAT_MAGIC 0x46504252

typedef struct flat_case {
    int32_t N, M, max_boxes, dim[3], key_min[3];
    double res, param[22];
    float* dist;
    double *start, *goal, *radius, *max_vel, *max_acc, *T;
    float* init_traj;
} flat_case;

static void* flat_read_array(FILE* f, size_t count, size_t elem) {
    void* p = malloc(count > 0 ? count * elem : 1);
    if (!p || fread(p, elem, count, f) != count) {
        fprintf(stderr, "flatfile: short read\n");
        exit(90);
    }
    return p;
}

static int flat_case_read(const char* path, flat_case* c) {
    FILE* f = fopen(path, "rb");
    int32_t head[10];
    size_t cells;
    if (!f) return -1;
    if (fread(head, sizeof(int32_t), 10, f) != 10 || head[0] != FLAT_MAGIC) {
        fclose(f);
        return -2;
    }
    c->N = head[1], c->M = head[2], c->max_boxes = head[3];
    c->dim[0] = head[4], c->dim[1] = head[5], c->dim[2] = head[6];
    c->key_min[0] = head[7], c->key_min[1] = head[8], c->key_min[2] = head[9];
    if (fread(&c->res, sizeof(double), 1, f) != 1 || fread(c->param, sizeof(double), 22, f) != 22) {
        fclose(f);
        return -3;
    }
    cells = (size_t)c->dim[0] * (size_t)c->dim[1] * (size_t)c->dim[2];
    c->dist = (float*)flat_read_array(f, cells, sizeof(float));
    c->start = (double*)flat_read_array(f, (size_t)c->N * 9, sizeof(double));
    c->goal = (double*)flat_read_array(f, (size_t)c->N * 9, sizeof(double));
    c->radius = (double*)flat_read_array(f, (size_t)c->N, sizeof(double));
    c->max_vel = (double*)flat_read_array(f, (size_t)c->N * 3, sizeof(double));
    c->max_acc = (double*)flat_read_array(f, (size_t)c->N * 3, sizeof(double));
    c->T = (double*)flat_read_array(f, (size_t)c->M + 1, sizeof(double));
    c->init_traj = (float*)flat_read_array(f, (size_t)c->N * ((size_t)c->M + 1) * 3, sizeof(float));
    fclose(f);
    return 0;
}

static void flat_case_free(flat_case* c) {
    free(c->dist), free(c->start), free(c->goal), free(c->radius), free(c->max_vel), free(c->max_acc), free(c->T), free(c->init_traj);
}
#endif
