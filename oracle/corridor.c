/* corridor.c — CPU restatement of Corridor::update.  TEST INFRASTRUCTURE ONLY (see rbp_oracle.h).
 *
 * Follows swarm_planner/include/rbp_corridor.hpp line by line in behaviour:
 *   isObstacleInBox :44-78     isBoxInBoundary :80-87     isPointInBox :89-97
 *   expand_box      :99-147    updateObsBox    :149-243   updateRelBox  :338-398
 * Third-party semantics restated from their published sources (absent from /root/reference):
 *   DynamicEDTOctomap::getDistance  — voxel key = floor((1/res) * (double)(float)coord); -1 outside the grid
 *   octomath::Vector3 (octomap::point3d) — three float32; operator-,*,/= in float; dot()/norm_sq() evaluate the
 *   float expression and widen to double; norm() = sqrt(double); normalize() divides by (float)norm if > 0.
 * Compile with -ffp-contract=off: the reference is built for generic x86-64 (no FMA contraction).
 */
#include "rbp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SP_EPSILON 1e-9       /* sp_const.hpp:3 */
#define SP_EPSILON_FLOAT 1e-6 /* sp_const.hpp:4 */

typedef struct {
    const rbp_world* w;
    const rbp_param* p;
    int64_t samples;
} ctx_t;

/* DynamicEDTOctomap::getDistance(point3d) */
static float get_distance(ctx_t* c, float x, float y, float z) {
    const rbp_world* w = c->w;
    const double rf = 1.0 / w->res; /* octomap resolution_factor */
    int kx = (int)floor(rf * (double)x) - w->key_min[0];
    int ky = (int)floor(rf * (double)y) - w->key_min[1];
    int kz = (int)floor(rf * (double)z) - w->key_min[2];
    c->samples++;
    if (kx < 0 || ky < 0 || kz < 0 || kx >= w->dim[0] || ky >= w->dim[1] || kz >= w->dim[2]) return -1.0f;
    return w->dist[((size_t)kx * w->dim[1] + ky) * w->dim[2] + kz];
}

/* rbp_corridor.hpp:44-78 */
static int is_obstacle_in_box(ctx_t* c, const double* box, double margin) {
    const rbp_param* p = c->p;
    double x, y, z;
    int count1 = 0;
    for (double i = box[0]; i < box[3] + SP_EPSILON_FLOAT; i += p->box_xy_res) {
        int count2 = 0;
        for (double j = box[1]; j < box[4] + SP_EPSILON_FLOAT; j += p->box_xy_res) {
            int count3 = 0;
            for (double k = box[2]; k < box[5] + SP_EPSILON_FLOAT; k += p->box_z_res) {
                x = i + SP_EPSILON_FLOAT;
                if (count1 == 0 && box[0] > p->world_min[0] + SP_EPSILON_FLOAT) x = box[0] - SP_EPSILON_FLOAT;
                y = j + SP_EPSILON_FLOAT;
                if (count2 == 0 && box[1] > p->world_min[1] + SP_EPSILON_FLOAT) y = box[1] - SP_EPSILON_FLOAT;
                z = k + SP_EPSILON_FLOAT;
                if (count3 == 0 && box[2] > p->world_min[2] + SP_EPSILON_FLOAT) z = box[2] - SP_EPSILON_FLOAT;
                float dist = get_distance(c, (float)x, (float)y, (float)z); /* octomap::point3d is float32 */
                if (dist < margin - SP_EPSILON_FLOAT) return 1;
                count3++;
            }
            count2++;
        }
        count1++;
    }
    return 0;
}

/* rbp_corridor.hpp:80-87 */
static int is_box_in_boundary(const rbp_param* p, const double* box) {
    return box[0] > p->world_min[0] - SP_EPSILON && box[1] > p->world_min[1] - SP_EPSILON &&
           box[2] > p->world_min[2] - SP_EPSILON && box[3] < p->world_max[0] + SP_EPSILON &&
           box[4] < p->world_max[1] + SP_EPSILON && box[5] < p->world_max[2] + SP_EPSILON;
}

/* rbp_corridor.hpp:89-97 (point is float32, promoted) */
static int is_point_in_box(const float* pt, const double* box) {
    return pt[0] > box[0] - SP_EPSILON && pt[1] > box[1] - SP_EPSILON && pt[2] > box[2] - SP_EPSILON &&
           pt[0] < box[3] + SP_EPSILON && pt[1] < box[4] + SP_EPSILON && pt[2] < box[5] + SP_EPSILON;
}

/* rbp_corridor.hpp:99-147 */
static void expand_box(ctx_t* c, double* box, double margin) {
    const rbp_param* p = c->p;
    double box_cand[6], box_update[6];
    int axis_cand[6] = {0, 1, 2, 3, 4, 5};
    int n_cand = 6;
    int i = -1;
    int axis;
    while (n_cand > 0) {
        memcpy(box_cand, box, sizeof(box_cand));
        memcpy(box_update, box, sizeof(box_update));
        while (!is_obstacle_in_box(c, box_update, margin) && is_box_in_boundary(p, box_update)) {
            i++;
            if (i >= n_cand) i = 0;
            axis = axis_cand[i];
            memcpy(box, box_cand, sizeof(box_cand));
            memcpy(box_update, box_cand, sizeof(box_cand));
            if (axis < 3) {
                box_update[axis + 3] = box_cand[axis];
                if (axis == 2)
                    box_cand[axis] = box_cand[axis] - p->box_z_res;
                else
                    box_cand[axis] = box_cand[axis] - p->box_xy_res;
                box_update[axis] = box_cand[axis];
            } else {
                box_update[axis - 3] = box_cand[axis];
                if (axis == 5)
                    box_cand[axis] = box_cand[axis] + p->box_z_res;
                else
                    box_cand[axis] = box_cand[axis] + p->box_xy_res;
                box_update[axis] = box_cand[axis];
            }
        }
        /* axis_cand.erase(begin + i).  NB: if the very first test of a fresh expand_box fails, the reference
         * erases begin()+(-1) (undefined behaviour); cannot happen because updateObsBox tests the seed first. */
        if (i < 0) i = 0;
        for (int k = i; k + 1 < n_cand; ++k) axis_cand[k] = axis_cand[k + 1];
        n_cand--;
        if (i > 0)
            i--;
        else
            i = n_cand - 1;
    }
}

/* rbp_corridor.hpp:149-243 */
static int update_obs_box(ctx_t* c, const rbp_mission* mission, rbp_plan* plan) {
    const rbp_param* p = c->p;
    const int N = plan->N, M = plan->M, P = M + 1;
    const double makespan = plan->T[M]; /* :24 */
    int* box_log = (int*)malloc(sizeof(int) * (size_t)plan->max_boxes * P);
    for (int qi = 0; qi < N; ++qi) {
        const float* traj = plan->init_traj + (size_t)qi * P * 3;
        double* boxes = plan->sfc_box + (size_t)qi * plan->max_boxes * 6;
        double* times = plan->sfc_time + (size_t)qi * plan->max_boxes;
        int nbox = 0;
        double box_prev[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < P - 1; ++i) {
            double x = traj[3 * i], y = traj[3 * i + 1], z = traj[3 * i + 2];
            double xn = traj[3 * i + 3], yn = traj[3 * i + 4], zn = traj[3 * i + 5];
            if (is_point_in_box(traj + 3 * i + 3, box_prev)) continue; /* :169 */
            double box[6];
            box[0] = round(fmin(x, xn) / p->box_xy_res) * p->box_xy_res; /* :174-179 */
            box[1] = round(fmin(y, yn) / p->box_xy_res) * p->box_xy_res;
            box[2] = round(fmin(z, zn) / p->box_z_res) * p->box_z_res;
            box[3] = round(fmax(x, xn) / p->box_xy_res) * p->box_xy_res;
            box[4] = round(fmax(y, yn) / p->box_xy_res) * p->box_xy_res;
            box[5] = round(fmax(z, zn) / p->box_z_res) * p->box_z_res;
            if (is_obstacle_in_box(c, box, mission->radius[qi])) { /* :181-187 */
                free(box_log);
                return RBP_ERR_OBSTACLE_IN_INIT_TRAJ;
            }
            expand_box(c, box, mission->radius[qi]);
            if (nbox >= plan->max_boxes) {
                free(box_log);
                return RBP_ERR_SFC_OVERFLOW;
            }
            memcpy(boxes + 6 * nbox, box, sizeof(box));
            times[nbox] = -1;
            nbox++;
            memcpy(box_prev, box, sizeof(box));
        }
        plan->sfc_count[qi] = nbox;
        /* box time segments :195-237 */
        const int box_max = nbox, path_max = P;
        for (int i = 0; i < box_max; ++i)
            for (int j = 0; j < path_max; ++j) {
                int v = 0;
                if (is_point_in_box(traj + 3 * j, boxes + 6 * i)) v = (j == 0) ? 1 : box_log[i * P + j - 1] + 1;
                box_log[i * P + j] = v;
            }
        int box_iter = 0;
        for (int path_iter = 0; path_iter < path_max; path_iter++) {
            if (box_iter == box_max - 1) {
                if (box_log[box_iter * P + path_iter] > 0)
                    continue;
                else
                    box_iter--;
            }
            /* the reference indexes box_log(box_iter, path_iter) with whatever box_iter/path_iter it has reached;
             * a negative index is undefined behaviour there — guarded: stop the walk. */
            if (box_iter < 0 || path_iter < 0) break;
            if (box_log[box_iter * P + path_iter] > 0 && box_log[(box_iter + 1) * P + path_iter] > 0) {
                int count = 1;
                while (path_iter + count < path_max && box_log[box_iter * P + path_iter + count] > 0 &&
                       box_log[(box_iter + 1) * P + path_iter + count] > 0)
                    count++;
                int obs_index = path_iter + count / 2;
                times[box_iter] = plan->T[obs_index];
                path_iter = path_iter + count / 2;
                box_iter++;
            } else if (box_log[box_iter * P + path_iter] == 0) {
                box_iter--;
                path_iter--;
            }
        }
        if (box_max > 0) times[box_max - 1] = makespan; /* :237 */
    }
    free(box_log);
    return RBP_OK;
}

/* octomath::Vector3 helpers (float32 arithmetic, widened results) */
static double v_dot(const float* a, const float* b) { return (double)(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]); }
static double v_norm(const float* a) { return sqrt((double)(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])); }
static void v_normalize(float* a) {
    double len = v_norm(a);
    if (len > 0) {
        float l = (float)len;
        a[0] /= l, a[1] /= l, a[2] /= l;
    }
}

/* rbp_corridor.hpp:353-390 for one (pair, segment).  returns 0 ok, 1 if the normal has zero length */
int oracle_rsfc_normal(const float* pi0, const float* pi1, const float* pj0, const float* pj1, double downwash,
                       float* out) {
    float a[3], b[3], c[3], n[3], m[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = pj0[k] - pi0[k];
        b[k] = pj1[k] - pi1[k];
    }
    a[2] = (float)((double)a[2] / downwash); /* float / double -> double -> float */
    b[2] = (float)((double)b[2] / downwash);
    if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2]) {
        memcpy(m, a, sizeof(m));
    } else {
        memcpy(m, a, sizeof(m));
        double dist_min = v_norm(a);
        double dist = v_norm(b);
        if (dist_min > dist) {
            memcpy(m, b, sizeof(m));
            dist_min = dist;
        }
        for (int k = 0; k < 3; ++k) n[k] = b[k] - a[k];
        v_normalize(n);
        float adn = (float)v_dot(a, n); /* operator*(float) */
        for (int k = 0; k < 3; ++k) c[k] = a[k] - n[k] * adn;
        dist = v_norm(c);
        float ca[3], cb[3];
        for (int k = 0; k < 3; ++k) {
            ca[k] = c[k] - a[k];
            cb[k] = c[k] - b[k];
        }
        if (v_dot(ca, cb) < 0 && dist_min > dist) memcpy(m, c, sizeof(m));
    }
    v_normalize(m);
    m[2] = (float)((double)m[2] / downwash);
    memcpy(out, m, sizeof(m));
    return v_norm(m) == 0;
}

/* rbp_corridor.hpp:338-398 */
static int update_rel_box(const rbp_param* p, rbp_plan* plan) {
    const int N = plan->N, M = plan->M, P = M + 1;
    size_t pair = 0;
    for (int qi = 0; qi < N; ++qi)
        for (int qj = qi + 1; qj < N; ++qj, ++pair) {
            const float* ti = plan->init_traj + (size_t)qi * P * 3;
            const float* tj = plan->init_traj + (size_t)qj * P * 3;
            for (int iter = 1; iter <= M; ++iter) {
                float* out = plan->rsfc_normal + (pair * M + (iter - 1)) * 3;
                if (oracle_rsfc_normal(ti + 3 * (iter - 1), ti + 3 * iter, tj + 3 * (iter - 1), tj + 3 * iter,
                                       p->downwash, out))
                    return RBP_ERR_INIT_TRAJ_COLLIDE;
            }
        }
    for (int iter = 1; iter <= M; ++iter) plan->rsfc_time[iter - 1] = plan->T[iter]; /* :390 */
    return RBP_OK;
}

int oracle_corridor_update(const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan,
                           int64_t* n_samples) {
    if (!world || !mission || !param || !plan || mission->N != plan->N) return RBP_ERR_BAD_ARGUMENT;
    ctx_t c = {world, param, 0};
    int rc = update_obs_box(&c, mission, plan); /* :25  updateObsBox() && updateRelBox() */
    if (n_samples) *n_samples = c.samples;
    if (rc) return rc;
    return update_rel_box(param, plan);
}

int oracle_is_obstacle_in_box(const rbp_world* world, const rbp_param* param, const double box[6], double margin) {
    ctx_t c = {world, param, 0};
    return is_obstacle_in_box(&c, box, margin);
}

void oracle_expand_box(const rbp_world* world, const rbp_param* param, double box[6], double margin) {
    ctx_t c = {world, param, 0};
    expand_box(&c, box, margin);
}
