/* planner.c — CPU restatement of RBPPlanner::update.  TEST INFRASTRUCTURE ONLY (see rbp_oracle.h).
 *
 * Follows swarm_planner/include/rbp_planner.hpp:
 *   setBatch :849-872, isQuadInBatch :874-881, build_Q_base :327-347, build_Q_p :349-351,
 *   build_Aeq_base :353-405, build_deq :408-432, build_dlq :435-511, build_dummy :513-549,
 *   populatebyrow :551-688 (variable/row order), solveQP :111-206, timeScale :209-266 and helpers :695-847.
 * The one thing that cannot be restated is `cplex.solve()` (:158, IBM CPLEX 12.10, proprietary, absent).
 * Each batch QP is strictly convex on its feasible set (unique optimum), so it is replaced
 * by an own Mehrotra predictor-corrector interior-point method that works in the reference's variable
 * space (Bernstein control points, explicit equality multipliers) and certifies its answer by KKT
 * residuals and a duality gap (oracle_qp_report).  Linear algebra: either a dense LU of the full KKT
 * matrix (small cases, obviously correct) or a null-space (continuity-eliminating) block-tridiagonal
 * Cholesky over the knots (same answer, used at mission sizes).
 */
#include "rbp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NCTRL 6 /* n + 1 */

/* ------------------------------------------------------------------------------------------------
 * constant matrices
 * ---------------------------------------------------------------------------------------------- */
void oracle_build_Q_base(double Q[36], double basis[36]) { /* rbp_planner.hpp:330-343 */
    static const double q[36] = {720,  -1800, 1200,  0,     0,     -120, -1800, 4800,  -3600, 0,     600,   0,
                                 1200, -3600, 3600,  -1200, 0,     0,    0,     0,     -1200, 3600,  -3600, 1200,
                                 0,    600,   0,     -3600, 4800,  -1800, -120, 0,     0,     1200,  -1800, 720};
    static const double b[36] = {-1, 5,   -10, 10, -5, 1, 5,  -20, 30, -20, 5, 0, -10, 30, -30, 10, 0, 0,
                                 10, -20, 10,  0,  0,  0, -5, 5,   0,  0,   0, 0, 1,   0,  0,   0,  0, 0};
    if (Q) memcpy(Q, q, sizeof(q));
    if (basis) memcpy(basis, b, sizeof(b));
}

void oracle_build_Aeq_base(int M, const double* T, double* Aeq) { /* rbp_planner.hpp:353-405 */
    static const double A_0[3][6] = {{1, 0, 0, 0, 0, 0}, {-1, 1, 0, 0, 0, 0}, {1, -2, 1, 0, 0, 0}};
    static const double A_T[3][6] = {{0, 0, 0, 0, 0, 1}, {0, 0, 0, 0, -1, 1}, {0, 0, 0, 1, -2, 1}};
    const int n = 5, phi = 3, cols = 6 * M, rows = 2 * phi + (M - 1) * phi;
    memset(Aeq, 0, sizeof(double) * rows * cols);
    int nn = 1;
    for (int i = 0; i < phi; ++i) { /* A_waypoints :380-387 */
        double s0 = pow(T[1] - T[0], -i) * nn, sT = pow(T[M] - T[M - 1], -i) * nn;
        for (int c = 0; c < 6; ++c) {
            Aeq[i * cols + c] = s0 * A_0[i][c];
            Aeq[(phi + i) * cols + 6 * (M - 1) + c] = sT * A_T[i][c];
        }
        nn = nn * (n - i);
    }
    for (int m = 1; m < M; ++m) { /* A_cont :390-399 */
        nn = 1;
        for (int j = 0; j < phi; ++j) {
            double sl = pow(T[m] - T[m - 1], -j) * nn, sr = -pow(T[m + 1] - T[m], -j) * nn;
            int r = 2 * phi + phi * (m - 1) + j;
            for (int c = 0; c < 6; ++c) {
                Aeq[r * cols + 6 * (m - 1) + c] = sl * A_T[j][c];
                Aeq[r * cols + 6 * m + c] = sr * A_0[j][c];
            }
            nn = nn * (n - j);
        }
    }
}

/* dummy layout here: [N][3][6M] (agent, dim, m*6+i) */
void oracle_build_dummy(int N, int M, const float* init_traj, double* dummy) { /* rbp_planner.hpp:513-549 */
    const int P = M + 1;
    for (int qi = 0; qi < N; ++qi) {
        const float* tr = init_traj + (size_t)qi * P * 3;
        int m = 0, idx = 0;
        while (m < M) {
            if (idx >= P - 1) {
                idx = P - 1;
                for (int j = 0; j < 6; ++j)
                    for (int k = 0; k < 3; ++k) dummy[((size_t)qi * 3 + k) * 6 * M + m * 6 + j] = tr[3 * idx + k];
                m++;
            } else {
                for (int j = 0; j < 6; ++j) {
                    int a = (j < 3) ? 0 : 1;
                    for (int k = 0; k < 3; ++k)
                        dummy[((size_t)qi * 3 + k) * 6 * M + m * 6 + j] =
                            (1 - a) * (double)tr[3 * idx + k] + a * (double)tr[3 * (idx + 1) + k];
                }
                m++;
            }
            idx++;
        }
    }
}

/* box per (agent, segment): first SFC box whose end time >= T[m+1]   rbp_planner.hpp:447-453 */
static void select_boxes(const rbp_plan* plan, int qi, int* sel) {
    int bi = 0;
    const int nb = plan->sfc_count[qi];
    for (int m = 0; m < plan->M; ++m) {
        while (bi < nb && plan->sfc_time[(size_t)qi * plan->max_boxes + bi] < plan->T[m + 1]) bi++;
        sel[m] = (bi < nb) ? bi : nb - 1; /* reference would read past the end (UB) */
    }
}
static int select_rsfc(const rbp_plan* plan, int m) { /* rbp_planner.hpp:485-489 */
    int ri = 0;
    while (ri < plan->M && plan->rsfc_time[ri] < plan->T[m + 1]) ri++;
    return ri < plan->M ? ri : plan->M - 1;
}
static size_t pair_index(int N, int qi, int qj) { return (size_t)qi * N - (size_t)qi * (qi + 1) / 2 + (qj - qi - 1); }

/* ------------------------------------------------------------------------------------------------
 * QP in the reference's variable order (populatebyrow :551-688), inequalities as G x <= h
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int nb, M, nx, ne, nc, ne_base;
    int nc_full;    /* rows as the reference counts them (count_lq), before presolve */
    int infeasible; /* a pinned row is violated */
    double* Qseg; /* [M][36]  Q_p(m) = Q_base * dt^-5 (objective is x'Qx, no 1/2) */
    double* Aeq;  /* [ne_base][6M] shared by every (dim, agent) */
    double* beq;  /* [ne] order (k, bi, r) */
    int* gcol;    /* [nc][6] */
    double* gval; /* [nc][6] */
    int* gnnz;    /* [nc] */
    double* h;    /* [nc] */
} qp_t;

static void qp_free(qp_t* q) {
    free(q->Qseg), free(q->Aeq), free(q->beq), free(q->gcol), free(q->gval), free(q->gnnz), free(q->h);
    memset(q, 0, sizeof(*q));
}

static int in_batch(const int* batch, int nb, int qi) { /* isQuadInBatch :874-881 */
    for (int b = 0; b < nb; ++b)
        if (batch[b] == qi) return b;
    return -1;
}

static void qp_build(const rbp_mission* mission, const rbp_plan* plan, const int* batch, int nb, const double* dummy,
                     qp_t* q) {
    const int N = plan->N, M = plan->M, oq = 6 * M; /* offset_quad */
    const int od = nb * oq;                         /* offset_dim */
    memset(q, 0, sizeof(*q));
    q->nb = nb, q->M = M, q->nx = 3 * od, q->ne_base = 3 * (M + 1), q->ne = 3 * nb * q->ne_base;
    double Qb[36];
    oracle_build_Q_base(Qb, NULL);
    q->Qseg = (double*)malloc(sizeof(double) * 36 * M);
    for (int m = 0; m < M; ++m) {
        double s = pow(plan->T[m + 1] - plan->T[m], -2 * 3 + 1); /* build_Q_p :349-351 */
        for (int e = 0; e < 36; ++e) q->Qseg[36 * m + e] = Qb[e] * s;
    }
    q->Aeq = (double*)malloc(sizeof(double) * q->ne_base * oq);
    oracle_build_Aeq_base(M, plan->T, q->Aeq);
    q->beq = (double*)calloc(q->ne, sizeof(double));
    for (int k = 0; k < 3; ++k) /* build_deq :408-432, rows in the order of :608-622 */
        for (int b = 0; b < nb; ++b) {
            int qi = batch[b];
            double* d = q->beq + (size_t)(k * nb + b) * q->ne_base;
            d[0] = mission->start[9 * qi + k], d[1] = mission->start[9 * qi + k + 3], d[2] = mission->start[9 * qi + k + 6];
            d[3] = mission->goal[9 * qi + k], d[4] = mission->goal[9 * qi + k + 3], d[5] = mission->goal[9 * qi + k + 6];
        }
    /* count inequality rows */
    int nrs = 0;
    for (int qi = 0; qi < N; ++qi)
        for (int qj = qi + 1; qj < N; ++qj)
            if (in_batch(batch, nb, qi) >= 0 || in_batch(batch, nb, qj) >= 0) nrs++;
    q->nc = 2 * q->nx + nrs * oq;
    q->gcol = (int*)calloc((size_t)q->nc * 6, sizeof(int));
    q->gval = (double*)calloc((size_t)q->nc * 6, sizeof(double));
    q->gnnz = (int*)calloc(q->nc, sizeof(int));
    q->h = (double*)calloc(q->nc, sizeof(double));
    int row = 0;
    int* sel = (int*)malloc(sizeof(int) * M);
    /* SFC rows :626-635   x <= upper ; -x <= lower(= -box_min)   (dlq_box :443-474) */
    for (int k = 0; k < 3; ++k)
        for (int b = 0; b < nb; ++b) {
            int qi = batch[b];
            select_boxes(plan, qi, sel);
            for (int j = 0; j < oq; ++j) {
                const double* box = plan->sfc_box + ((size_t)qi * plan->max_boxes + sel[j / 6]) * 6;
                int idx = k * od + b * oq + j;
                q->gcol[6 * row] = idx, q->gval[6 * row] = 1.0, q->gnnz[row] = 1, q->h[row] = box[3 + k];
                row++;
                q->gcol[6 * row] = idx, q->gval[6 * row] = -1.0, q->gnnz[row] = 1, q->h[row] = -box[k];
                row++;
            }
        }
    /* RSFC rows :636-684:  n . (p_j - p_i) >= r_i + r_j   ->   -n.(p_j - p_i) <= -(r_i + r_j) */
    for (int qi = 0; qi < N; ++qi)
        for (int qj = qi + 1; qj < N; ++qj) {
            int bi = in_batch(batch, nb, qi), bj = in_batch(batch, nb, qj);
            if (bi < 0 && bj < 0) continue;
            const float* normals = plan->rsfc_normal + pair_index(N, qi, qj) * M * 3;
            double rr = mission->radius[qi] + mission->radius[qj];
            for (int j = 0; j < oq; ++j) {
                const float* nv = normals + 3 * select_rsfc(plan, j / 6);
                int nz = 0;
                double hh = -rr;
                for (int k = 0; k < 3; ++k) {
                    double nk = (double)nv[k];
                    if (bj >= 0) {
                        q->gcol[6 * row + nz] = k * od + bj * oq + j, q->gval[6 * row + nz] = -nk, nz++;
                    } else
                        hh += nk * dummy[((size_t)qj * 3 + k) * oq + j]; /* -n.(dummy_j - x_i) <= -rr */
                    if (bi >= 0) {
                        q->gcol[6 * row + nz] = k * od + bi * oq + j, q->gval[6 * row + nz] = nk, nz++;
                    } else
                        hh -= nk * dummy[((size_t)qi * 3 + k) * oq + j];
                }
                q->gnnz[row] = nz, q->h[row] = hh;
                row++;
            }
        }
    free(sel);
    /* presolve (what any LP/QP presolve, CPLEX's included, does first): rows whose every column is one of the six
     * end control points per (agent, dim) are constants -- those control points are pinned by the start/goal
     * equalities (rows 0-5 of Aeq_base).  They are checked (tolerance 1e-6 = CPLEX's default feasibility
     * tolerance) and dropped; kept, a zero-slack pinned row leaves the feasible set without interior and an
     * interior-point iteration jams on it. */
    {
        double* xfix = (double*)calloc(q->nx, sizeof(double));
        char* pinned = (char*)calloc(q->nx, 1);
        for (int u = 0; u < 3 * nb; ++u) {
            const double* d = q->beq + (size_t)u * q->ne_base;
            double h0 = plan->T[1] - plan->T[0], hT = plan->T[M] - plan->T[M - 1];
            double* xs = xfix + (size_t)u * oq;
            xs[0] = d[0], xs[1] = xs[0] + h0 * d[1] / 5, xs[2] = 2 * xs[1] - xs[0] + h0 * h0 * d[2] / 20;
            double* xe = xs + 6 * (M - 1);
            xe[5] = d[3], xe[4] = xe[5] - hT * d[4] / 5, xe[3] = 2 * xe[4] - xe[5] + hT * hT * d[5] / 20;
            for (int i = 0; i < 3; ++i) pinned[(size_t)u * oq + i] = pinned[(size_t)u * oq + 6 * (M - 1) + 3 + i] = 1;
        }
        int keep = 0;
        q->infeasible = 0;
        for (int c = 0; c < q->nc; ++c) {
            int all_pinned = 1;
            double gx = 0;
            for (int e = 0; e < q->gnnz[c]; ++e) {
                all_pinned &= pinned[q->gcol[6 * c + e]];
                gx += q->gval[6 * c + e] * xfix[q->gcol[6 * c + e]];
            }
            if (all_pinned) {
                if (gx - q->h[c] > 1e-6) q->infeasible = 1;
                continue;
            }
            if (keep != c) {
                memcpy(q->gcol + 6 * keep, q->gcol + 6 * c, sizeof(int) * 6);
                memcpy(q->gval + 6 * keep, q->gval + 6 * c, sizeof(double) * 6);
                q->gnnz[keep] = q->gnnz[c], q->h[keep] = q->h[c];
            }
            keep++;
        }
        q->nc_full = q->nc, q->nc = keep;
        free(xfix), free(pinned);
    }
}

/* ---- operators ---------------------------------------------------------------------------------- */
static void op_Hx(const qp_t* q, const double* x, double* y) { /* y = 2 Q x */
    const int M = q->M;
    for (int u = 0; u < 3 * q->nb; ++u)
        for (int m = 0; m < M; ++m) {
            const double* Q = q->Qseg + 36 * m;
            const double* xs = x + (size_t)u * 6 * M + 6 * m;
            double* ys = y + (size_t)u * 6 * M + 6 * m;
            for (int i = 0; i < 6; ++i) {
                double s = 0;
                for (int j = 0; j < 6; ++j) s += Q[6 * i + j] * xs[j];
                ys[i] = 2 * s;
            }
        }
}
static void op_Ax(const qp_t* q, const double* x, double* r) {
    const int oq = 6 * q->M;
    for (int u = 0; u < 3 * q->nb; ++u)
        for (int e = 0; e < q->ne_base; ++e) {
            const double* a = q->Aeq + (size_t)e * oq;
            const double* xs = x + (size_t)u * oq;
            double s = 0;
            for (int j = 0; j < oq; ++j) s += a[j] * xs[j];
            r[(size_t)u * q->ne_base + e] = s;
        }
}
static void op_ATy_add(const qp_t* q, const double* y, double* r) {
    const int oq = 6 * q->M;
    for (int u = 0; u < 3 * q->nb; ++u)
        for (int e = 0; e < q->ne_base; ++e) {
            const double* a = q->Aeq + (size_t)e * oq;
            double ye = y[(size_t)u * q->ne_base + e];
            double* rs = r + (size_t)u * oq;
            for (int j = 0; j < oq; ++j) rs[j] += a[j] * ye;
        }
}
static void op_Gx(const qp_t* q, const double* x, double* r) {
    for (int c = 0; c < q->nc; ++c) {
        double s = 0;
        for (int e = 0; e < q->gnnz[c]; ++e) s += q->gval[6 * c + e] * x[q->gcol[6 * c + e]];
        r[c] = s;
    }
}
static void op_GTz_add(const qp_t* q, const double* z, double* r) {
    for (int c = 0; c < q->nc; ++c)
        for (int e = 0; e < q->gnnz[c]; ++e) r[q->gcol[6 * c + e]] += q->gval[6 * c + e] * z[c];
}

/* ---- small dense kernels (row-major) -------------------------------------------------------------- */
static int chol_lower(int n, double* A, int lda) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * lda + j];
        for (int k = 0; k < j; ++k) d -= A[j * lda + k] * A[j * lda + k];
        if (!(d > 0)) return 1;
        d = sqrt(d);
        A[j * lda + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * lda + j];
            for (int k = 0; k < j; ++k) s -= A[i * lda + k] * A[j * lda + k];
            A[i * lda + j] = s / d;
        }
    }
    return 0;
}
/* B (n x m, row-major) <- L^{-1} B */
static void trsm_lower(int n, int m, const double* L, int ldl, double* B, int ldb) {
    for (int i = 0; i < n; ++i) {
        double* bi = B + (size_t)i * ldb;
        for (int k = 0; k < i; ++k) {
            double l = L[i * ldl + k];
            if (l == 0) continue;
            const double* bk = B + (size_t)k * ldb;
            for (int c = 0; c < m; ++c) bi[c] -= l * bk[c];
        }
        double d = 1.0 / L[i * ldl + i];
        for (int c = 0; c < m; ++c) bi[c] *= d;
    }
}
/* B <- L^{-T} B */
static void trsm_lower_t(int n, int m, const double* L, int ldl, double* B, int ldb) {
    for (int i = n - 1; i >= 0; --i) {
        double* bi = B + (size_t)i * ldb;
        double d = 1.0 / L[i * ldl + i];
        for (int c = 0; c < m; ++c) bi[c] *= d;
        for (int k = 0; k < i; ++k) {
            double l = L[i * ldl + k];
            if (l == 0) continue;
            double* bk = B + (size_t)k * ldb;
            for (int c = 0; c < m; ++c) bk[c] -= l * bi[c];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * KKT solver:  [Hbar A'; A 0] [dx; dy] = [r1; r2],  Hbar = 2Q + G' diag(w) G
 *   mode 1: dense LU with partial pivoting of the full matrix (small cases; the independent check)
 *   mode 0: null-space method.  The equalities are C2 continuity + end states, so null(A) has the explicit
 *           local basis F: per (agent, dim) and interior knot j the three control points right of the knot
 *           are free (u_j) and the three left of it follow from continuity (L_j u_j).  F' Hbar F is block
 *           tridiagonal over knots (block = 9 * batch agents) and positive definite without any help from
 *           the barrier, so a plain block Cholesky stays benign as mu -> 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const qp_t* q;
    int mode;
    double flops;
    const double* w;
    /* dense */
    double* K;
    int* piv;
    /* reduced */
    int nk;      /* block size 9*nb */
    double* Lk;  /* [M-1][9]  left-control-point map of knot j: c_{j-1,3..5} = Lk[j] u_j */
    double* AAt; /* Cholesky of A A' (ne_base x ne_base) */
    double* Td;  /* [M-1][nk*nk] diagonal blocks, then their Cholesky factors */
    double* To;  /* [M-2][nk*nk] T_{j+1,j} then L_{j+1,j} */
    double *t1, *t2, *gw, *ru, *xp;
} kkt_t;

static void kkt_init(kkt_t* k, const qp_t* q, int mode) {
    memset(k, 0, sizeof(*k));
    k->q = q, k->mode = mode;
    const int M = q->M;
    if (mode == 1) {
        size_t n = (size_t)q->nx + q->ne;
        k->K = (double*)malloc(sizeof(double) * n * n);
        k->piv = (int*)malloc(sizeof(int) * n);
        return;
    }
    const int nk = k->nk = 9 * q->nb, neb = q->ne_base, oq = 6 * M;
    k->Lk = (double*)calloc((size_t)(M + 1) * 9, sizeof(double));
    /* continuity rows of Aeq_base at knot j (rows 6+3(j-1)+i): sl_i * A_T.row(i) on segment j-1 and
     * sr_i * A_0.row(i) on segment j.  Solve the 3x3 system for the left control points (c3,c4,c5)_{j-1}
     * in terms of u = (c0,c1,c2)_j:  Al * cl = -Ar * u. */
    for (int j = 1; j < M; ++j) {
        double Al[9], Ar[9];
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) {
                Al[3 * i + c] = q->Aeq[(size_t)(6 + 3 * (j - 1) + i) * oq + 6 * (j - 1) + 3 + c];
                Ar[3 * i + c] = q->Aeq[(size_t)(6 + 3 * (j - 1) + i) * oq + 6 * j + c];
            }
        /* Al is upper-anti-triangular: row0 = (0,0,a), row1 = (0,-b,b), row2 = (c,-2c,c) */
        double* L = k->Lk + 9 * j;
        for (int c = 0; c < 3; ++c) {
            double c5 = -Ar[0 + c] / Al[2];
            double c4 = (-Ar[3 + c] - Al[5] * c5) / Al[4];
            double c3 = (-Ar[6 + c] - Al[7] * c4 - Al[8] * c5) / Al[6];
            L[0 + c] = c3, L[3 + c] = c4, L[6 + c] = c5;
        }
    }
    k->AAt = (double*)calloc((size_t)neb * neb, sizeof(double));
    for (int a = 0; a < neb; ++a)
        for (int b = 0; b <= a; ++b) {
            double s = 0;
            for (int c = 0; c < oq; ++c) s += q->Aeq[(size_t)a * oq + c] * q->Aeq[(size_t)b * oq + c];
            k->AAt[a * neb + b] = s;
        }
    chol_lower(neb, k->AAt, neb);
    k->Td = (double*)malloc(sizeof(double) * (size_t)(M - 1) * nk * nk);
    k->To = (double*)malloc(sizeof(double) * (size_t)(M > 2 ? M - 2 : 1) * nk * nk);
    k->t1 = (double*)malloc(sizeof(double) * q->nx), k->xp = (double*)malloc(sizeof(double) * q->nx);
    k->t2 = (double*)malloc(sizeof(double) * q->ne);
    k->gw = (double*)malloc(sizeof(double) * q->nc);
    k->ru = (double*)malloc(sizeof(double) * (size_t)(M - 1) * nk);
}
static void kkt_free(kkt_t* k) {
    free(k->K), free(k->piv), free(k->Lk), free(k->AAt), free(k->Td), free(k->To), free(k->t1), free(k->t2), free(k->gw),
        free(k->ru), free(k->xp);
}

static int kkt_factor_dense(kkt_t* k, const double* w) {
    const qp_t* q = k->q;
    const int nx = q->nx, ne = q->ne, n = nx + ne, oq = 6 * q->M;
    double* K = k->K;
    memset(K, 0, sizeof(double) * (size_t)n * n);
    for (int u = 0; u < 3 * q->nb; ++u)
        for (int m = 0; m < q->M; ++m)
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j)
                    K[(size_t)(u * oq + 6 * m + i) * n + (u * oq + 6 * m + j)] = 2 * q->Qseg[36 * m + 6 * i + j];
    for (int c = 0; c < q->nc; ++c)
        for (int a = 0; a < q->gnnz[c]; ++a)
            for (int b = 0; b < q->gnnz[c]; ++b)
                K[(size_t)q->gcol[6 * c + a] * n + q->gcol[6 * c + b]] += w[c] * q->gval[6 * c + a] * q->gval[6 * c + b];
    for (int u = 0; u < 3 * q->nb; ++u)
        for (int e = 0; e < q->ne_base; ++e)
            for (int j = 0; j < oq; ++j) {
                double a = q->Aeq[(size_t)e * oq + j];
                if (a == 0) continue;
                int r = nx + u * q->ne_base + e, c = u * oq + j;
                K[(size_t)r * n + c] = a, K[(size_t)c * n + r] = a;
            }
    for (int c = 0; c < n; ++c) { /* LU, partial pivoting */
        int p = c;
        double best = fabs(K[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(K[(size_t)r * n + c]) > best) best = fabs(K[(size_t)r * n + c]), p = r;
        if (best == 0) return 1;
        k->piv[c] = p;
        if (p != c)
            for (int j = c; j < n; ++j) { /* LINPACK-style: multipliers of earlier columns stay in place */
                double tmp = K[(size_t)c * n + j];
                K[(size_t)c * n + j] = K[(size_t)p * n + j], K[(size_t)p * n + j] = tmp;
            }
        double d = 1.0 / K[(size_t)c * n + c];
        for (int r = c + 1; r < n; ++r) {
            double l = K[(size_t)r * n + c] * d;
            if (l == 0) continue;
            K[(size_t)r * n + c] = l;
            for (int j = c + 1; j < n; ++j) K[(size_t)r * n + j] -= l * K[(size_t)c * n + j];
        }
    }
    k->flops += 2.0 / 3.0 * n * (double)n * n;
    return 0;
}
static void kkt_solve_dense(kkt_t* k, const double* r1, const double* r2, double* dx, double* dy) {
    const qp_t* q = k->q;
    const int nx = q->nx, n = nx + q->ne;
    double* b = (double*)malloc(sizeof(double) * n);
    memcpy(b, r1, sizeof(double) * nx);
    memcpy(b + nx, r2, sizeof(double) * q->ne);
    for (int c = 0; c < n; ++c) {
        int p = k->piv[c];
        if (p != c) {
            double tmp = b[c];
            b[c] = b[p], b[p] = tmp;
        }
        for (int r = c + 1; r < n; ++r) b[r] -= k->K[(size_t)r * n + c] * b[c];
    }
    for (int c = n - 1; c >= 0; --c) {
        for (int j = c + 1; j < n; ++j) b[c] -= k->K[(size_t)c * n + j] * b[j];
        b[c] /= k->K[(size_t)c * n + c];
    }
    memcpy(dx, b, sizeof(double) * nx);
    memcpy(dy, b + nx, sizeof(double) * q->ne);
    free(b);
    k->flops += 2.0 * n * (double)n;
}

/* reduced coordinates of a control-point column: knot j (0 or M = fixed end), and the 3 coefficients t with
 * x[col] = t . u_{(u,j)} ; u = k*nb + b */
static inline void col_reduced(const kkt_t* k, int col, int* u, int* j, double t[3]) {
    const qp_t* q = k->q;
    const int oq = 6 * q->M;
    *u = col / oq;
    int jj = col % oq, m = jj / 6, i = jj % 6;
    if (i < 3) {
        *j = m;
        t[0] = t[1] = t[2] = 0, t[i] = 1;
    } else {
        *j = m + 1;
        const double* L = k->Lk + 9 * (m + 1) + 3 * (i - 3);
        t[0] = L[0], t[1] = L[1], t[2] = L[2];
    }
}
/* y_u = F' x  (per (u) and interior knot) */
static void op_FTx(const kkt_t* k, const double* x, double* yu) {
    const qp_t* q = k->q;
    const int M = q->M, oq = 6 * M, nk = k->nk;
    for (int u = 0; u < 3 * q->nb; ++u)
        for (int j = 1; j < M; ++j) {
            const double* xr = x + (size_t)u * oq + 6 * j;           /* right: identity */
            const double* xl = x + (size_t)u * oq + 6 * (j - 1) + 3; /* left: L_j' */
            const double* L = k->Lk + 9 * j;
            for (int e = 0; e < 3; ++e)
                yu[(size_t)(j - 1) * nk + u * 3 + e] = xr[e] + L[0 + e] * xl[0] + L[3 + e] * xl[1] + L[6 + e] * xl[2];
        }
}
/* x += F u */
static void op_Fu_add(const kkt_t* k, const double* uu, double* x) {
    const qp_t* q = k->q;
    const int M = q->M, oq = 6 * M, nk = k->nk;
    for (int u = 0; u < 3 * q->nb; ++u)
        for (int j = 1; j < M; ++j) {
            const double* v = uu + (size_t)(j - 1) * nk + u * 3;
            double* xr = x + (size_t)u * oq + 6 * j;
            double* xl = x + (size_t)u * oq + 6 * (j - 1) + 3;
            const double* L = k->Lk + 9 * j;
            for (int e = 0; e < 3; ++e) {
                xr[e] += v[e];
                xl[e] += L[3 * e] * v[0] + L[3 * e + 1] * v[1] + L[3 * e + 2] * v[2];
            }
        }
}

static int kkt_factor_reduced(kkt_t* k, const double* w) {
    const qp_t* q = k->q;
    const int M = q->M, nk = k->nk, nu = 3 * q->nb, nj = M - 1;
    memset(k->Td, 0, sizeof(double) * (size_t)nj * nk * nk);
    memset(k->To, 0, sizeof(double) * (size_t)(nj > 1 ? nj - 1 : 1) * nk * nk);
    /* F' (2Q) F : identical 3x3 blocks for every (agent, dim) */
    for (int j = 1; j < M; ++j) {
        const double* Ql = q->Qseg + 36 * (j - 1); /* segment left of the knot: its Q22 block through L_j */
        const double* Qr = q->Qseg + 36 * j;       /* segment right of the knot: its Q11 block */
        const double* L = k->Lk + 9 * j;
        double D[9], QL[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0;
                for (int c = 0; c < 3; ++c) s += Ql[6 * (3 + a) + 3 + c] * L[3 * c + b];
                QL[3 * a + b] = s;
            }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double s = 0;
                for (int c = 0; c < 3; ++c) s += L[3 * c + a] * QL[3 * c + b];
                D[3 * a + b] = 2 * (s + Qr[6 * a + b]);
            }
        double* T = k->Td + (size_t)(j - 1) * nk * nk;
        for (int u = 0; u < nu; ++u)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) T[(u * 3 + a) * nk + u * 3 + b] = D[3 * a + b];
        if (j + 1 < M) { /* T_{j+1,j} = (2 Q12^{(j)} L_{j+1})' : rows knot j+1, cols knot j */
            const double* Ln = k->Lk + 9 * (j + 1);
            double E[9];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double s = 0;
                    for (int c = 0; c < 3; ++c) s += Qr[6 * a + 3 + c] * Ln[3 * c + b];
                    E[3 * a + b] = 2 * s; /* rows u_j, cols u_{j+1} */
                }
            double* O = k->To + (size_t)(j - 1) * nk * nk;
            for (int u = 0; u < nu; ++u)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) O[(u * 3 + b) * nk + u * 3 + a] = E[3 * a + b];
        }
    }
    /* F' G' W G F : every inequality row lives in one knot */
    for (int c = 0; c < q->nc; ++c) {
        int idx[18], n = 0, knot = -1;
        double val[18];
        for (int a = 0; a < q->gnnz[c]; ++a) {
            int u, j;
            double t[3];
            col_reduced(k, q->gcol[6 * c + a], &u, &j, t);
            if (j <= 0 || j >= M) continue; /* fixed end control point: constant */
            knot = j;
            for (int e = 0; e < 3; ++e)
                if (t[e] != 0) idx[n] = u * 3 + e, val[n] = q->gval[6 * c + a] * t[e], n++;
        }
        if (knot < 0) continue;
        double* T = k->Td + (size_t)(knot - 1) * nk * nk;
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) T[idx[a] * nk + idx[b]] += w[c] * val[a] * val[b];
    }
    /* block tridiagonal Cholesky */
    for (int j = 0; j < nj; ++j) {
        double* D = k->Td + (size_t)j * nk * nk;
        if (j > 0) {
            const double* Lo = k->To + (size_t)(j - 1) * nk * nk; /* L_{j,j-1} */
            for (int a = 0; a < nk; ++a)
                for (int b = 0; b <= a; ++b) {
                    double s = 0;
                    for (int c = 0; c < nk; ++c) s += Lo[a * nk + c] * Lo[b * nk + c];
                    D[a * nk + b] -= s;
                }
        }
        if (chol_lower(nk, D, nk)) return 1;
        if (j + 1 < nj) { /* L_{j+1,j} = T_{j+1,j} D^{-T}:  solve D X' = T_{j+1,j}'  */
            double* O = k->To + (size_t)j * nk * nk;
            /* transpose, forward-solve, transpose back */
            for (int a = 0; a < nk; ++a)
                for (int b = a + 1; b < nk; ++b) {
                    double tmp = O[a * nk + b];
                    O[a * nk + b] = O[b * nk + a], O[b * nk + a] = tmp;
                }
            trsm_lower(nk, nk, D, nk, O, nk);
            for (int a = 0; a < nk; ++a)
                for (int b = a + 1; b < nk; ++b) {
                    double tmp = O[a * nk + b];
                    O[a * nk + b] = O[b * nk + a], O[b * nk + a] = tmp;
                }
        }
        k->flops += 7.0 / 3.0 * nk * (double)nk * nk;
    }
    return 0;
}

static void apply_Hbar(const kkt_t* k, const double* x, double* y) { /* y = (2Q + G'WG) x */
    const qp_t* q = k->q;
    op_Hx(q, x, y);
    op_Gx(q, x, k->gw);
    for (int c = 0; c < q->nc; ++c) k->gw[c] *= k->w[c];
    op_GTz_add(q, k->gw, y);
}
/* per (u): v <- (A A')^{-1} v */
static void solve_AAt(const kkt_t* k, double* v) {
    const qp_t* q = k->q;
    for (int u = 0; u < 3 * q->nb; ++u) {
        trsm_lower(q->ne_base, 1, k->AAt, q->ne_base, v + (size_t)u * q->ne_base, 1);
        trsm_lower_t(q->ne_base, 1, k->AAt, q->ne_base, v + (size_t)u * q->ne_base, 1);
    }
}

static void kkt_solve_reduced(kkt_t* k, const double* r1, const double* r2, double* dx, double* dy) {
    const qp_t* q = k->q;
    const int M = q->M, nk = k->nk, nj = M - 1, nx = q->nx, ne = q->ne;
    /* particular solution A xp = r2 (minimum norm) */
    memcpy(k->t2, r2, sizeof(double) * ne);
    solve_AAt(k, k->t2);
    memset(k->xp, 0, sizeof(double) * nx);
    op_ATy_add(q, k->t2, k->xp);
    /* reduced rhs F'(r1 - Hbar xp) */
    apply_Hbar(k, k->xp, k->t1);
    for (int i = 0; i < nx; ++i) k->t1[i] = r1[i] - k->t1[i];
    op_FTx(k, k->t1, k->ru);
    for (int j = 0; j < nj; ++j) { /* forward */
        double* v = k->ru + (size_t)j * nk;
        if (j > 0) {
            const double* Lo = k->To + (size_t)(j - 1) * nk * nk;
            for (int a = 0; a < nk; ++a) {
                double s = 0;
                for (int c = 0; c < nk; ++c) s += Lo[a * nk + c] * k->ru[(size_t)(j - 1) * nk + c];
                v[a] -= s;
            }
        }
        trsm_lower(nk, 1, k->Td + (size_t)j * nk * nk, nk, v, 1);
    }
    for (int j = nj - 1; j >= 0; --j) { /* backward */
        double* v = k->ru + (size_t)j * nk;
        if (j + 1 < nj) {
            const double* Lo = k->To + (size_t)j * nk * nk; /* L_{j+1,j} */
            for (int c = 0; c < nk; ++c) {
                double s = 0;
                for (int a = 0; a < nk; ++a) s += Lo[a * nk + c] * k->ru[(size_t)(j + 1) * nk + a];
                v[c] -= s;
            }
        }
        trsm_lower_t(nk, 1, k->Td + (size_t)j * nk * nk, nk, v, 1);
    }
    memcpy(dx, k->xp, sizeof(double) * nx);
    op_Fu_add(k, k->ru, dx);
    /* dy = (AA')^{-1} A (r1 - Hbar dx) */
    apply_Hbar(k, dx, k->t1);
    for (int i = 0; i < nx; ++i) k->t1[i] = r1[i] - k->t1[i];
    op_Ax(q, k->t1, dy);
    solve_AAt(k, dy);
    k->flops += (double)nj * 4.0 * nk * nk;
}
static int kkt_factor(kkt_t* k, const double* w) {
    k->w = w;
    return k->mode ? kkt_factor_dense(k, w) : kkt_factor_reduced(k, w);
}
static void kkt_solve(kkt_t* k, const double* r1, const double* r2, double* dx, double* dy) {
    if (k->mode)
        kkt_solve_dense(k, r1, r2, dx, dy);
    else
        kkt_solve_reduced(k, r1, r2, dx, dy);
}

/* ------------------------------------------------------------------------------------------------
 * Mehrotra predictor-corrector IPM for   min x'Qx   s.t.  Ax = b,  Gx + s = h,  s >= 0
 * ---------------------------------------------------------------------------------------------- */
void oracle_qp_default_options(oracle_qp_options* o) {
    o->linear_solver = 0, o->max_iter = 60, o->tol_feas = 1e-9, o->tol_gap = 1e-10, o->verbose = 0, o->polish = 1;
}

static double inf_norm(const double* v, int n) {
    double m = 0;
    for (int i = 0; i < n; ++i)
        if (fabs(v[i]) > m) m = fabs(v[i]);
    return m;
}

static int qp_polish(const qp_t* q, int linear_solver, double* x, double* y, double* z, double* s, double* flops,
                     int verbose);

typedef struct {
    double *x, *y, *z, *s;
    int iters, polished;
    double obj, dual_obj;
} qp_sol_t;

static int qp_solve(const qp_t* q, const double* x0, const oracle_qp_options* opt, qp_sol_t* sol,
                    oracle_qp_report* rep) {
    const int nx = q->nx, ne = q->ne, nc = q->nc;
    double* x = sol->x = (double*)calloc(nx, sizeof(double));
    double* y = sol->y = (double*)calloc(ne, sizeof(double));
    double* z = sol->z = (double*)calloc(nc, sizeof(double));
    double* s = sol->s = (double*)calloc(nc, sizeof(double));
    double* w = (double*)malloc(sizeof(double) * nc);
    double* rd = (double*)malloc(sizeof(double) * nx);
    double* rp = (double*)malloc(sizeof(double) * ne);
    double* rg = (double*)malloc(sizeof(double) * nc);
    double* r1 = (double*)malloc(sizeof(double) * nx);
    double* r2 = (double*)malloc(sizeof(double) * ne);
    double* dx = (double*)malloc(sizeof(double) * nx);
    double* dy = (double*)malloc(sizeof(double) * ne);
    double* dz = (double*)malloc(sizeof(double) * nc);
    double* ds = (double*)malloc(sizeof(double) * nc);
    double* dza = (double*)malloc(sizeof(double) * nc);
    double* dsa = (double*)malloc(sizeof(double) * nc);
    double* tc = (double*)malloc(sizeof(double) * nc);
    double* Hx = (double*)malloc(sizeof(double) * nx);
    kkt_t K;
    kkt_init(&K, q, opt->linear_solver);
    int rc = RBP_ERR_QP_FAILED;
    const double dreg = 1e-9;

    /* starting point: x0 = warm start (the reference's `dummy` control points: waypoint-constant segments, inside
     * their SFC boxes and RSFC half-spaces by construction), y0 = 0, s0 = max(h - G x0, s_floor), z0 = mu0 / s0
     * (perfectly centred).  Equality infeasibility of x0 is removed by the first full Newton step. */
    {
        const double s_floor = 1e-1, mu0 = 1e-1;  /* tuned on the 50-map sweep: fewest iterations of the grid tried */
        memcpy(x, x0, sizeof(double) * nx);
        op_Gx(q, x, tc);
        for (int c = 0; c < nc; ++c) {
            s[c] = q->h[c] - tc[c];
            if (s[c] < s_floor) s[c] = s_floor;
            z[c] = mu0 / s[c];
        }
    }

    int it;
    for (it = 0; it < opt->max_iter; ++it) {
        /* residuals */
        op_Hx(q, x, Hx);
        memcpy(rd, Hx, sizeof(double) * nx);
        op_ATy_add(q, y, rd);
        op_GTz_add(q, z, rd);
        op_Ax(q, x, rp);
        for (int e = 0; e < ne; ++e) rp[e] -= q->beq[e];
        op_Gx(q, x, rg);
        double gap = 0;
        for (int c = 0; c < nc; ++c) {
            rg[c] += s[c] - q->h[c];
            gap += s[c] * z[c];
        }
        double mu = gap / nc;
        double obj = 0;
        for (int i = 0; i < nx; ++i) obj += 0.5 * x[i] * Hx[i];
        double pres = fmax(inf_norm(rp, ne), inf_norm(rg, nc));
        double dres = inf_norm(rd, nx) / (1.0 + inf_norm(Hx, nx));
        double relgap = gap / fmax(1.0, fabs(obj));
        if (opt->verbose) printf("  it %2d obj %.10g pres %.2e dres %.2e gap %.2e mu %.2e\n", it, obj, pres, dres, gap, mu);
        (void)relgap;
        /* converged: residuals tiny and complementarity measure mu = s'z/nc below tol_gap */
        if (pres < opt->tol_feas && dres < opt->tol_feas && mu < opt->tol_gap) {
            rc = RBP_OK;
            break;
        }
        /* rows inconsistent at rounding level (no interior AND infeasible by ~1e-9, see qp_polish): the regularised
         * iteration settles on the least-violation point with pres stuck.  Accept it once complementarity and
         * stationarity are converged and the violation is below CPLEX's default feasibility tolerance 1e-6. */
        if (pres < 1e-6 && dres < opt->tol_feas && mu < 1e-3 * opt->tol_gap) {
            rc = RBP_OK;
            if (rep) rep->n_loose++;
            break;
        }
        /* dual proximal regularisation of the inequality rows (Friedlander-Orban):  G dx + ds - dreg dz = -rg.
         * It vanishes at a fixed point (dz = 0) but caps the barrier weights at 1/dreg, which keeps the Newton
         * systems well posed when the feasible set has NO interior -- common here: SFC faces and r_i + r_j = 0.3
         * live on the same 0.1 m lattice, so a frozen neighbour on a box face plus an RSFC row often pins a
         * control point between two parallel rows with zero gap (an implied equality). */
        for (int c = 0; c < nc; ++c) w[c] = 1.0 / (s[c] / z[c] + dreg);
        if (kkt_factor(&K, w)) {
            if (opt->verbose) printf("  factor failed\n");
            break;
        }
        /* predictor: rc = s.z */
        for (int c = 0; c < nc; ++c) tc[c] = -w[c] * (rg[c] - s[c]);
        for (int i = 0; i < nx; ++i) r1[i] = -rd[i];
        op_GTz_add(q, tc, r1); /* r1 = -rd - G' W (rg - rc/z) */
        for (int e = 0; e < ne; ++e) r2[e] = -rp[e];
        kkt_solve(&K, r1, r2, dx, dy);
        op_Gx(q, dx, dsa);
        double alpha = 1.0;
        for (int c = 0; c < nc; ++c) {
            double gdx = dsa[c];
            dza[c] = w[c] * (gdx + rg[c] - s[c]);
            dsa[c] = (-s[c] * z[c] - s[c] * dza[c]) / z[c];
            if (dsa[c] < 0) alpha = fmin(alpha, -s[c] / dsa[c]);
            if (dza[c] < 0) alpha = fmin(alpha, -z[c] / dza[c]);
        }
        double mu_aff = 0;
        for (int c = 0; c < nc; ++c) mu_aff += (s[c] + alpha * dsa[c]) * (z[c] + alpha * dza[c]);
        mu_aff /= nc;
        double sigma = pow(mu_aff / mu, 3.0);
        /* corrector: rc = s.z + dsa.dza - sigma mu */
        for (int c = 0; c < nc; ++c) {
            double rcc = s[c] * z[c] + dsa[c] * dza[c] - sigma * mu;
            tc[c] = -w[c] * (rg[c] - rcc / z[c]);
        }
        for (int i = 0; i < nx; ++i) r1[i] = -rd[i];
        op_GTz_add(q, tc, r1);
        kkt_solve(&K, r1, r2, dx, dy);
        op_Gx(q, dx, ds);
        alpha = 1e300;
        for (int c = 0; c < nc; ++c) {
            double gdx = ds[c], rcc = s[c] * z[c] + dsa[c] * dza[c] - sigma * mu;
            dz[c] = w[c] * (gdx + rg[c] - rcc / z[c]);
            ds[c] = (-rcc - s[c] * dz[c]) / z[c];
            if (ds[c] < 0) alpha = fmin(alpha, -s[c] / ds[c]);
            if (dz[c] < 0) alpha = fmin(alpha, -z[c] / dz[c]);
        }
        if (opt->verbose > 2 && alpha < 0.05) {
            for (int c = 0; c < nc; ++c) {
                double as = ds[c] < 0 ? -s[c] / ds[c] : 1e300, az = dz[c] < 0 ? -z[c] / dz[c] : 1e300;
                if (fmin(as, az) < 1.5 * alpha) {
                    printf("      block row %d nnz %d cols", c, q->gnnz[c]);
                    for (int e = 0; e < q->gnnz[c]; ++e) printf(" %d(%.3f)", q->gcol[6 * c + e], q->gval[6 * c + e]);
                    printf(" h %.6f s %.3e z %.3e ds %.3e dz %.3e rg %.3e\n", q->h[c], s[c], z[c], ds[c], dz[c], rg[c]);
                }
            }
        }
        /* EXPERIMENT (env ORACLE_GONDZIO = number of correctors, default 0 = off; not part of the checked path): Gondzio's multiple
         * centrality correctors on top of Mehrotra's direction.  Each corrector re-solves with the complementarity target moved by the
         * accumulated term T (products of the trial point projected onto [bmin, bmax] * sigma mu) and is kept if the step grows. */
        {
            static int ngond = -1;
            if (ngond < 0) ngond = getenv("ORACLE_GONDZIO") ? atoi(getenv("ORACLE_GONDZIO")) : 0;
            if (ngond > 0) {
                double* T = (double*)calloc(nc, sizeof(double));
                double* tn = (double*)malloc(sizeof(double) * nc);
                double* dx2 = (double*)malloc(sizeof(double) * nx);
                double* dy2 = (double*)malloc(sizeof(double) * ne);
                double* ds2 = (double*)malloc(sizeof(double) * nc);
                double* dz2 = (double*)malloc(sizeof(double) * nc);
                const double dalpha = getenv("OG_DA") ? atof(getenv("OG_DA")) : 0.3, bmin = getenv("OG_BMIN") ? atof(getenv("OG_BMIN")) : 0.1, bmax = getenv("OG_BMAX") ? atof(getenv("OG_BMAX")) : 10.0;
                const double mut = fmax(sigma * mu, 1e-3 * mu);
                for (int k = 0; k < ngond; ++k) {
                    const double ap = fmin(1.0, 0.99 * alpha);
                    if (ap >= 1.0) break;
                    const double at = fmin(1.0, ap + dalpha);
                    for (int c = 0; c < nc; ++c) {
                        double v = (s[c] + at * ds[c]) * (z[c] + at * dz[c]);
                        double vt = v < bmin * mut ? bmin * mut : (v > bmax * mut ? bmax * mut : v);
                        double t = vt - v;
                        if (t < -bmax * mut) t = -bmax * mut;
                        tn[c] = T[c] + t;
                    }
                    for (int c = 0; c < nc; ++c) {
                        double rcc = s[c] * z[c] + dsa[c] * dza[c] - sigma * mu - tn[c];
                        tc[c] = -w[c] * (rg[c] - rcc / z[c]);
                    }
                    for (int i = 0; i < nx; ++i) r1[i] = -rd[i];
                    op_GTz_add(q, tc, r1);
                    kkt_solve(&K, r1, r2, dx2, dy2);
                    op_Gx(q, dx2, ds2);
                    double a2 = 1e300;
                    for (int c = 0; c < nc; ++c) {
                        double gdx = ds2[c], rcc = s[c] * z[c] + dsa[c] * dza[c] - sigma * mu - tn[c];
                        dz2[c] = w[c] * (gdx + rg[c] - rcc / z[c]);
                        ds2[c] = (-rcc - s[c] * dz2[c]) / z[c];
                        if (ds2[c] < 0) a2 = fmin(a2, -s[c] / ds2[c]);
                        if (dz2[c] < 0) a2 = fmin(a2, -z[c] / dz2[c]);
                    }
                    if (opt->verbose > 1) printf("      gondzio %d: alpha %.4f -> %.4f\n", k, ap, fmin(1.0, 0.99 * a2));
                    if (fmin(1.0, 0.99 * a2) < ap + 0.1 * dalpha) break;
                    memcpy(T, tn, sizeof(double) * nc);
                    memcpy(dx, dx2, sizeof(double) * nx), memcpy(dy, dy2, sizeof(double) * ne);
                    memcpy(ds, ds2, sizeof(double) * nc), memcpy(dz, dz2, sizeof(double) * nc);
                    alpha = a2;
                }
                free(T), free(tn), free(dx2), free(dy2), free(ds2), free(dz2);
            }
        }
        alpha = fmin(1.0, 0.99 * alpha);
        /* stay in the wide neighbourhood N_-inf(gamma): no complementarity product may fall below gamma * mu(alpha).
         * Without this a few products collapse early and the iteration jams (alpha -> 0) near the solution. */
        for (int bt = 0; bt < 40; ++bt) {
            double mu_new = 0, pmin = 1e300;
            for (int c = 0; c < nc; ++c) {
                double pr = (s[c] + alpha * ds[c]) * (z[c] + alpha * dz[c]);
                mu_new += pr;
                if (pr < pmin) pmin = pr;
            }
            mu_new /= nc;
            if (pmin >= 1e-3 * mu_new) break;
            alpha *= 0.8;
        }
        if (opt->verbose > 1) printf("      sigma %.3e alpha %.4f\n", sigma, alpha);
        for (int i = 0; i < nx; ++i) x[i] += alpha * dx[i];
        for (int e = 0; e < ne; ++e) y[e] += alpha * dy[e];
        for (int c = 0; c < nc; ++c) s[c] += alpha * ds[c], z[c] += alpha * dz[c];
    }
    sol->iters = it;
    if (rc == RBP_OK && opt->polish) {
        double pf = 0;
        sol->polished = !qp_polish(q, opt->linear_solver, x, y, z, s, &pf, opt->verbose);
        K.flops += pf;
        if (rep) rep->n_polished += sol->polished;
    }
    /* certificate: KKT residuals at the returned point, independent of the iteration's bookkeeping */
    {
        op_Hx(q, x, Hx);
        memcpy(rd, Hx, sizeof(double) * nx);
        op_ATy_add(q, y, rd);
        op_GTz_add(q, z, rd);
        op_Ax(q, x, rp);
        for (int e = 0; e < ne; ++e) rp[e] -= q->beq[e];
        op_Gx(q, x, rg);
        double obj = 0, vi = 0, zmin = 1e300, compl = 0, dual = 0;
        for (int i = 0; i < nx; ++i) obj += 0.5 * x[i] * Hx[i];
        for (int c = 0; c < nc; ++c) {
            double slack = q->h[c] - rg[c];
            if (-slack > vi) vi = -slack;
            if (z[c] < zmin) zmin = z[c];
            if (fabs(z[c] * slack) > compl) compl = fabs(z[c] * slack);
            dual -= q->h[c] * z[c];
        }
        /* Lagrange dual of min 1/2 x'Hx: with stationarity Hx + A'y + G'z = 0 -> d = -1/2 x'Hx - b'y - h'z */
        for (int e = 0; e < ne; ++e) dual -= q->beq[e] * y[e];
        dual -= obj;
        sol->obj = obj, sol->dual_obj = dual;
        if (rep) {
            rep->n_qp++;
            rep->iters_total += it;
            if (it > rep->iters_max) rep->iters_max = it;
            rep->kkt_stationarity = fmax(rep->kkt_stationarity, inf_norm(rd, nx) / (1.0 + inf_norm(Hx, nx)));
            rep->kkt_primal_eq = fmax(rep->kkt_primal_eq, inf_norm(rp, ne));
            rep->kkt_primal_ineq = fmax(rep->kkt_primal_ineq, vi);
            rep->kkt_dual_min = fmin(rep->kkt_dual_min, zmin);
            rep->kkt_compl = fmax(rep->kkt_compl, compl);
            rep->duality_gap_rel = fmax(rep->duality_gap_rel, fabs(obj - dual) / fmax(1.0, fabs(obj)));
            rep->flops += K.flops;
        }
    }
    kkt_free(&K);
    free(w), free(rd), free(rp), free(rg), free(r1), free(r2), free(dx), free(dy), free(dz), free(ds), free(dza),
        free(dsa), free(tc), free(Hx);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Active-set polish ("crossover").  The interior-point iterate identifies a candidate set C of rows that may be
 * active (small slack or dominant multiplier).  The optimum of the QP is then obtained EXACTLY from the dual QP
 * restricted to C,
 *        min_z  1/2 z' S z - d' z ,  z >= 0 ,      S = G_C K0^{-1} G_C' ,  d = G_C x0 - h_C ,
 * (K0 = KKT matrix of the equality-constrained problem, x0 = its solution), solved by the Lawson-Hanson
 * active-set method: finite, monotone, and indifferent to linearly dependent rows -- which are the rule here,
 * SFC faces and r_i + r_j sharing one 0.1 m lattice.  x = x0 - V z.  Rows outside C that come out violated are
 * added to C and the iteration continues.  The result is ACCEPTED ONLY IF it satisfies every KKT condition of the
 * full QP.  This removes the O(sqrt(mu)) bias an interior-point answer keeps in the weakly curved directions of
 * this problem (reduced-Hessian condition number ~1e9).  returns 0 if accepted (x, y, z, s overwritten).
 * ---------------------------------------------------------------------------------------------- */
static int qp_polish(const qp_t* q, int linear_solver, double* x, double* y, double* z, double* s, double* flops,
                     int verbose) {
    const int nx = q->nx, ne = q->ne, nc = q->nc, cap = 1024;
    int* cand = (int*)malloc(sizeof(int) * cap);
    char* in_c = (char*)calloc(nc, 1);
    int ncand = 0;
    for (int c = 0; c < nc; ++c)
        if ((z[c] > s[c] || s[c] < 1e-6) && ncand < cap) cand[ncand++] = c, in_c[c] = 1;
    double* w0 = (double*)calloc(nc, sizeof(double));
    kkt_t K;
    kkt_init(&K, q, linear_solver);
    int rc = 1;
    double* V = (double*)malloc(sizeof(double) * (size_t)cap * nx);  /* x-part of K0^{-1}[g_c;0], one row per candidate */
    double* Yv = (double*)malloc(sizeof(double) * (size_t)cap * ne); /* y-part */
    double* S = (double*)malloc(sizeof(double) * (size_t)cap * cap);
    double* Sp = (double*)malloc(sizeof(double) * (size_t)cap * cap);
    double* x0 = (double*)malloc(sizeof(double) * nx), *xn = (double*)malloc(sizeof(double) * nx);
    double* y0 = (double*)malloc(sizeof(double) * ne), *yn = (double*)malloc(sizeof(double) * ne);
    double* r1 = (double*)calloc(nx, sizeof(double)), *r2 = (double*)calloc(ne, sizeof(double));
    double* zc = (double*)calloc(cap, sizeof(double)), *d = (double*)malloc(sizeof(double) * cap);
    double* yv = (double*)malloc(sizeof(double) * cap), *wv = (double*)malloc(sizeof(double) * cap);
    int* P = (int*)malloc(sizeof(int) * cap);
    char* in_p = (char*)calloc(cap, 1);
    double* gx = (double*)malloc(sizeof(double) * nc);
    int nV = 0, nP = 0, lh_iters = 0, rejected = 0, redo = 0, drops = 0;
    if (kkt_factor(&K, w0)) goto out;
    kkt_solve(&K, r1, q->beq, x0, y0); /* K0 [x0;y0] = [0;b] */
    op_Gx(q, x0, gx);
    for (int outer = 0; outer < 20; ++outer) {
        for (; nV < ncand; ++nV) { /* new candidates: K0 [V;Y] = [g;0], extend S and d */
            int c = cand[nV];
            memset(r1, 0, sizeof(double) * nx);
            for (int e = 0; e < q->gnnz[c]; ++e) r1[q->gcol[6 * c + e]] = q->gval[6 * c + e];
            kkt_solve(&K, r1, r2, V + (size_t)nV * nx, Yv + (size_t)nV * ne);
            d[nV] = gx[c] - q->h[c];
            zc[nV] = 0, in_p[nV] = 0;
            for (int b = 0; b <= nV; ++b) {
                int cb = cand[b];
                double sum = 0, sum2 = 0;
                for (int e = 0; e < q->gnnz[c]; ++e) sum += q->gval[6 * c + e] * V[(size_t)b * nx + q->gcol[6 * c + e]];
                for (int e = 0; e < q->gnnz[cb]; ++e) sum2 += q->gval[6 * cb + e] * V[(size_t)nV * nx + q->gcol[6 * cb + e]];
                S[(size_t)nV * cap + b] = S[(size_t)b * cap + nV] = 0.5 * (sum + sum2);
            }
        }
        /* Lawson-Hanson on the candidate rows */
        double dscale = 1e-300;
        for (int a = 0; a < ncand; ++a) dscale = fmax(dscale, fabs(d[a]));
        const double wtol = 1e-12 * fmax(1.0, dscale);
        for (int itn = 0; itn < 4 * ncand + 10; ++itn, ++lh_iters) {
            int jbest = -1;
            double wbest = wtol;
            for (int a = 0; a < ncand; ++a) {
                double sum = d[a];
                for (int b = 0; b < nP; ++b) sum -= S[(size_t)a * cap + P[b]] * zc[P[b]];
                wv[a] = sum; /* residual violation of row a */
                if (!in_p[a] && sum > wbest) wbest = sum, jbest = a;
            }
            if (jbest < 0) break;
            P[nP++] = jbest, in_p[jbest] = 1;
            for (int inner = 0; inner < 2 * cap; ++inner) {
                double tr = 0;
                for (int a = 0; a < nP; ++a) {
                    for (int b = 0; b <= a; ++b) Sp[(size_t)a * nP + b] = S[(size_t)P[a] * cap + P[b]];
                    tr += Sp[(size_t)a * nP + a];
                }
                for (int a = 0; a < nP; ++a) Sp[(size_t)a * nP + a] += 1e-14 * tr / nP + 1e-300;
                for (int a = 0; a < nP; ++a) yv[a] = d[P[a]];
                if (chol_lower(nP, Sp, nP)) { /* dependent row slipped in: drop the newcomer */
                    in_p[P[nP - 1]] = 0, nP--;
                    break;
                }
                trsm_lower(nP, 1, Sp, nP, yv, 1);
                trsm_lower_t(nP, 1, Sp, nP, yv, 1);
                *flops += (double)nP * nP * nP / 3;
                double ymin = 1e300;
                for (int a = 0; a < nP; ++a) ymin = fmin(ymin, yv[a]);
                if (ymin > 0) {
                    for (int a = 0; a < nP; ++a) zc[P[a]] = yv[a];
                    break;
                }
                double alpha = 1e300;
                for (int a = 0; a < nP; ++a)
                    if (yv[a] <= 0) alpha = fmin(alpha, zc[P[a]] / (zc[P[a]] - yv[a]));
                if (!(alpha >= 0)) alpha = 0;
                int keep = 0;
                for (int a = 0; a < nP; ++a) {
                    int ia = P[a];
                    zc[ia] += alpha * (yv[a] - zc[ia]);
                    if (yv[a] <= 0 && zc[ia] <= 1e-14 * fmax(1.0, fabs(yv[a]))) {
                        zc[ia] = 0, in_p[ia] = 0;
                        continue;
                    }
                    P[keep++] = ia;
                }
                if (keep == nP) { /* numerical safety: drop the most negative target */
                    int worst = 0;
                    for (int a = 1; a < nP; ++a)
                        if (yv[a] < yv[worst]) worst = a;
                    zc[P[worst]] = 0, in_p[P[worst]] = 0;
                    for (int a = worst; a + 1 < nP; ++a) P[a] = P[a + 1];
                    keep = nP - 1;
                }
                nP = keep;
                if (nP == 0) break;
            }
        }
        /* primal point; iterative refinement of z on the active rows against the ACTUAL residual G_P x - h_P
         * (S and d were formed from V and x0, which carry the ~1e-9 relative error of solves with K0) */
      for (int attempt = 0; attempt < 2 && rc; ++attempt) {
        if (attempt == 0) memcpy(wv, zc, sizeof(double) * ncand); /* the Lawson-Hanson multipliers, before refinement */
        else memcpy(zc, wv, sizeof(double) * ncand);              /* second attempt: verify them as they are */
        const int last_round = attempt == 0 ? 4 : 0;
        for (int round = 0; round < 5; ++round) {
            memcpy(xn, x0, sizeof(double) * nx);
            for (int a = 0; a < nP; ++a) {
                const double* v = V + (size_t)P[a] * nx;
                double zz = zc[P[a]];
                for (int i = 0; i < nx; ++i) xn[i] -= zz * v[i];
            }
            if (nP == 0 || round == last_round) break;
            double rmax = 0;
            for (int a = 0; a < nP; ++a) {
                int c = cand[P[a]];
                double sum = -q->h[c];
                for (int e = 0; e < q->gnnz[c]; ++e) sum += q->gval[6 * c + e] * xn[q->gcol[6 * c + e]];
                yv[a] = sum;
                rmax = fmax(rmax, fabs(sum));
            }
            if (rmax < 1e-13) break;
            double tr = 0;
            for (int a = 0; a < nP; ++a) {
                for (int b = 0; b <= a; ++b) Sp[(size_t)a * nP + b] = S[(size_t)P[a] * cap + P[b]];
                tr += Sp[(size_t)a * nP + a];
            }
            for (int a = 0; a < nP; ++a) Sp[(size_t)a * nP + a] += 1e-14 * tr / nP + 1e-300;
            if (chol_lower(nP, Sp, nP)) break;
            trsm_lower(nP, 1, Sp, nP, yv, 1);
            trsm_lower_t(nP, 1, Sp, nP, yv, 1);
            for (int a = 0; a < nP; ++a) zc[P[a]] += yv[a];
        }
        op_Gx(q, xn, gx);
        double vmax = 0, zmin = 0, zmax = 0;
        int n_add = 0;
        for (int a = 0; a < nP; ++a) zmin = fmin(zmin, zc[P[a]]), zmax = fmax(zmax, zc[P[a]]);
        for (int c = 0; c < nc; ++c) {
            double v = gx[c] - q->h[c];
            if (v > vmax) vmax = v;
            if (v > 1e-11 && !in_c[c] && ncand < cap) cand[ncand++] = c, in_c[c] = 1, n_add++;
        }
        if (verbose)
            printf("    polish round %d: candidates %d, active %d, LH iterations %d, max violation %.3e, added %d, zmin %.3e zmax %.3e\n", outer,
                   ncand - n_add, nP, lh_iters, vmax, n_add, zmin, zmax);
        if (n_add == 0) {
            /* feasibility 5e-9: rows numerically dependent on the active ones (lattice-aligned SFC faces against r_i + r_j rows)
             * can stay inconsistent at the 1e-9 level whatever the active set; multipliers reach 1e3, so their sign test is
             * relative.  (Both thresholds as in the HIP kernel's polish.) */
            if (vmax > 5e-9 || zmin < -1e-9 * fmax(1.0, zmax)) { /* not converged */
                if (attempt == 0 && vmax <= 5e-9 && drops < 8) {
                    /* the refined multipliers say some active rows should not be active (the dual solve's S carries the ~1e-9
                     * relative error of solves with K0): they leave, and Lawson-Hanson continues from the refined point */
                    int keep = 0;
                    for (int a = 0; a < nP; ++a) {
                        if (zc[P[a]] < 0) zc[P[a]] = 0, in_p[P[a]] = 0;
                        else P[keep++] = P[a];
                    }
                    nP = keep, drops++, redo = 1;
                    break;
                }
                if (attempt == 0) continue; /* the refinement can push a multiplier of nearly dependent rows negative: verify the unrefined ones */
                rejected = 1;
                break;
            }
            memcpy(yn, y0, sizeof(double) * ne);
            for (int a = 0; a < nP; ++a)
                for (int e = 0; e < ne; ++e) yn[e] -= zc[P[a]] * Yv[(size_t)P[a] * ne + e];
            /* restore A x = b to rounding level (x0 and the columns of V carry ~1e-13 relative solve error, scaled
             * by multipliers up to 1e3), then put control points with an ACTIVE SFC bound exactly on the face.
             * Both matter for the next Gauss-Seidel batch: its rows against these (then frozen) control points sit
             * on the same 0.1 m lattice, and a 1e-9 inconsistency there makes that batch infeasible by 1e-9. */
            if (!linear_solver) {
                op_Ax(q, xn, r2);
                for (int e = 0; e < ne; ++e) r2[e] -= q->beq[e];
                solve_AAt(&K, r2);
                for (int e = 0; e < ne; ++e) r2[e] = -r2[e];
                op_ATy_add(q, r2, xn);
                memset(r2, 0, sizeof(double) * ne);
            }
            for (int a = 0; a < nP; ++a) {
                int c = cand[P[a]];
                if (q->gnnz[c] == 1 && zc[P[a]] > 0) xn[q->gcol[6 * c]] = q->h[c] / q->gval[6 * c];
            }
            op_Gx(q, xn, gx);
            memcpy(x, xn, sizeof(double) * nx);
            memcpy(y, yn, sizeof(double) * ne);
            for (int c = 0; c < nc; ++c) z[c] = 0, s[c] = fmax(q->h[c] - gx[c], 0.0);
            for (int a = 0; a < nP; ++a) z[cand[P[a]]] = zc[P[a]];
            rc = 0;
            break;
        }
        break; /* new candidates: extend V, S, d and solve the dual again */
      }
        if (redo) {
            redo = 0;
            op_Gx(q, x0, gx);
            continue;
        }
        if (rc == 0 || rejected) break;
        op_Gx(q, x0, gx); /* d of the new candidates is measured at x0 */
    }
out:
    *flops += K.flops;
    kkt_free(&K);
    free(cand), free(in_c), free(w0), free(V), free(Yv), free(S), free(Sp), free(x0), free(xn), free(y0), free(yn), free(r1),
        free(r2), free(zc), free(d), free(yv), free(wv), free(P), free(in_p), free(gx);
    return rc;
}

static void qp_sol_free(qp_sol_t* s) { free(s->x), free(s->y), free(s->z), free(s->s); }

/* ------------------------------------------------------------------------------------------------
 * post-processing
 * ---------------------------------------------------------------------------------------------- */
void oracle_ctrl_to_coef(int N, int M, const double* T, const double* ctrl, double* coef) { /* :170-196 */
    double basis[36];
    oracle_build_Q_base(NULL, basis);
    for (int qi = 0; qi < N; ++qi)
        for (int k = 0; k < 3; ++k)
            for (int m = 0; m < M; ++m) {
                double tm[36], inv = 1.0 / (T[m + 1] - T[m]);
                for (int i = 0; i < 6; ++i)
                    for (int c = 0; c < 6; ++c) tm[6 * i + c] = basis[6 * i + c] * pow(inv, 5 - c); /* basis * timeMatrix */
                const double* v = ctrl + ((size_t)qi * 3 + k) * 6 * M + 6 * m;
                double* out = coef + ((size_t)qi * 3 + k) * 6 * M + 6 * m;
                for (int c = 0; c < 6; ++c) {
                    double acc = 0;
                    for (int i = 0; i < 6; ++i) acc = acc + v[i] * tm[6 * i + c];
                    out[c] = acc;
                }
            }
}

static int coef_derivative(int i, int j) { return (i == 0) ? 1 : coef_derivative(i - 1, j - 1) * j; } /* :721-723 */

/* real roots of a*t^3 + b*t^2 + c*t + d (leading zeros stripped like roots_derivative :729-736).
 * DEVIATION (documented in oracle/README.md): the reference takes eigenvalues of the companion matrix from
 * Eigen::EigenSolver and, through the `j < i` bound at :747, looks only at the first two of them in Eigen's
 * internal order; Eigen is absent and that order is not reproducible, so ALL real roots are considered. */
static int real_roots(const double* c, int deg, double* out) {
    while (deg > 0 && c[0] == 0) c++, deg--;
    if (deg == 0) return 0;
    if (deg == 1) {
        out[0] = -c[1] / c[0];
        return 1;
    }
    if (deg == 2) {
        double D = c[1] * c[1] - 4 * c[0] * c[2];
        if (D < 0) return 0;
        double sq = sqrt(D);
        out[0] = (-c[1] + sq) / (2 * c[0]), out[1] = (-c[1] - sq) / (2 * c[0]);
        return 2;
    }
    /* cubic: trigonometric / Cardano on the depressed form, polished by Newton */
    double a = c[1] / c[0], b = c[2] / c[0], d = c[3] / c[0];
    double p = b - a * a / 3, qq = 2 * a * a * a / 27 - a * b / 3 + d;
    double disc = qq * qq / 4 + p * p * p / 27;
    int n = 0;
    if (disc > 0) {
        double sq = sqrt(disc);
        out[n++] = cbrt(-qq / 2 + sq) + cbrt(-qq / 2 - sq) - a / 3;
    } else if (p == 0) {
        out[n++] = -a / 3;
    } else {
        double r = sqrt(-p / 3), arg = 3 * qq / (2 * p * r);
        if (arg > 1) arg = 1;
        if (arg < -1) arg = -1;
        double ph = acos(arg) / 3;
        for (int k = 0; k < 3; ++k) out[n++] = 2 * r * cos(ph - 2 * M_PI * k / 3) - a / 3;
    }
    for (int k = 0; k < n; ++k)
        for (int itn = 0; itn < 3; ++itn) {
            double t = out[k], f = ((c[0] * t + c[1]) * t + c[2]) * t + c[3], fp = (3 * c[0] * t + 2 * c[1]) * t + c[2];
            if (fp != 0) out[k] = t - f / fp;
        }
    return n;
}


/* ---- roots_derivative AS WRITTEN (:727-754): eigenvalues of the companion matrix, the first `i` of them ---------------------------
 * The reference builds the companion matrix A of the stripped polynomial (:737-744: first row -c_{j+1}/c_0, ones on the subdiagonal),
 * takes Eigen::EigenSolver<MatrixXd>(A).eigenvalues() and keeps the REAL ones among the first i = 2 entries (`j < i`, :746-751).  Which
 * two of the three roots those are is decided by the order in which Eigen's real Schur decomposition deflates them.  Eigen is an
 * un-vendored, un-pinned system dependency of the reference (CMakeLists.txt:21 `EIGEN3_INCLUDE_DIR`); what is restated here is its
 * PUBLISHED algorithm as of Eigen 3.3.x (the release line of Ubuntu 18.04 / ROS Melodic, 3.3.4): RealSchur<MatrixType>::compute =
 * scale by the largest |entry|, Hessenberg reduction (the identity on a companion matrix: its Householder vectors have zero tails),
 * computeFromHessenberg = Francis double-shift QR sweeps with deflation test |T(k,k-1)| <= eps (|T(k-1,k-1)| + |T(k,k)|), Wilkinson's /
 * MATLAB's exceptional shifts at local iterations 10 / 30, splitOffTwoRows for 2 x 2 blocks (a block with real eigenvalues is rotated
 * to upper triangular form), then EigenSolver reads the eigenvalues off the quasi-triangular T from the top (EigenSolver::compute).
 * Arithmetic of the Householder / Givens applications is written out in the order of Eigen's Householder.h / Jacobi.h; last-bit
 * differences to a particular Eigen build (vectorisation, FMA) cannot be excluded and do not matter: only the ORDER of well separated
 * roots and their values to ~1e-15 enter timeScale.  Guarded where the reference reads out of bounds: with fewer than i eigenvalues
 * (n_der < 2) it indexes past the end of es.eigenvalues() (:747); here the loop stops at n_der.
 * PARITY UNPINNED against a real Eigen (absent); pinned against an independent numpy restatement of the same published algorithm
 * (tests/golden/make_kkt_reference.py, tests/test_timescale_rule.py). */
static void es_householder(const double* v, int n, double* ess, double* tau, double* beta) { /* Householder.h makeHouseholder */
    double tail = 0;
    for (int i = 1; i < n; ++i) tail += v[i] * v[i];
    const double c0 = v[0];
    if (tail <= DBL_MIN) {
        *tau = 0, *beta = c0;
        for (int i = 1; i < n; ++i) ess[i - 1] = 0;
    } else {
        double b = sqrt(c0 * c0 + tail);
        if (c0 >= 0) b = -b;
        for (int i = 1; i < n; ++i) ess[i - 1] = v[i] / (c0 - b);
        *tau = (b - c0) / b, *beta = b;
    }
}
static void es_house_left(double T[3][3], int r0, int nr, int c0, int c1, const double* ess, double tau) { /* applyHouseholderOnTheLeft */
    if (nr == 1) {
        for (int j = c0; j <= c1; ++j) T[r0][j] *= (1 - tau);
    } else if (tau != 0) {
        for (int j = c0; j <= c1; ++j) {
            double tmp = 0;
            for (int i = 1; i < nr; ++i) tmp += ess[i - 1] * T[r0 + i][j];
            tmp += T[r0][j];
            T[r0][j] -= tau * tmp;
            for (int i = 1; i < nr; ++i) T[r0 + i][j] -= tau * ess[i - 1] * tmp;
        }
    }
}
static void es_house_right(double T[3][3], int r0, int r1, int c0, int nc, const double* ess, double tau) { /* applyHouseholderOnTheRight */
    if (nc == 1) {
        for (int i = r0; i <= r1; ++i) T[i][c0] *= (1 - tau);
    } else if (tau != 0) {
        for (int i = r0; i <= r1; ++i) {
            double tmp = 0;
            for (int j = 1; j < nc; ++j) tmp += T[i][c0 + j] * ess[j - 1];
            tmp += T[i][c0];
            T[i][c0] -= tau * tmp;
            for (int j = 1; j < nc; ++j) T[i][c0 + j] -= tau * tmp * ess[j - 1];
        }
    }
}
/* eigenvalues of the n x n (n <= 3) upper Hessenberg matrix A in EigenSolver's order; returns 0 if the QR iteration did not converge */
static int es_eigenvalues(double A[3][3], int n, double* re, double* im) {
    double T[3][3], scale = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) scale = fmax(scale, fabs(A[i][j]));
    if (scale < DBL_MIN) {
        for (int i = 0; i < n; ++i) re[i] = 0, im[i] = 0;
        return 1;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) T[i][j] = A[i][j] / scale;
    double norm = 0; /* computeNormOfT */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < (j + 2 < n ? j + 2 : n); ++i) norm += fabs(T[i][j]);
    int iu = n - 1, iter = 0, total = 0;
    double exshift = 0;
    if (norm != 0)
        while (iu >= 0) {
            int il = iu; /* findSmallSubdiagEntry */
            while (il > 0) {
                const double sd = fabs(T[il - 1][il - 1]) + fabs(T[il][il]);
                if (fabs(T[il][il - 1]) <= DBL_EPSILON * sd) break;
                il--;
            }
            if (il == iu) { /* one root */
                T[iu][iu] += exshift;
                if (iu > 0) T[iu][iu - 1] = 0;
                iu--, iter = 0;
            } else if (il == iu - 1) { /* splitOffTwoRows */
                const double p = 0.5 * (T[iu - 1][iu - 1] - T[iu][iu]);
                const double q = p * p + T[iu][iu - 1] * T[iu - 1][iu];
                T[iu][iu] += exshift, T[iu - 1][iu - 1] += exshift;
                if (q >= 0) {
                    const double z = sqrt(fabs(q));
                    const double gp = (p >= 0) ? p + z : p - z, gq = T[iu][iu - 1];
                    double c, sn; /* JacobiRotation::makeGivens (real) */
                    if (gq == 0) {
                        c = gp < 0 ? -1 : 1, sn = 0;
                    } else if (gp == 0) {
                        c = 0, sn = gq < 0 ? 1 : -1;
                    } else if (fabs(gp) > fabs(gq)) {
                        const double t = gq / gp;
                        double u = sqrt(1 + t * t);
                        if (gp < 0) u = -u;
                        c = 1 / u, sn = -t * c;
                    } else {
                        const double t = gp / gq;
                        double u = sqrt(1 + t * t);
                        if (gq < 0) u = -u;
                        sn = -1 / u, c = -t * sn;
                    }
                    for (int j = iu - 1; j < n; ++j) { /* rightCols(size-iu+1).applyOnTheLeft(iu-1, iu, rot.adjoint()) */
                        const double x = T[iu - 1][j], y = T[iu][j];
                        T[iu - 1][j] = c * x - sn * y, T[iu][j] = sn * x + c * y;
                    }
                    for (int i = 0; i <= iu; ++i) { /* topRows(iu+1).applyOnTheRight(iu-1, iu, rot) */
                        const double x = T[i][iu - 1], y = T[i][iu];
                        T[i][iu - 1] = c * x - sn * y, T[i][iu] = sn * x + c * y;
                    }
                    T[iu][iu - 1] = 0;
                }
                if (iu > 1) T[iu - 1][iu - 2] = 0;
                iu -= 2, iter = 0;
            } else { /* il < iu - 1: only n = 3, il = 0, iu = 2 */
                double sh0 = T[iu][iu], sh1 = T[iu - 1][iu - 1], sh2 = T[iu][iu - 1] * T[iu - 1][iu]; /* computeShift */
                if (iter == 10) {
                    exshift += sh0;
                    for (int i = 0; i <= iu; ++i) T[i][i] -= sh0;
                    const double sd = fabs(T[iu][iu - 1]) + fabs(T[iu - 1][iu - 2]);
                    sh0 = 0.75 * sd, sh1 = 0.75 * sd, sh2 = -0.4375 * sd * sd;
                }
                if (iter == 30) {
                    double sd = (sh1 - sh0) / 2.0;
                    sd = sd * sd + sh2;
                    if (sd > 0) {
                        sd = sqrt(sd);
                        if (sh1 < sh0) sd = -sd;
                        sd = sd + (sh1 - sh0) / 2.0;
                        sd = sh0 - sh2 / sd;
                        exshift += sd;
                        for (int i = 0; i <= iu; ++i) T[i][i] -= sd;
                        sh0 = sh1 = sh2 = 0.964;
                    }
                }
                iter++, total++;
                if (total > 40 * n) return 0;
                int imm; /* initFrancisQRStep */
                double v[3] = {0, 0, 0};
                for (imm = iu - 2; imm >= il; --imm) {
                    const double Tmm = T[imm][imm], r = sh0 - Tmm, sd = sh1 - Tmm;
                    v[0] = (r * sd - sh2) / T[imm + 1][imm] + T[imm][imm + 1];
                    v[1] = T[imm + 1][imm + 1] - Tmm - r - sd;
                    v[2] = T[imm + 2][imm + 1];
                    if (imm == il) break;
                    const double lhs = T[imm][imm - 1] * (fabs(v[1]) + fabs(v[2]));
                    const double rhs = v[0] * (fabs(T[imm - 1][imm - 1]) + fabs(Tmm) + fabs(T[imm + 1][imm + 1]));
                    if (fabs(lhs) < DBL_EPSILON * rhs) break;
                }
                for (int k = imm; k <= iu - 2; ++k) { /* performFrancisQRStep */
                    const int first = (k == imm);
                    double w[3], ess[2], tau, beta;
                    if (first)
                        w[0] = v[0], w[1] = v[1], w[2] = v[2];
                    else
                        w[0] = T[k][k - 1], w[1] = T[k + 1][k - 1], w[2] = T[k + 2][k - 1];
                    es_householder(w, 3, ess, &tau, &beta);
                    if (beta != 0) {
                        if (first && k > il)
                            T[k][k - 1] = -T[k][k - 1];
                        else if (!first)
                            T[k][k - 1] = beta;
                        es_house_left(T, k, 3, k, n - 1, ess, tau);
                        es_house_right(T, 0, (iu < k + 3 ? iu : k + 3), k, 3, ess, tau);
                    }
                }
                {
                    double w[2] = {T[iu - 1][iu - 2], T[iu][iu - 2]}, ess[1], tau, beta;
                    es_householder(w, 2, ess, &tau, &beta);
                    if (beta != 0) {
                        T[iu - 1][iu - 2] = beta;
                        es_house_left(T, iu - 1, 2, iu - 1, n - 1, ess, tau);
                        es_house_right(T, 0, iu, iu - 1, 2, ess, tau);
                    }
                }
                for (int i = imm + 2; i <= iu; ++i) { /* clean up pollution due to round-off errors */
                    T[i][i - 2] = 0;
                    if (i > imm + 2) T[i][i - 3] = 0;
                }
            }
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) T[i][j] *= scale;
    for (int i = 0; i < n; ++i) { /* EigenSolver::compute: eigenvalues from the quasi-triangular T, top to bottom */
        if (i == n - 1 || T[i + 1][i] == 0) {
            re[i] = T[i][i], im[i] = 0;
        } else {
            const double p = 0.5 * (T[i][i] - T[i + 1][i + 1]);
            double t0 = T[i + 1][i], t1 = T[i][i + 1];
            const double mx = fmax(fabs(p), fmax(fabs(t0), fabs(t1)));
            t0 /= mx, t1 /= mx;
            const double p0 = p / mx, z = mx * sqrt(fabs(p0 * p0 + t0 * t1));
            re[i] = T[i + 1][i + 1] + p, im[i] = z;
            re[i + 1] = T[i + 1][i + 1] + p, im[i + 1] = -z;
            ++i;
        }
    }
    return 1;
}
/* roots_derivative(i, coef_der) :727-754 for the polynomial c[0] t^deg + ... + c[deg] (row i of coef_der): the real ones among the first
 * `take` eigenvalues (the reference passes the derivative order i = 2 as that bound, :746) */
static int first_eigen_roots(const double* c, int deg, int take, double* out) {
    while (deg > 0 && c[0] == 0) c++, deg--; /* :729-733 */
    if (deg == 0) return 0;
    double A[3][3] = {{0}}, re[3], im[3];
    for (int j = 0; j < deg; ++j) { /* :737-744 */
        if (j < deg - 1) A[j + 1][j] = 1;
        A[0][j] = -c[j + 1] / c[0];
    }
    if (!es_eigenvalues(A, deg, re, im)) return 0;
    int n = 0;
    for (int j = 0; j < take && j < deg; ++j) /* :746-751 (guard: j < deg) */
        if (im[j] == 0) out[n++] = re[j];
    return n;
}
/* for the tests: eigenvalues in EigenSolver order of the companion matrix of c[0] t^deg + ... (deg <= 3, c[0] != 0) */
int oracle_companion_eigenvalues(const double* c, int deg, double* re, double* im) {
    if (deg < 1 || deg > 3 || c[0] == 0) return -1;
    double A[3][3] = {{0}};
    for (int j = 0; j < deg; ++j) {
        if (j < deg - 1) A[j + 1][j] = 1;
        A[0][j] = -c[j + 1] / c[0];
    }
    return es_eigenvalues(A, deg, re, im) ? deg : 0;
}

static double time_scale_by_rule(const rbp_mission* mission, const rbp_plan* plan, int rule) { /* :209-233, :708-847; rule: rbp_param.timescale_rule */
    const int N = plan->N, M = plan->M, n = 5;
    double time_scale = 1;
    for (int qi = 0; qi < N; ++qi)
        for (int k = 0; k < 3; ++k)
            for (int m = 0; m < M; ++m) {
                const double* cf = plan->coef + ((size_t)qi * 3 + k) * 6 * M + 6 * m;
                double cd[4][6]; /* derivative_segment :708-718 */
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 6; ++j) cd[i][n - j] = (i <= j) ? coef_derivative(i, j) * cf[n - j] : 0;
                const double dt = plan->T[m + 1] - plan->T[m];
                /* scale_to_max_vel :756-794 */
                {
                    double ts[8];
                    int nt = rule == RBP_TIMESCALE_FIRST_EIGENVALUES ? first_eigen_roots(cd[2], 3, 2, ts) /* roots_derivative(2, coef_der) :761 */
                                                                     : real_roots(cd[2], 3, ts); /* every velocity extremum */
                    ts[nt++] = 0, ts[nt++] = dt;
                    double vel_max = 0, t_max = 0;
                    for (int a = 0; a < nt; ++a) {
                        double t = ts[a];
                        if (t < 0 || t > dt) continue;
                        double vel = 0;
                        for (int i = 0; i <= n - 1; ++i) vel += cd[1][i] * pow(t, n - 1 - i);
                        vel = fabs(vel);
                        if (vel_max < vel) vel_max = vel, t_max = t;
                    }
                    double sc = 1;
                    while (vel_max > mission->max_vel[3 * qi + k]) {
                        sc *= 1.1;
                        double vel = 0;
                        for (int i = 0; i <= n - 1; ++i) vel += cd[1][i] * pow(1 / sc, n - i) * pow(t_max, n - 1 - i);
                        vel_max = fabs(vel);
                    }
                    if (time_scale < sc) time_scale = sc;
                }
                /* scale_to_max_acc :797-847 */
                {
                    double a = cd[3][0], b = cd[3][1], c = cd[3][2], D = b * b - 4 * a * c;
                    double ts[4] = {0, dt, 0, 0};
                    int nt = 2;
                    if (D >= 0 && a != 0) {
                        ts[nt++] = (-b + sqrt(D)) / (2 * a);
                        ts[nt++] = (-b - sqrt(D)) / (2 * a);
                    } else if (a == 0 && b != 0)
                        ts[nt++] = -c / b;
                    double acc_max = 0, t_max = 0;
                    for (int e = 0; e < nt; ++e) {
                        double t = ts[e];
                        if (t < 0 || t > dt) continue;
                        double acc = 0;
                        for (int i = 0; i < 4; ++i) acc += cd[2][i] * pow(t, 3 - i);
                        acc = fabs(acc);
                        if (acc_max < acc) acc_max = acc, t_max = t;
                    }
                    double sc = 1;
                    while (acc_max > mission->max_acc[3 * qi + k]) {
                        sc *= 1.1;
                        double acc = 0;
                        for (int i = 0; i < 4; ++i) acc += cd[2][i] * pow(1 / sc, n - i) * pow(t_max, 3 - i);
                        acc_max = fabs(acc);
                    }
                    if (time_scale < sc) time_scale = sc;
                }
            }
    return time_scale;
}

double oracle_time_scale_rule(const rbp_mission* mission, rbp_plan* plan, int rule) { /* :209-266 */
    const int N = plan->N, M = plan->M, n = 5;
    const double time_scale = time_scale_by_rule(mission, plan, rule);
    plan->time_scale_alt = time_scale_by_rule(mission, plan, rule == RBP_TIMESCALE_FIRST_EIGENVALUES ? RBP_TIMESCALE_ALL_REAL_ROOTS : RBP_TIMESCALE_FIRST_EIGENVALUES);
    if (time_scale != 1) { /* :236-265 */
        for (int qi = 0; qi < N; ++qi) {
            for (int k = 0; k < 3; ++k)
                for (int m = 0; m < M; ++m) {
                    double* cf = plan->coef + ((size_t)qi * 3 + k) * 6 * M + 6 * m;
                    for (int i = 0; i < 6; ++i) cf[i] = pow(1.0 / time_scale, n - i) * cf[i];
                }
            for (int b = 0; b < plan->sfc_count[qi]; ++b) plan->sfc_time[(size_t)qi * plan->max_boxes + b] *= time_scale;
        }
        for (int m = 0; m < M; ++m) plan->rsfc_time[m] *= time_scale;
        for (int m = 0; m <= M; ++m) plan->T[m] *= time_scale;
    }
    return time_scale;
}
double oracle_time_scale(const rbp_mission* mission, rbp_plan* plan) { return oracle_time_scale_rule(mission, plan, RBP_TIMESCALE_ALL_REAL_ROOTS); }

/* ------------------------------------------------------------------------------------------------
 * solver-independent evaluation of a candidate set of control points
 * ---------------------------------------------------------------------------------------------- */
int oracle_evaluate_ctrl(const rbp_mission* mission, const rbp_plan* plan, const double* ctrl, double* objective,
                         double* viol_eq, double* viol_box, double* viol_rsfc) {
    const int N = plan->N, M = plan->M, oq = 6 * M, neb = 3 * (M + 1);
    double Qb[36];
    oracle_build_Q_base(Qb, NULL);
    double* Aeq = (double*)malloc(sizeof(double) * neb * oq);
    oracle_build_Aeq_base(M, plan->T, Aeq);
    int* sel = (int*)malloc(sizeof(int) * M);
    double obj = 0, veq = 0, vbox = 0, vr = 0;
    for (int qi = 0; qi < N; ++qi) {
        select_boxes(plan, qi, sel);
        for (int k = 0; k < 3; ++k) {
            const double* c = ctrl + ((size_t)qi * 3 + k) * oq;
            for (int m = 0; m < M; ++m) {
                double s = pow(plan->T[m + 1] - plan->T[m], -5);
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) obj += Qb[6 * i + j] * s * c[6 * m + i] * c[6 * m + j];
                const double* box = plan->sfc_box + ((size_t)qi * plan->max_boxes + sel[m]) * 6;
                for (int i = 0; i < 6; ++i) {
                    vbox = fmax(vbox, c[6 * m + i] - box[3 + k]);
                    vbox = fmax(vbox, box[k] - c[6 * m + i]);
                }
            }
            for (int e = 0; e < neb; ++e) {
                double s = 0;
                for (int j = 0; j < oq; ++j) s += Aeq[(size_t)e * oq + j] * c[j];
                double d = 0;
                if (e < 3) d = mission->start[9 * qi + k + 3 * e];
                else if (e < 6) d = mission->goal[9 * qi + k + 3 * (e - 3)];
                veq = fmax(veq, fabs(s - d));
            }
        }
    }
    for (int qi = 0; qi < N; ++qi)
        for (int qj = qi + 1; qj < N; ++qj) {
            const float* normals = plan->rsfc_normal + pair_index(N, qi, qj) * M * 3;
            double rr = mission->radius[qi] + mission->radius[qj];
            for (int j = 0; j < oq; ++j) {
                const float* nv = normals + 3 * select_rsfc(plan, j / 6);
                double s = 0;
                for (int k = 0; k < 3; ++k)
                    s += (double)nv[k] * (ctrl[((size_t)qj * 3 + k) * oq + j] - ctrl[((size_t)qi * 3 + k) * oq + j]);
                vr = fmax(vr, rr - s);
            }
        }
    free(Aeq), free(sel);
    if (objective) *objective = obj;
    if (viol_eq) *viol_eq = veq;
    if (viol_box) *viol_box = vbox;
    if (viol_rsfc) *viol_rsfc = vr;
    return RBP_OK;
}

/* ------------------------------------------------------------------------------------------------
 * RBPPlanner::update
 * ---------------------------------------------------------------------------------------------- */
int oracle_planner_update(const rbp_mission* mission, const rbp_param* param_in, rbp_plan* plan,
                          const oracle_qp_options* opt_in, oracle_qp_report* report) {
    if (!mission || !param_in || !plan || mission->N != plan->N) return RBP_ERR_BAD_ARGUMENT;
    rbp_param param = *param_in;
    if (param.n != 5 || param.phi != 3) return RBP_ERR_UNSUPPORTED_DEGREE; /* :344-346 (reference logs and goes on) */
    oracle_qp_options opt;
    if (opt_in) opt = *opt_in; else oracle_qp_default_options(&opt);
    oracle_qp_report rep_local;
    if (!report) report = &rep_local;
    memset(report, 0, sizeof(*report));
    report->kkt_dual_min = 1e300;
    const int N = plan->N, M = plan->M, oq = 6 * M;

    /* setBatch :849-872 */
    int batch_max_iter = (int)ceil((double)N / (double)param.batch_size);
    if (param.sequential) {
        if (param.batch_iter < 0 || param.batch_iter > batch_max_iter) param.batch_iter = batch_max_iter;
    } else {
        param.batch_size = N;
        param.batch_iter = 1;
    }
    /* buildConstMtx :100-109 */
    double* ctrl = (double*)calloc((size_t)N * 3 * oq, sizeof(double)); /* `dummy`, then solved values */
    if (param.sequential) oracle_build_dummy(N, M, plan->init_traj, ctrl);
    /* solver warm start (not part of the reference's semantics): dummy-style control points for every mode;
     * in sequential mode this IS `dummy` and follows the Gauss-Seidel updates */
    double* warm = ctrl;
    if (!param.sequential) {
        warm = (double*)calloc((size_t)N * 3 * oq, sizeof(double));
        oracle_build_dummy(N, M, plan->init_traj, warm);
    }

    int rc = RBP_OK;
    double total_cost = 0;
    int* batch = (int*)malloc(sizeof(int) * N);
    if (!(param.sequential && param.batch_iter == 0)) { /* :119-138 otherwise: coef straight from dummy */
        for (int iter = 0; iter < param.iteration && rc == RBP_OK; ++iter) {
            total_cost = 0;
            for (int l = 0; l < param.batch_iter && rc == RBP_OK; ++l) {
                int nb = 0;
                for (int qi = 0; qi < N; ++qi)
                    if (qi / param.batch_size == l) batch[nb++] = qi;
                qp_t q;
                qp_build(mission, plan, batch, nb, ctrl, &q);
                plan->x_size = q.nx, plan->eq_size = q.ne, plan->ineq_size = q.nc_full;
                qp_sol_t sol;
                memset(&sol, 0, sizeof(sol));
                double* x0 = (double*)malloc(sizeof(double) * q.nx);
                for (int k = 0; k < 3; ++k)
                    for (int b = 0; b < nb; ++b)
                        memcpy(x0 + (size_t)k * nb * oq + (size_t)b * oq, warm + ((size_t)batch[b] * 3 + k) * oq, sizeof(double) * oq);
                rc = q.infeasible ? RBP_ERR_QP_FAILED : qp_solve(&q, x0, &opt, &sol, report);
                free(x0);
                if (rc == RBP_OK) {
                    total_cost += sol.obj; /* cplex.getObjValue() :164 */
                    for (int k = 0; k < 3; ++k)
                        for (int b = 0; b < nb; ++b)
                            memcpy(ctrl + ((size_t)batch[b] * 3 + k) * oq, sol.x + (size_t)k * nb * oq + (size_t)b * oq,
                                   sizeof(double) * oq); /* vals -> dummy :181-184 */
                }
                plan->qp_iterations = report->iters_total;
                qp_sol_free(&sol);
                qp_free(&q);
            }
        }
    }
    free(batch);
    if (rc == RBP_OK) {
        oracle_ctrl_to_coef(N, M, plan->T, ctrl, plan->coef);
        if (plan->ctrl) memcpy(plan->ctrl, ctrl, sizeof(double) * (size_t)N * 3 * oq);
        plan->total_cost = total_cost;
        plan->time_scale = 1;
        plan->time_scale_alt = 1;
        if (param.time_scale) plan->time_scale = oracle_time_scale_rule(mission, plan, param.timescale_rule); /* :72-77 */
    }
    if (warm != ctrl) free(warm);
    free(ctrl);
    return rc;
}
