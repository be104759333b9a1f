// qp_lp.cpp — the QP of one batch as a CPLEX LP-format file.
//
// The reference dumps the Concert model of every batch with cplex.exportModel(".../log/QPmodel.lp") when `log` is set
// (swarm_planner/include/rbp_planner.hpp:150-152).  This writer emits the same model -- variables named and ordered as in
// populatebyrow (x_/y_/z_<qi>_<m>_<i>, :552-577), objective sum Q_p(i,j) x_i x_j without 1/2 (:582-605), the equality rows of
// Aeq_base / deq (:608-622), two SFC rows per variable (:626-635) and one RSFC row per pair and control point with the frozen
// agent replaced by `dummy` (:638-684) -- from the flat arrays of include/rbp.h.  Host side, debugging aid, not accelerated.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "rbp_host.h"

namespace {

const double kQ[36] = {720,  -1800, 1200,  0,     0,     -120, -1800, 4800,  -3600, 0,     600,   0,
                       1200, -3600, 3600,  -1200, 0,     0,    0,     0,     -1200, 3600,  -3600, 1200,
                       0,    600,   0,     -3600, 4800,  -1800, -120, 0,     0,     1200,  -1800, 720};  // :330-335
const double kA0[3][6] = {{1, 0, 0, 0, 0, 0}, {-1, 1, 0, 0, 0, 0}, {1, -2, 1, 0, 0, 0}};                 // :362-367 rows 0..2
const double kAT[3][6] = {{0, 0, 0, 0, 0, 1}, {0, 0, 0, 0, -1, 1}, {0, 0, 0, 1, -2, 1}};                 // :369-374 rows 0..2

std::string var(int k, int qi, int m, int i) {
    char b[64];
    snprintf(b, sizeof b, "%c_%d_%d_%d", "xyz"[k], qi, m, i);
    return b;
}

void term(FILE* f, double c, const std::string& v, bool first) {
    if (c < 0)
        fprintf(f, first ? " - %.17g %s" : " - %.17g %s", -c, v.c_str());
    else
        fprintf(f, first ? " %.17g %s" : " + %.17g %s", c, v.c_str());
}

}  // namespace

extern "C" int rbp_write_qp_lp(const char* path, const rbp_mission* ms, const rbp_param* pr, const rbp_plan* pl, int32_t l, const double* dummy) {
    if (!path || !ms || !pr || !pl || !pl->T || !pl->sfc_count || !pl->sfc_box || !pl->sfc_time || !pl->rsfc_normal || !pl->rsfc_time || !pl->init_traj)
        return RBP_ERR_BAD_ARGUMENT;
    const int N = pl->N, M = pl->M, oq = 6 * M, MB = pl->max_boxes;
    // setBatch (:849-872)
    int bs = pr->sequential ? pr->batch_size : N;
    if (bs <= 0) bs = 1;
    if (bs > N) bs = N;
    const int bmax = (N + bs - 1) / bs;
    if (l < 0 || l >= bmax) return RBP_ERR_BAD_ARGUMENT;
    const int first = l * bs, nb = std::min(bs, N - first);
    const double* T = pl->T;
    // dummy (:513-549) unless the caller hands in the current control points ([N][3][6M], the layout of rbp_plan.ctrl)
    std::vector<double> dm;
    if (!dummy) {
        dm.resize((size_t)N * 3 * oq);
        for (int qi = 0; qi < N; ++qi)
            for (int k = 0; k < 3; ++k)
                for (int m = 0; m < M; ++m)
                    for (int j = 0; j < 6; ++j)
                        dm[((size_t)qi * 3 + k) * oq + 6 * m + j] = (double)pl->init_traj[((size_t)qi * (M + 1) + (j < 3 ? m : m + 1)) * 3 + k];
        dummy = dm.data();
    }
    FILE* f = fopen(path, "w");
    if (!f) return RBP_ERR_BAD_ARGUMENT;
    fprintf(f, "\\ENCODING=ISO-8859-1\n\\Problem name: RBP batch %d (agents %d..%d), variables in the order of rbp_planner.hpp:552-577\n\nMinimize\n obj: [", l, first,
            first + nb - 1);
    // objective (:582-605): sum_{i,j} Q_p(i,j) x_i x_j ; inside "[ ... ] / 2" the coefficients are doubled
    for (int k = 0; k < 3; ++k)
        for (int bi = 0; bi < nb; ++bi)
            for (int m = 0; m < M; ++m) {
                const double sc = std::pow(T[m + 1] - T[m], -5.0);  // build_Q_p :349-351
                for (int i = 0; i < 6; ++i)
                    for (int j = i; j < 6; ++j) {
                        const double q = kQ[6 * i + j] * sc;
                        if (q == 0) continue;
                        const std::string vi = var(k, first + bi, m, i), vj = var(k, first + bi, m, j);
                        if (i == j)
                            fprintf(f, "%s %.17g %s ^2", q < 0 ? " -" : " +", 2 * std::fabs(q), vi.c_str());
                        else
                            fprintf(f, "%s %.17g %s * %s", q < 0 ? " -" : " +", 4 * std::fabs(q), vi.c_str(), vj.c_str());
                    }
                fprintf(f, "\n  ");
            }
    fprintf(f, "] / 2\nSubject To\n");
    int cid = 1;
    // equalities (:608-622): per (k, bi) the rows of Aeq_base (:353-405) with deq (:408-432)
    for (int k = 0; k < 3; ++k)
        for (int bi = 0; bi < nb; ++bi) {
            const int qi = first + bi;
            for (int r = 0; r < 3 * (M + 1); ++r) {
                fprintf(f, " c%d:", cid++);
                double rhs = 0;
                bool fst = true;
                const double nn[3] = {1, 5, 20};
                if (r < 3) {  // start state
                    for (int c = 0; c < 6; ++c) {
                        const double v = std::pow(T[1] - T[0], -r) * nn[r] * kA0[r][c];
                        if (v != 0) term(f, v, var(k, qi, 0, c), fst), fst = false;
                    }
                    rhs = ms->start[(size_t)qi * 9 + k + 3 * r];
                } else if (r < 6) {  // goal state
                    const int i = r - 3;
                    for (int c = 0; c < 6; ++c) {
                        const double v = std::pow(T[M] - T[M - 1], -i) * nn[i] * kAT[i][c];
                        if (v != 0) term(f, v, var(k, qi, M - 1, c), fst), fst = false;
                    }
                    rhs = ms->goal[(size_t)qi * 9 + k + 3 * i];
                } else {  // continuity at knot m = 1 .. M-1
                    const int m = (r - 6) / 3 + 1, j = (r - 6) % 3;
                    for (int c = 0; c < 6; ++c) {
                        const double v = std::pow(T[m] - T[m - 1], -j) * nn[j] * kAT[j][c];
                        if (v != 0) term(f, v, var(k, qi, m - 1, c), fst), fst = false;
                    }
                    for (int c = 0; c < 6; ++c) {
                        const double v = -std::pow(T[m + 1] - T[m], -j) * nn[j] * kA0[j][c];
                        if (v != 0) term(f, v, var(k, qi, m, c), fst), fst = false;
                    }
                }
                fprintf(f, " = %.17g\n", rhs);
            }
        }
    // SFC (:626-635): box of segment m = first box whose end time is not before T[m+1] (:447-469)
    for (int k = 0; k < 3; ++k)
        for (int bi = 0; bi < nb; ++bi) {
            const int qi = first + bi;
            int bx = 0;
            for (int m = 0; m < M; ++m) {
                while (bx < pl->sfc_count[qi] && pl->sfc_time[(size_t)qi * MB + bx] < T[m + 1]) bx++;
                const int sel = bx < pl->sfc_count[qi] ? bx : pl->sfc_count[qi] - 1;
                const double* box = pl->sfc_box + ((size_t)qi * MB + sel) * 6;
                for (int i = 0; i < 6; ++i) {
                    fprintf(f, " c%d: %s <= %.17g\n", cid++, var(k, qi, m, i).c_str(), box[3 + k]);
                    fprintf(f, " c%d: - %s <= %.17g\n", cid++, var(k, qi, m, i).c_str(), -box[k]);
                }
            }
        }
    // RSFC (:638-684)
    for (int qi = 0; qi < N; ++qi)
        for (int qj = qi + 1; qj < N; ++qj) {
            const bool ini = qi >= first && qi < first + nb, inj = qj >= first && qj < first + nb;
            if (!ini && !inj) continue;
            const size_t pair = (size_t)qi * N - (size_t)qi * (qi + 1) / 2 + (qj - qi - 1);
            const double rr = ms->radius[qi] + ms->radius[qj];
            for (int m = 0; m < M; ++m) {
                int ri = 0;
                while (ri < M && pl->rsfc_time[ri] < T[m + 1]) ri++;  // :487-491
                if (ri >= M) ri = M - 1;
                const float* nv = pl->rsfc_normal + (pair * M + ri) * 3;
                for (int i = 0; i < 6; ++i) {
                    const int j6 = 6 * m + i;
                    fprintf(f, " c%d:", cid++);
                    double rhs = rr;
                    bool fst = true;
                    for (int k = 0; k < 3; ++k) {
                        const double n = (double)nv[k];
                        // n . (p_j - p_i) >= rr ; a frozen agent's control point is the constant `dummy`
                        if (inj) {
                            if (n != 0) term(f, n, var(k, qj, m, i), fst), fst = false;
                        } else {
                            rhs -= n * dummy[((size_t)qj * 3 + k) * oq + j6];
                        }
                        if (ini) {
                            if (n != 0) term(f, -n, var(k, qi, m, i), fst), fst = false;
                        } else {
                            rhs += n * dummy[((size_t)qi * 3 + k) * oq + j6];
                        }
                    }
                    if (fst) fprintf(f, " 0 %s", var(0, ini ? qi : qj, m, i).c_str());  // a zero normal leaves an empty row
                    fprintf(f, " >= %.17g\n", rhs);
                }
            }
        }
    fprintf(f, "Bounds\n");
    for (int k = 0; k < 3; ++k)
        for (int bi = 0; bi < nb; ++bi)
            for (int m = 0; m < M; ++m)
                for (int i = 0; i < 6; ++i) fprintf(f, " %s free\n", var(k, first + bi, m, i).c_str());
    fprintf(f, "End\n");
    fclose(f);
    return RBP_OK;
}
