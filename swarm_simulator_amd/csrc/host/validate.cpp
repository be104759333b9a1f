// Validation metrics and the crazyswarm CSV writer — the reference's own acceptance signals.
//   sampler            rbp_publisher.hpp:50-54 (t = i*0.1, floor(T.back()/dt) samples), :169-194 (timeMatrix), :670-683
//   flight distance    rbp_publisher.hpp:685-695
//   safety ratio       rbp_publisher.hpp:769-798   (downwash-scaled distance / (r_i + r_j); should be >= 1)
//   coef CSV           rbp_planner.hpp:295-324
#include "rbp_host.h"

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

extern "C" int rbp_validate(const rbp_mission* mission, const rbp_param* param, int32_t M, const double* T,
                            const double* coef, double dt, double* min_safety_ratio, double* total_flight_distance) {
    if (!mission || !param || !T || !coef || M <= 0 || dt <= 0) return RBP_ERR_BAD_ARGUMENT;
    const int N = mission->N, n = 5;
    const int nt = (int)std::floor(T[M] / dt);
    std::vector<double> pos((size_t)N * nt * 3);
    for (int j = 0; j < nt; ++j) {
        double t = j * dt;
        // rbp_publisher.hpp:173-183: last segment whose start time is strictly before t (segment 0 at t=0)
        int index = 0;
        double tseg = 0;
        for (int m = 0; m < M; ++m) {
            if (T[m] < t) {
                tseg = T[m];
                index = m;
            } else
                break;
        }
        tseg = t - tseg;
        for (int qi = 0; qi < N; ++qi)
            for (int k = 0; k < 3; ++k) {
                const double* c = coef + ((size_t)qi * 3 + k) * 6 * M + 6 * index;  // descending powers
                double v = 0;
                for (int i = 0; i <= n; ++i) v += c[i] * std::pow(tseg, n - i);
                pos[((size_t)qi * nt + j) * 3 + k] = v;
            }
    }
    double len = 0;
    for (int qi = 0; qi < N; ++qi)
        for (int j = 0; j + 1 < nt; ++j) {
            const double* a = &pos[((size_t)qi * nt + j) * 3];
            const double* b = a + 3;
            len += std::sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]) + (b[2] - a[2]) * (b[2] - a[2]));
        }
    double ratio_min = 1e9;  // SP_INFINITY
    for (int j = 0; j < nt; ++j)
        for (int qi = 0; qi < N; ++qi)
            for (int qj = qi + 1; qj < N; ++qj) {
                const double* a = &pos[((size_t)qi * nt + j) * 3];
                const double* b = &pos[((size_t)qj * nt + j) * 3];
                double dz = (a[2] - b[2]) / param->downwash;
                double r = std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + dz * dz) /
                           (mission->radius[qi] + mission->radius[qj]);
                if (r < ratio_min) ratio_min = r;
            }
    if (min_safety_ratio) *min_safety_ratio = ratio_min;
    if (total_flight_distance) *total_flight_distance = len;
    return RBP_OK;
}

extern "C" int rbp_write_coef_csv(const char* dir, int32_t N, int32_t M, const double* T, const double* coef) {
    if (!dir || !T || !coef) return RBP_ERR_BAD_ARGUMENT;
    const int n = 5;
    for (int qi = 0; qi < N; ++qi) {
        std::string path = std::string(dir) + "/coef" + std::to_string(qi + 1) + ".csv";
        FILE* f = fopen(path.c_str(), "w");
        if (!f) return RBP_ERR_BAD_ARGUMENT;
        fprintf(f, "duration");
        for (const char* ax : {"x", "y", "z", "yaw"})
            for (int i = 0; i < 8; ++i) fprintf(f, ",%s^%d", ax, i);
        fprintf(f, "\n");
        for (int m = 0; m < M; ++m) {
            fprintf(f, "%g,", T[m + 1] - T[m]);
            for (int k = 0; k < 3; ++k) {
                const double* c = coef + ((size_t)qi * 3 + k) * 6 * M + 6 * m;
                for (int i = n; i >= 0; --i) fprintf(f, "%g,", c[i]);  // ascending powers
                for (int i = 0; i < 7 - n; ++i) fprintf(f, "0,");
            }
            for (int i = 0; i < 8; ++i) fprintf(f, "0,");
            fprintf(f, "\n");
        }
        fclose(f);
    }
    return RBP_OK;
}
