// adapter_main.cpp -- drives the C++ adapter of INTEGRATION.md (extracted from the markdown by tests/test_abi_c.py into
// rbp_hip_adapter.hpp) exactly as the reference's call site does (swarm_planner/src/swarm_traj_planner_rbp.cpp:96-116), on objects of the
// reference's own shapes (mock headers under tests/abi_c/mock), and writes what ends up in PlanResult to the same flat result file as
// smoke.c -- so the positional initialisers, the layouts and both conversions of the adapter are seen by a compiler AND checked
// against the golden vectors.  Test infrastructure.  usage: adapter_main <case.flat> <result.flat>; 30 = no device.
#include "rbp_hip_adapter.hpp"

extern "C" {
#include "flatfile.h"
}

using namespace SwarmPlanning;

int main(int argc, char** argv) {
    flat_case c;
    if (argc < 3 || flat_case_read(argv[1], &c) != 0) return 91;
    const int N = c.N, M = c.M;
    Mission mission;
    mission.qn = N;
    for (int q = 0; q < N; ++q) {
        mission.startState.emplace_back(c.start + 9 * q, c.start + 9 * q + 9);
        mission.goalState.emplace_back(c.goal + 9 * q, c.goal + 9 * q + 9);
        mission.max_vel.emplace_back(c.max_vel + 3 * q, c.max_vel + 3 * q + 3);
        mission.max_acc.emplace_back(c.max_acc + 3 * q, c.max_acc + 3 * q + 3);
        mission.quad_size.push_back(c.radius[q]);
        mission.quad_speed.push_back(1.0);
    }
    Param param;
    const double* v = c.param;
    param.world_x_min = v[0], param.world_y_min = v[1], param.world_z_min = v[2];
    param.world_x_max = v[3], param.world_y_max = v[4], param.world_z_max = v[5];
    param.box_xy_res = v[6], param.box_z_res = v[7], param.downwash = v[8], param.time_step = v[9];
    param.ecbs_w = v[10], param.grid_xy_res = v[11], param.grid_z_res = v[12], param.grid_margin = v[13];
    param.n = (int)v[14], param.phi = (int)v[15], param.sequential = v[16] != 0, param.batch_size = (int)v[17];
    param.batch_iter = (int)v[18], param.iteration = (int)v[19], param.time_scale = v[20] != 0, param.log = v[21] != 0;

    const size_t cells = (size_t)c.dim[0] * c.dim[1] * c.dim[2];
    std::shared_ptr<DynamicEDTOctomap> distmap_obj(new DynamicEDTOctomap(c.dim, c.key_min, c.res, std::vector<float>(c.dist, c.dist + cells)));

    PlanResult planResult;  // what ECBSPlanner::update leaves behind: initTraj and T (ecbs_planner.hpp:34-70)
    planResult.T.assign(c.T, c.T + M + 1);
    planResult.initTraj.resize(N);
    for (int q = 0; q < N; ++q)
        for (int s = 0; s <= M; ++s) {
            const float* p = c.init_traj + ((size_t)q * (M + 1) + s) * 3;
            planResult.initTraj[q].emplace_back(p[0], p[1], p[2]);
        }

    // ---- the reference's call site, verbatim in shape ----
    std::shared_ptr<Corridor> corridor_obj;
    std::shared_ptr<RBPPlanner> RBPPlanner_obj;
    corridor_obj.reset(new Corridor(distmap_obj, mission, param, c.res));
    const bool ok_c = corridor_obj.get()->update(param.log, &planResult);
    if (!ok_c && rbp_device_count() < 1) return RBP_ERR_NO_DEVICE;
    bool ok_p = false;
    if (ok_c) {
        RBPPlanner_obj.reset(new RBPPlanner(mission, param));
        ok_p = RBPPlanner_obj.get()->update(param.log, &planResult);
    }

    // ---- PlanResult -> flat result file ----
    int MB = c.max_boxes;
    std::vector<int32_t> sfc_count(N, 0);
    std::vector<double> sfc_box((size_t)N * MB * 6, 0), sfc_time((size_t)N * MB, 0), rsfc_time(M, 0), coef((size_t)N * 18 * M, 0), ctrl;
    std::vector<float> normals((size_t)N * (N - 1) / 2 * M * 3, 0);
    if (ok_c) {
        for (int q = 0; q < N; ++q) {
            sfc_count[q] = (int32_t)planResult.SFC[q].size();
            for (int b = 0; b < sfc_count[q] && b < MB; ++b) {
                for (int k = 0; k < 6; ++k) sfc_box[((size_t)q * MB + b) * 6 + k] = planResult.SFC[q][b].first[k];
                sfc_time[(size_t)q * MB + b] = planResult.SFC[q][b].second;
            }
        }
        size_t pair = 0;
        for (int qi = 0; qi < N; ++qi)
            for (int qj = qi + 1; qj < N; ++qj, ++pair)
                for (int s = 0; s < M; ++s) {
                    const auto& e = planResult.RSFC[qi][qj][s];
                    float* n = &normals[(pair * M + s) * 3];
                    n[0] = e.first.x(), n[1] = e.first.y(), n[2] = e.first.z();
                    rsfc_time[s] = e.second;
                }
    }
    if (ok_p) {
        if ((int)planResult.msgs_traj_info.data.size() != 2 + M + 1 || planResult.msgs_traj_info.data[0] != N || planResult.msgs_traj_info.data[1] != 5) return 94;
        for (int q = 0; q < N; ++q) {
            const auto& msg = planResult.msgs_traj_coef[q];
            if (msg.layout.dim.size() != 2 || msg.layout.dim[0].size != (uint32_t)(6 * M) || msg.layout.dim[1].size != 3 || msg.data.size() != (size_t)18 * M) return 95;
            std::copy(msg.data.begin(), msg.data.end(), coef.begin() + (size_t)q * 18 * M);
        }
    }
    ctrl.assign(coef.size(), 0.0);  // the reference's PlanResult does not carry the control points
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 93;
    int32_t head[8] = {FLAT_MAGIC, ok_c ? 0 : 1, ok_c ? (ok_p ? 0 : 1) : -1, N, M, MB, 0, 0};
    fwrite(head, sizeof(int32_t), 8, f);
    fwrite(sfc_count.data(), sizeof(int32_t), N, f);
    fwrite(sfc_box.data(), sizeof(double), sfc_box.size(), f);
    fwrite(sfc_time.data(), sizeof(double), sfc_time.size(), f);
    fwrite(normals.data(), sizeof(float), normals.size(), f);
    fwrite(rsfc_time.data(), sizeof(double), M, f);
    fwrite(coef.data(), sizeof(double), coef.size(), f);
    fwrite(ctrl.data(), sizeof(double), ctrl.size(), f);
    fwrite(planResult.T.data(), sizeof(double), M + 1, f);
    const double ts = planResult.T[1] / c.T[1], zero = 0;  // T was rescaled in place by time_scale (rbp_planner.hpp:262-264)
    fwrite(&ts, sizeof(double), 1, f);
    fwrite(&zero, sizeof(double), 1, f);
    fclose(f);
    flat_case_free(&c);
    return 0;
}
