// c4_joint_rank.cpp -- lib/rbp_c4_joint_rank: ONE rank of BASELINE config C4's joint solve with NO Python and NO torch in the process.
// The 256-agent mission as one joint QP (plan/sequential = false, the reference's code default param.hpp:67); with nranks = 2 the rank
// shares the solve's knot elimination with its peer through the STREAM-ORDERED RCCL exchange (rbp_session_shard_joint_stream +
// rbp_rccl_exchange_stream: grouped ncclSend / ncclRecv enqueued on the run's stream, no host synchronisation per exchange).  Everything goes
// through the C ABIs of this repository only: include/rbp_host.h (mission JSON, octomap .bt, distance grid, ECBS), include/rbp.h, include/rbp_rccl.h
// -- what a C++ node such as the reference's swarm_traj_planner_rbp.cpp:96-116 would link.
//   usage: rbp_c4_joint_rank <rank> <nranks 1|2> <device> <id file> <data dir> [steps = 1] [mission json] [world bt]
// Rank 0 writes the communicator's 128-byte unique id to <id file> (tmp + rename), rank 1 waits for it.  Prints one JSON line.
// `bench.py --config c4 --joint --native-pair` launches one of these per rank.  nranks = 1 (the whole solve on this rank) is what a one-GPU
// box can run: tests/test_gpu_rccl_pair.py compares its answer with the Python path's.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rbp.h"
#include "rbp_host.h"
#include "rbp_rccl.h"

static int die(const char* what, int rc) {
    std::fprintf(stderr, "rbp_c4_joint_rank: %s (rc %d): %s\n", what, rc, rbp_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 6) return std::fprintf(stderr, "usage: %s rank nranks device idfile datadir [steps] [mission] [world]\n", argv[0]), 2;
    const int rank = std::atoi(argv[1]), nranks = std::atoi(argv[2]), device = std::atoi(argv[3]);
    const std::string idfile = argv[4], data = argv[5];
    const int steps = argc > 6 ? std::atoi(argv[6]) : 1;
    const std::string mfile = data + "/missions/" + (argc > 7 ? argv[7] : "mission_256agents_c4.json");
    const std::string wfile = data + "/worlds/" + (argc > 8 ? argv[8] : "map1.bt");
    if (nranks < 1 || nranks > 2 || rank < 0 || rank >= nranks) return std::fprintf(stderr, "rank / nranks out of range\n"), 2;

    // launch/plan_rbp_test.launch:27-59 with plan/sequential = false; the 256-agent mission flies in x in [-5, 15] (tools/make_mission_256.py)
    rbp_param p;
    rbp_param_defaults(&p);
    p.world_min[0] = -5, p.world_min[1] = -5, p.world_min[2] = 0.3, p.world_max[0] = argc > 7 ? 5 : 15, p.world_max[1] = 5, p.world_max[2] = 2.5;
    p.ecbs_w = 1.5, p.grid_xy_res = 0.5, p.grid_z_res = 1.0, p.grid_margin = 0.2;
    p.sequential = 0, p.batch_size = 4, p.batch_iter = -1, p.iteration = 1;

    rbp_mission_buf mb;
    if (int rc = rbp_mission_load_json(mfile.c_str(), &mb)) return die(mfile.c_str(), rc);
    rbp_octomap_buf ob;
    if (int rc = rbp_octomap_load_bt(wfile.c_str(), &ob)) return die(wfile.c_str(), rc);
    rbp_world_buf wb;
    if (int rc = rbp_world_build(&ob, p.world_min, p.world_max, 1.0, &wb)) return die("rbp_world_build", rc);
    rbp_mission m = {mb.N, mb.start, mb.goal, mb.radius, mb.max_vel, mb.max_acc};
    rbp_init_traj_buf it;
    if (int rc = rbp_ecbs_plan(&wb, &m, &p, 200000, &it)) return die("rbp_ecbs_plan", rc);
    const int N = it.N, M = it.M;
    const size_t npair = (size_t)N * (N - 1) / 2;
    std::vector<int32_t> sfc_count(N);
    std::vector<double> T(it.T, it.T + M + 1), sfc_box((size_t)N * M * 6), sfc_time((size_t)N * M), rsfc_time(M), coef((size_t)N * 18 * M), ctrl((size_t)N * 18 * M);
    std::vector<float> normals(npair * M * 3);
    rbp_world w = {{wb.dim[0], wb.dim[1], wb.dim[2]}, {wb.key_min[0], wb.key_min[1], wb.key_min[2]}, wb.res, wb.dist};
    rbp_plan pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.N = N, pl.M = M, pl.T = T.data(), pl.init_traj = it.init_traj, pl.max_boxes = M, pl.sfc_count = sfc_count.data(), pl.sfc_box = sfc_box.data();
    pl.sfc_time = sfc_time.data(), pl.rsfc_normal = normals.data(), pl.rsfc_time = rsfc_time.data(), pl.coef = coef.data(), pl.ctrl = ctrl.data();

    rbp_session* s = nullptr;
    if (int rc = rbp_session_create(&s, device, 1, &w, &m, &p, &pl)) return die("rbp_session_create", rc);
    rbp_rccl_pair* pair = nullptr;
    if (nranks == 2) {
        unsigned char id[RBP_RCCL_ID_BYTES];
        if (rank == 0) {
            if (rbp_rccl_unique_id(id)) return std::fprintf(stderr, "unique id: %s\n", rbp_rccl_last_error()), 1;
            const std::string tmp = idfile + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) return std::fprintf(stderr, "cannot write %s\n", tmp.c_str()), 1;
            std::fclose(f);
            if (std::rename(tmp.c_str(), idfile.c_str())) return std::fprintf(stderr, "cannot rename to %s\n", idfile.c_str()), 1;
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 6000 && !(f = std::fopen(idfile.c_str(), "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(50));
            if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) return std::fprintf(stderr, "no unique id in %s\n", idfile.c_str()), 1;
            std::fclose(f);
        }
        if (rbp_rccl_pair_create(&pair, device, rank, 2, id)) return std::fprintf(stderr, "pair: %s\n", rbp_rccl_last_error()), 1;
        if (int rc = rbp_session_shard_joint_stream(s, rank, 2, rbp_rccl_exchange_stream, rbp_rccl_abort, pair, 300.0)) return die("rbp_session_shard_joint_stream", rc);
    }
    if (int rc = rbp_session_run(s, RBP_STAGE_CORRIDOR, nullptr)) return die("corridor", rc);
    if (int rc = rbp_session_reserve_workspace(s, nullptr)) return die("workspace", rc);
    double best = 1e300, total = 0;
    for (int k = 0; k <= steps; ++k) {  // (step 0: warm-up)
        if (k > 0 && rbp_session_reset(s, nullptr)) return die("reset", 0);
        if (k > 0 && rbp_session_run(s, RBP_STAGE_CORRIDOR, nullptr)) return die("corridor", 0);
        const auto t0 = std::chrono::steady_clock::now();
        if (int rc = rbp_session_run(s, RBP_STAGE_PLANNER, nullptr)) return die("planner", rc);
        int32_t st = 0;
        if (int rc = rbp_session_download(s, &pl, &st, nullptr)) return die("download", rc);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (k > 0) best = dt < best ? dt : best, total += dt;
    }
    double csum = 0;
    for (double v : ctrl) csum += v * v;
    std::printf("{\"rank\": %d, \"nranks\": %d, \"agents\": %d, \"segments\": %d, \"steps\": %d, \"ms_per_step\": %.3f, \"best_ms\": %.3f, \"qp_iterations\": %d, "
                "\"qp_unpolished\": %d, \"kkt_max\": %.3e, \"total_cost\": %.12f, \"ctrl_sumsq\": %.12f, \"exchange\": \"%s\"}\n",
                rank, nranks, N, M, steps, 1e3 * total / (steps > 0 ? steps : 1), 1e3 * best, pl.qp_iterations, pl.qp_unpolished, pl.kkt_max, pl.total_cost, csum,
                nranks == 2 ? "stream-ordered ncclSend / ncclRecv (librbp_rccl.so), no torch" : "none (whole solve on this rank)");
    if (nranks == 2) {
        rbp_session_shard_joint_stream(s, 0, 1, nullptr, nullptr, nullptr, 0.0);
        rbp_rccl_pair_destroy(pair);
    }
    rbp_session_destroy(s);
    rbp_init_traj_free(&it), rbp_world_free(&wb), rbp_octomap_free(&ob), rbp_mission_free(&mb);
    return 0;
}
