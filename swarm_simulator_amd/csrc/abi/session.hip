// session.hip — the C ABI of include/rbp.h on top of the HIP kernels.
//
// rbp_corridor_update / rbp_planner_update are the drop-in calls for Corridor::update (rbp_corridor.hpp:21-26)
// and RBPPlanner::update (rbp_planner.hpp:33-84); they are thin wrappers over the device-resident, batched
// session API that bench.py times.  There is NO CPU fallback: without a HIP device every call fails with
// RBP_ERR_NO_DEVICE.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../kernels/jqp.h"
#include "../kernels/rbp_dev.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? RBP_ERR_NO_DEVICE : RBP_ERR_HIP, \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                              \
    } while (0)

struct Arena {
    char* base = nullptr;
    size_t size = 0, off = 0;
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// host [N][mbk][w] <-> device slot [N][MB][w]
template <class T>
void repack_boxes(T* dst, int dst_mb, const T* src, int src_mb, int N, int w) {
    const int nb = std::min(dst_mb, src_mb);
    for (int a = 0; a < N; ++a) memcpy(dst + (size_t)a * dst_mb * w, src + (size_t)a * src_mb * w, sizeof(T) * (size_t)nb * w);
}

}  // namespace

int rbp_set_error(int code, const char* msg) { return fail(code, msg); }  // for the other translation units (kernels/edt.hip)

// A context owns one device arena that successive sessions (and the synchronous one-shot calls) reuse: the drop-in calls
// rbp_corridor_update / rbp_planner_update would otherwise hipMalloc + hipFree ~10 MB per plan.
static rbp_solver_opts default_solver_opts() {
    rbp_solver_opts o;
    rbp_solver_opts_defaults(&o);
    return o;
}

struct rbp_ctx {
    int device = 0;
    char* base = nullptr;
    size_t cap = 0;
    bool busy = false;  // a live session is using the arena
    rbp_solver_opts opts = default_solver_opts();
    char* ws_base = nullptr;  // QP workspace of the context's sessions: reserved by the first PLANNER run that needs it, kept, grown on demand
    size_t ws_cap = 0;
};

struct rbp_session {
    int device = 0;
    int n_cu = 256;
    DevSession d{};
    rbp_param param{};
    Arena arena;
    rbp_ctx* ctx = nullptr;  // non-null: the arena belongs to this context
    rbp_solver_opts opts = default_solver_opts();
    // QP workspace: reserved by the first PLANNER run for the options then in force (not by create: a corridor-only session needs none, and
    // the grid-wide joint solver's is large); a session's own allocation, or its context's
    void* qp_ws = nullptr;
    size_t qp_ws_per_mission = 0;
    char* ws_own = nullptr;
    bool ws_joint = false;    // what the workspace was laid out for
    int bs = 1, biter = 0;    // batch schedule of the plan (setBatch, rbp_planner.hpp:849-872)
    std::vector<int> Mk, MBk;         // per-mission segments / box capacity (host copy of DevSession::Mk, MBk)
    std::vector<DevWorld> worlds_h;
    // planner-stage inputs as uploaded, in device layout (so that a run which overwrote them can be reset)
    std::vector<int> sfc_count0;
    std::vector<double> sfc_box0, sfc_time0, rsfc_time0;
    std::vector<float> rsfc_normal0;
    bool have_corridor_inputs = false;
    bool joint_wide = false;  // the joint QP (plan/sequential = false) runs on the grid-wide solver (kernels/jqp.hip): decided per PLANNER run
    JointStats joint_stats{};
    // phase-split schedule of the batch QPs (kernels/qp_phase.inc): the session's missions run as up to QP_MAX_GROUPS groups on streams
    // of their own, forked from / joined into the caller's stream by events (created on first use)
    static constexpr int QP_MAX_GROUPS = 8;
    hipStream_t gstream[QP_MAX_GROUPS] = {};
    hipEvent_t gevent[QP_MAX_GROUPS + 1] = {};
    int n_gstream = 0;
    int last_stages = 0;      // stages of the last rbp_session_run (time_scale only concerns a run that included the planner)
    // rbp_session_run_async of a grid-wide joint session: the solver's host loop (one synchronisation per interior-point round) runs on a
    // thread and a stream of the session's own; every later call on the session joins it first (join_worker)
    std::thread worker;
    int worker_rc = RBP_OK;
    hipStream_t jstream = nullptr;
    hipEvent_t jev_in = nullptr;
    // rbp_session_shard_joint: this session is one rank of a pair that shares a joint solve's factorisation (kernels/jqp.h JointShard);
    // the exchange buffers are the session's own
    JointShard shard{};
    char* shard_buf = nullptr;
};

// waits for the session's asynchronous joint run, if one is in flight; reports its error once
static int join_worker(rbp_session* s) {
    if (!s->worker.joinable()) return RBP_OK;
    s->worker.join();
    const int rc = s->worker_rc;
    s->worker_rc = RBP_OK;
    return rc ? fail(rc, "joint QP (asynchronous run): HIP error") : RBP_OK;
}

extern "C" {

const char* rbp_version(void) { return "rbp-mi355x 0.3 (gfx950)"; }
int rbp_abi_version(void) { return RBP_ABI_VERSION; }
size_t rbp_sizeof(int which) {
    switch (which) {
        case RBP_SIZEOF_WORLD: return sizeof(rbp_world);
        case RBP_SIZEOF_MISSION: return sizeof(rbp_mission);
        case RBP_SIZEOF_PARAM: return sizeof(rbp_param);
        case RBP_SIZEOF_PLAN: return sizeof(rbp_plan);
        case RBP_SIZEOF_COUNTERS: return sizeof(rbp_counters);
        case RBP_SIZEOF_DEVICE_ARRAYS: return sizeof(rbp_device_arrays);
        case RBP_SIZEOF_SOLVER_OPTS: return sizeof(rbp_solver_opts);
        default: return 0;
    }
}
const char* rbp_last_error(void) { return g_err.c_str(); }

int rbp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void rbp_param_defaults(rbp_param* p) {  // param.hpp:44-70
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->world_min[0] = -5, p->world_min[1] = -5, p->world_min[2] = 0;
    p->world_max[0] = 5, p->world_max[1] = 5, p->world_max[2] = 2.5;
    p->grid_xy_res = 0.3, p->grid_z_res = 0.6, p->grid_margin = 0.2, p->ecbs_w = 1.3;
    p->box_xy_res = 0.1, p->box_z_res = 0.1;
    p->time_scale = 1, p->time_step = 1, p->downwash = 2.0;
    p->n = 5, p->phi = 3, p->sequential = 0, p->batch_size = 4, p->batch_iter = 0, p->iteration = 1;
    p->log = 0;
    p->timescale_rule = RBP_TIMESCALE_ALL_REAL_ROOTS;
}

void rbp_solver_opts_defaults(rbp_solver_opts* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->size = (int32_t)sizeof(*o);
    o->polish = 1, o->joint_wide_min_agents = 16, o->joint_corrector = 1, o->joint_schedule = 0;
    o->qp_schedule = 0, o->qp_variant = 0, o->qp_block_order = 1, o->qp_groups = 0, o->qp_rounds = 0;
    o->qp_far_slack = 0.7;
}

static int check_solver_opts(const rbp_solver_opts* o) {
    if (!o) return fail(RBP_ERR_BAD_ARGUMENT, "null solver options");
    if (o->size != (int32_t)sizeof(rbp_solver_opts)) return fail(RBP_ERR_BAD_ARGUMENT, "rbp_solver_opts.size does not match this library (fill it with rbp_solver_opts_defaults)");
    if (o->joint_wide_min_agents < 0 || o->joint_schedule < 0 || o->joint_schedule > 3 || o->qp_schedule < 0 || o->qp_schedule > 2 ||
        !(o->qp_variant == 0 || o->qp_variant == 2 || o->qp_variant == 4) || o->qp_groups < 0 || o->qp_rounds < 0 || !(o->qp_far_slack == o->qp_far_slack))
        return fail(RBP_ERR_BAD_ARGUMENT, "rbp_solver_opts: field out of range");
    return RBP_OK;
}

static size_t al(size_t n) { return ((n + 255) & ~size_t(255)) + 256; }

// effective batch size and number of batches solved per pass (setBatch, rbp_planner.hpp:849-872; the loop of :142)
static void batch_schedule(const rbp_param& p, int N, int* bs_out, int* biter_out) {
    int bs = p.sequential ? p.batch_size : N;
    if (bs <= 0) bs = 1;
    if (bs > N) bs = N;
    const int bmax = (N + bs - 1) / bs;
    int biter = p.sequential ? p.batch_iter : 1;
    if (p.sequential && (biter < 0 || biter > bmax)) biter = bmax;
    *bs_out = bs, *biter_out = biter;
}

static int session_create_impl(rbp_session** out, int device, int K, const rbp_world* worlds, const rbp_mission* missions,
                               const rbp_param* param, const rbp_plan* plans, rbp_ctx* ctx) {
    if (!out || K <= 0 || !worlds || !missions || !param || !plans) return fail(RBP_ERR_BAD_ARGUMENT, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RBP_ERR_NO_DEVICE, "no HIP device: the RBP path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(RBP_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    // every mission keeps its own M = makespan + 2 (ecbs_planner.hpp:41-43) and box capacity; only N is common
    const int N = plans[0].N;
    int M = 0, MB = 0;  // session maxima = slot strides
    for (int k = 0; k < K; ++k) {
        if (plans[k].N != N || missions[k].N != N) return fail(RBP_ERR_BAD_ARGUMENT, "all missions of a session must share N");
        if (plans[k].N <= 0 || plans[k].M < 2 || plans[k].max_boxes <= 0)
            return fail(RBP_ERR_BAD_ARGUMENT, "need N >= 1, M >= 2, max_boxes >= 1");
        if (!plans[k].T || !plans[k].init_traj || !worlds[k].dist)
            return fail(RBP_ERR_BAD_ARGUMENT, "plan.T / plan.init_traj / world.dist must be set");
        M = std::max(M, (int)plans[k].M), MB = std::max(MB, (int)plans[k].max_boxes);
    }
    if (param->n != 5 || param->phi != 3) return fail(RBP_ERR_UNSUPPORTED_DEGREE, "RBPPlanner: n should be 5, phi 3");
    if (param->timescale_rule != RBP_TIMESCALE_ALL_REAL_ROOTS && param->timescale_rule != RBP_TIMESCALE_FIRST_EIGENVALUES)
        return fail(RBP_ERR_BAD_ARGUMENT, "rbp_param.timescale_rule must be RBP_TIMESCALE_ALL_REAL_ROOTS (0) or RBP_TIMESCALE_FIRST_EIGENVALUES (1)");
    // the SFC kernel caches at most SFC_MAXS sample keys per axis (isObstacleInBox walks the box on the box_res lattice,
    // rbp_corridor.hpp:47-63): reject worlds / resolutions that would be truncated instead of growing boxes through obstacles
    if (!(param->box_xy_res > 0) || !(param->box_z_res > 0)) return fail(RBP_ERR_BAD_ARGUMENT, "box/xy_res and box/z_res must be positive");
    for (int a = 0; a < 3; ++a) {
        const double res = a < 2 ? param->box_xy_res : param->box_z_res;
        const double ext = param->world_max[a] - param->world_min[a];
        if (!(ext >= 0) || std::ceil(ext / res) + 3 > SFC_MAXS)
            return fail(RBP_ERR_BAD_ARGUMENT, "world extent / box resolution exceeds the SFC sample cache (" + std::to_string(SFC_MAXS - 3) + " steps per axis)");
    }
    int bs = 1, biter = 0;
    batch_schedule(*param, N, &bs, &biter);

    struct Guard {  // every error path below releases the session (and with it the arena)
        rbp_session* s;
        ~Guard() {
            if (s) rbp_session_destroy(s);
        }
    } guard{new rbp_session()};
    rbp_session* s = guard.s;
    s->device = device;
    s->param = *param;
    s->Mk.resize(K), s->MBk.resize(K);
    for (int k = 0; k < K; ++k) s->Mk[k] = plans[k].M, s->MBk[k] = plans[k].max_boxes;
    const int P = M + 1, npair = N * (N - 1) / 2, oq = 6 * M;
    s->bs = bs, s->biter = biter;
    if (ctx) s->opts = ctx->opts;
    {
        hipDeviceProp_t prop;
        s->n_cu = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
    }

    // missions that share a map (same host grid pointer and shape, e.g. several passes of a map sweep) share one device copy
    auto same_grid = [&](int a, int b) {
        return worlds[a].dist == worlds[b].dist && worlds[a].dim[0] == worlds[b].dim[0] && worlds[a].dim[1] == worlds[b].dim[1] &&
               worlds[a].dim[2] == worlds[b].dim[2];
    };
    std::vector<int> grid_of(K);
    for (int k = 0; k < K; ++k) {
        grid_of[k] = k;
        for (int j = 0; j < k; ++j)
            if (grid_of[j] == j && same_grid(j, k)) {
                grid_of[k] = j;
                break;
            }
    }
    size_t grid_bytes = 0;
    for (int k = 0; k < K; ++k)
        if (grid_of[k] == k) grid_bytes += al(sizeof(float) * (size_t)worlds[k].dim[0] * worlds[k].dim[1] * worlds[k].dim[2]);
    const size_t total = grid_bytes + al(sizeof(DevWorld) * K) + 2 * al(sizeof(int) * K) + al(sizeof(float) * (size_t)K * N * P * 3) +
                         al(sizeof(double) * K * P) + 2 * al(sizeof(double) * (size_t)K * N * 9) + al(sizeof(double) * K * N) +
                         2 * al(sizeof(double) * (size_t)K * N * 3) + al(sizeof(int) * K * N) + al(sizeof(unsigned) * (size_t)K * SFC_MASK_WORDS) +
                         al(sizeof(double) * (size_t)K * N * MB * 6) + al(sizeof(double) * (size_t)K * N * MB) +
                         al(sizeof(float) * (size_t)K * npair * M * 3 + 16) + al(sizeof(double) * K * M) +
                         2 * al(sizeof(double) * (size_t)K * N * 3 * oq) + al(sizeof(int) * K) + al(sizeof(double) * K * SC_N) +
                         al(sizeof(unsigned long long) * K * CT_N) + al(sizeof(int) * K) + al(sizeof(unsigned long long) * K) + 4096;
    if (ctx) {
        if (ctx->busy) return fail(RBP_ERR_BAD_ARGUMENT, "rbp_ctx: the context's arena is in use by another session");
        if (ctx->device != device) return fail(RBP_ERR_BAD_ARGUMENT, "rbp_ctx: context belongs to another device");
        if (ctx->cap < total) {  // grow (with headroom, so that a sequence of slightly different plans does not reallocate)
            if (ctx->base) (void)hipFree(ctx->base);
            ctx->base = nullptr, ctx->cap = 0;
            const size_t want = total + total / 4;
            hipError_t e = hipMalloc((void**)&ctx->base, want);
            if (e != hipSuccess) return fail(RBP_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
            ctx->cap = want;
        }
        s->ctx = ctx, ctx->busy = true;
        s->arena.base = ctx->base;
    } else {
        hipError_t e = hipMalloc((void**)&s->arena.base, total);
        if (e != hipSuccess) return fail(RBP_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    s->arena.size = total;
    // synchronous on purpose: small pageable host-to-device copies may be carried out by the CPU ahead of work queued on the
    // stream, so an asynchronous clear could wipe what the uploads below have just written
    HIP_TRY(hipMemset(s->arena.base, 0, total));
    HIP_TRY(hipDeviceSynchronize());
    DevSession& d = s->d;
    d.K = K, d.N = N, d.M = M, d.max_boxes = MB, d.npair = npair;
    d.agent_begin = 0, d.agent_end = N;
    for (int a = 0; a < 3; ++a) {
        const double bres = a < 2 ? param->box_xy_res : param->box_z_res;
        d.sfc_cap[a] = std::min(SFC_MAXS, (int)std::ceil((param->world_max[a] - param->world_min[a]) / bres) + 4);
    }
    d.sfc_mask_words = 0;
    for (int k = 0; k < K; ++k) {
        const size_t nc = (size_t)worlds[k].dim[0] * worlds[k].dim[1] * worlds[k].dim[2];
        const size_t words = ((((nc + 63) >> 6) * 2 + 2) + 3) & ~size_t(3);
        if (words <= SFC_MASK_WORDS) d.sfc_mask_words = std::max(d.sfc_mask_words, (int)words);
    }
    {   // sfc_kernel keeps box_log [max_boxes][M + 1] per agent of a workgroup in LDS, beside the mask and the key lists: refuse here, with the
        // numbers, what the launch would refuse with a generic HIP error (the QP kernels take M <= QP_MAX_M = 128; the corridor's
        // bound depends on max_boxes, which defaults to M)
        int lds_max = 160 * 1024;
        (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
        const size_t need = corridor_lds_bytes(d);
        if (need > (size_t)std::max(lds_max, 64 * 1024))
            return fail(RBP_ERR_BAD_ARGUMENT, "Corridor: " + std::to_string(need) + " bytes of LDS per workgroup for max_boxes = " + std::to_string(MB) +
                                                  ", M = " + std::to_string(M) + " (box_log is [max_boxes][M + 1] per agent) exceed the device's " +
                                                  std::to_string(lds_max) + "; pass a smaller plan.max_boxes (the reference's corridors hold ~6-10 boxes per agent)");
    }
    for (int a = 0; a < 3; ++a) d.p.world_min[a] = param->world_min[a], d.p.world_max[a] = param->world_max[a];
    d.p.box_xy_res = param->box_xy_res, d.p.box_z_res = param->box_z_res, d.p.downwash = param->downwash;
    d.p.sequential = param->sequential, d.p.batch_size = param->batch_size, d.p.batch_iter = param->batch_iter;
    d.p.iteration = param->iteration, d.p.time_scale = param->time_scale;
    d.p.timescale_rule = param->timescale_rule;
    d.p.polish = 1, d.p.far_slack = 0.7;  // (rbp_solver_opts.polish / qp_far_slack of the run)

    Arena& A = s->arena;
    s->worlds_h.resize(K);
#define UP(dst, src, bytes) HIP_TRY(hipMemcpy((void*)(dst), (src), (bytes), hipMemcpyHostToDevice))
    for (int k = 0; k < K; ++k) {
        size_t n = (size_t)worlds[k].dim[0] * worlds[k].dim[1] * worlds[k].dim[2];
        float* g;
        if (grid_of[k] == k) {
            g = A.take<float>(n);
            UP(g, worlds[k].dist, sizeof(float) * n);
        } else {
            g = const_cast<float*>(s->worlds_h[grid_of[k]].dist);
        }
        DevWorld& w = s->worlds_h[k];
        for (int a = 0; a < 3; ++a) w.dim[a] = worlds[k].dim[a], w.key_min[a] = worlds[k].key_min[a];
        w.res = worlds[k].res;
        w.dist = g;
    }
    DevWorld* dw = A.take<DevWorld>(K);
    UP(dw, s->worlds_h.data(), sizeof(DevWorld) * K);
    d.worlds = dw;
    int* dMk = A.take<int>(K);
    int* dMBk = A.take<int>(K);
    UP(dMk, s->Mk.data(), sizeof(int) * K);
    UP(dMBk, s->MBk.data(), sizeof(int) * K);
    d.Mk = dMk, d.MBk = dMBk;
    float* traj = A.take<float>((size_t)K * N * P * 3);
    double* T = A.take<double>((size_t)K * P);
    double* start = A.take<double>((size_t)K * N * 9);
    double* goal = A.take<double>((size_t)K * N * 9);
    double* radius = A.take<double>((size_t)K * N);
    double* mv = A.take<double>((size_t)K * N * 3);
    double* ma = A.take<double>((size_t)K * N * 3);
    d.sfc_mask = A.take<unsigned>((size_t)K * SFC_MASK_WORDS);
    d.sfc_count = A.take<int>((size_t)K * N);
    d.sfc_box = A.take<double>((size_t)K * N * MB * 6);
    d.sfc_time = A.take<double>((size_t)K * N * MB);
    d.rsfc_normal = A.take<float>((size_t)K * npair * M * 3 + 4);
    d.rsfc_time = A.take<double>((size_t)K * M);
    d.ctrl = A.take<double>((size_t)K * N * 3 * oq);
    d.coef = A.take<double>((size_t)K * N * 3 * oq);
    d.status = A.take<int>(K);
    d.scalars = A.take<double>((size_t)K * SC_N);
    d.counters = A.take<unsigned long long>((size_t)K * CT_N);
    d.qp_order = A.take<int>(K);
    d.qp_cost = A.take<unsigned long long>(K);
    if (A.off > A.size) return fail(RBP_ERR_HIP, "arena overflow (internal sizing error)");
    bool have_corr = true;
    for (int k = 0; k < K; ++k)
        have_corr = have_corr && plans[k].sfc_count && plans[k].sfc_box && plans[k].sfc_time && plans[k].rsfc_normal && plans[k].rsfc_time;
    // corridor outputs handed in as planner inputs only count if some agent actually has boxes
    if (have_corr) {
        bool any = false;
        for (int k = 0; k < K && !any; ++k)
            for (int a = 0; a < N && !any; ++a) any = plans[k].sfc_count[a] > 0;
        have_corr = any;
    }
    s->have_corridor_inputs = have_corr;
    if (have_corr) {
        s->sfc_count0.assign((size_t)K * N, 0), s->sfc_box0.assign((size_t)K * N * MB * 6, 0.0), s->sfc_time0.assign((size_t)K * N * MB, 0.0);
        s->rsfc_normal0.assign((size_t)K * npair * M * 3, 0.0f), s->rsfc_time0.assign((size_t)K * M, 0.0);
    }
    for (int k = 0; k < K; ++k) {
        const int Mq = plans[k].M, Pq = Mq + 1, MBq = plans[k].max_boxes;
        UP(traj + (size_t)k * N * P * 3, plans[k].init_traj, sizeof(float) * (size_t)N * Pq * 3);
        UP(T + (size_t)k * P, plans[k].T, sizeof(double) * Pq);
        UP(start + (size_t)k * N * 9, missions[k].start, sizeof(double) * N * 9);
        UP(goal + (size_t)k * N * 9, missions[k].goal, sizeof(double) * N * 9);
        UP(radius + (size_t)k * N, missions[k].radius, sizeof(double) * N);
        UP(mv + (size_t)k * N * 3, missions[k].max_vel, sizeof(double) * N * 3);
        UP(ma + (size_t)k * N * 3, missions[k].max_acc, sizeof(double) * N * 3);
        if (have_corr) {
            memcpy(&s->sfc_count0[(size_t)k * N], plans[k].sfc_count, sizeof(int) * N);
            repack_boxes(&s->sfc_box0[(size_t)k * N * MB * 6], MB, plans[k].sfc_box, MBq, N, 6);
            repack_boxes(&s->sfc_time0[(size_t)k * N * MB], MB, plans[k].sfc_time, MBq, N, 1);
            memcpy(&s->rsfc_normal0[(size_t)k * npair * M * 3], plans[k].rsfc_normal, sizeof(float) * (size_t)npair * Mq * 3);
            memcpy(&s->rsfc_time0[(size_t)k * M], plans[k].rsfc_time, sizeof(double) * Mq);
        }
    }
    d.init_traj = traj, d.T = T, d.start = start, d.goal = goal, d.radius = radius, d.max_vel = mv, d.max_acc = ma;
#undef UP
    int rc = rbp_session_reset(s, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    guard.s = nullptr;
    *out = s;
    return RBP_OK;
}

int rbp_session_create(rbp_session** out, int device, int K, const rbp_world* worlds, const rbp_mission* missions,
                       const rbp_param* param, const rbp_plan* plans) {
    return session_create_impl(out, device, K, worlds, missions, param, plans, nullptr);
}

int rbp_session_create_in(rbp_ctx* ctx, rbp_session** out, int K, const rbp_world* worlds, const rbp_mission* missions,
                          const rbp_param* param, const rbp_plan* plans) {
    if (!ctx) return fail(RBP_ERR_BAD_ARGUMENT, "null context");
    return session_create_impl(out, ctx->device, K, worlds, missions, param, plans, ctx);
}

// T, the SFC end times and the RSFC times are never rescaled on the device (timescale_kernel only records the factor and
// rescales the coefficients; rbp_session_download applies it to its host copies), so a session can be re-run as it is:
// reset clears the per-run status / diagnostics and puts back corridor inputs the caller uploaded, in case a CORRIDOR
// stage overwrote them.
int rbp_session_reset(rbp_session* s, void* stream) {
    if (!s) return fail(RBP_ERR_BAD_ARGUMENT, "null session");
    if (int jrc = join_worker(s)) return jrc;
    hipStream_t st = (hipStream_t)stream;
    const DevSession& d = s->d;
    const int K = d.K, N = d.N, M = d.M, MB = d.max_boxes;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(d.status, 0, sizeof(int) * K, st));
    HIP_TRY(hipMemsetAsync(d.scalars, 0, sizeof(double) * K * SC_N, st));
    HIP_TRY(hipMemsetAsync(d.counters, 0, sizeof(unsigned long long) * K * CT_N, st));
    if (s->have_corridor_inputs) {
        HIP_TRY(hipMemcpyAsync(d.sfc_count, s->sfc_count0.data(), sizeof(int) * (size_t)K * N, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d.sfc_box, s->sfc_box0.data(), sizeof(double) * (size_t)K * N * MB * 6, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d.sfc_time, s->sfc_time0.data(), sizeof(double) * (size_t)K * N * MB, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d.rsfc_normal, s->rsfc_normal0.data(), sizeof(float) * (size_t)K * d.npair * M * 3, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d.rsfc_time, s->rsfc_time0.data(), sizeof(double) * (size_t)K * M, hipMemcpyHostToDevice, st));
    }
    return RBP_OK;
}

int rbp_session_set_agent_range(rbp_session* s, int32_t agent_begin, int32_t agent_end) {
    if (!s) return fail(RBP_ERR_BAD_ARGUMENT, "null session");
    if (int jrc = join_worker(s)) return jrc;
    if (agent_begin < 0 || agent_end < agent_begin || agent_end > s->d.N) return fail(RBP_ERR_BAD_ARGUMENT, "agent range outside [0, N]");
    s->d.agent_begin = agent_begin, s->d.agent_end = agent_end;
    return RBP_OK;
}

// The QP workspace of a PLANNER run.  Which solver runs -- and with it the workspace layout -- depends on the plan and on the session's
// solver options; the workspace is reserved here, by the first run that needs it (a corridor-only session reserves nothing), in the
// session's own allocation or its context's (kept across the context's sessions, grown on demand), and cleared once.
static int ensure_planner_workspace(rbp_session* s, hipStream_t st) {
    const DevSession& d = s->d;
    const int N = d.N, M = d.M, K = d.K;
    const rbp_solver_opts& o = s->opts;
    // The joint QP of a mission (plan/sequential = false: one batch of all N agents) is spread over the whole chip by kernels/jqp.hip when it
    // is wide enough to pay for a launch per phase (default: 16 agents or more, i.e. knot blocks of order >= 144).  Measured
    // (tools/gpu_joint_sweep.py, one mission / 250 / 1000 resident): 16 agents 0.067 s against 0.283 s, 9.2 k against 8.0 k, 11.2 k against
    // 7.4 k agent-trajectories/s; 32 agents 0.105 s against 1.6 s, and the one-workgroup polish (<= 256 candidate rows) accepts none of the 50
    // maps there; 8 agents: 0.038 s against 0.062 s alone but 10.6 k against 18.1 k at 250 resident -- below 16 agents a workgroup per mission
    // stays.  The grid-wide solver has no limit on N.
    const bool joint_wide = !s->param.sequential && s->biter > 0 && N >= 2 && o.joint_wide_min_agents > 0 && N >= o.joint_wide_min_agents;
    if (!joint_wide) {
        // a batch wider than the one-workgroup kernel factorises (the joint QP of a mission with more than planner_max_batch() agents), or a
        // mission with more than QP_MAX_M segments (the factor chains' LDS progress words sit behind 64 per-step assembly counters)
        if (M > QP_MAX_M) return fail(RBP_ERR_BAD_ARGUMENT, "more than " + std::to_string(QP_MAX_M) + " segments per mission are not supported by the QP kernel");
        if (s->biter > 0 && s->bs > planner_max_batch())
            return fail(RBP_ERR_BAD_ARGUMENT, "batch wider than " + std::to_string(planner_max_batch()) +
                                                  " agents (joint QP of a large mission) is not supported by the one-workgroup QP kernel "
                                                  "(rbp_solver_opts.joint_wide_min_agents selects the grid-wide solver)");
    }
    const size_t per = joint_wide ? joint_workspace_bytes(N, M) : std::max(planner_workspace_bytes_w2(N, M, s->bs), planner_workspace_bytes_w4(N, M, s->bs));
    s->joint_wide = joint_wide;
    if (s->qp_ws && s->ws_joint == joint_wide && s->qp_ws_per_mission == per) return RBP_OK;
    const size_t total = per * (size_t)K + 256;
    auto too_large = [&](hipError_t e) {
        return fail(RBP_ERR_HIP, "QP workspace: " + std::to_string(total) + " bytes (" + std::to_string(K) + " missions x " + std::to_string(per) +
                                     " bytes for " + (joint_wide ? "the grid-wide joint solver" : "the batch QP kernels") + ") could not be reserved: " +
                                     hipGetErrorString(e));
    };
    char* base = nullptr;
    if (s->ctx) {
        rbp_ctx* c = s->ctx;
        if (c->ws_cap < total) {
            HIP_TRY(hipStreamSynchronize(st));
            if (c->ws_base) (void)hipFree(c->ws_base);
            c->ws_base = nullptr, c->ws_cap = 0;
            const size_t want = total + total / 4;
            hipError_t e = hipMalloc((void**)&c->ws_base, want);
            if (e != hipSuccess) return too_large(e);
            c->ws_cap = want;
        }
        base = c->ws_base;
    } else {
        if (s->ws_own) {
            HIP_TRY(hipStreamSynchronize(st));
            (void)hipFree(s->ws_own);
            s->ws_own = nullptr;
        }
        hipError_t e = hipMalloc((void**)&s->ws_own, total);
        if (e != hipSuccess) return too_large(e);
        base = s->ws_own;
    }
    HIP_TRY(hipMemsetAsync(base, 0, total, st));
    s->qp_ws = base, s->qp_ws_per_mission = per, s->ws_joint = joint_wide;
    return RBP_OK;
}

size_t rbp_session_workspace_bytes(rbp_session* s) { return s && s->qp_ws ? s->qp_ws_per_mission : 0; }

int rbp_session_reserve_workspace(rbp_session* s, void* stream) {
    if (!s) return fail(RBP_ERR_BAD_ARGUMENT, "null session");
    if (int jrc = join_worker(s)) return jrc;
    HIP_TRY(hipSetDevice(s->device));
    return ensure_planner_workspace(s, (hipStream_t)stream);
}

int rbp_session_set_solver_opts(rbp_session* s, const rbp_solver_opts* o) {
    if (!s) return fail(RBP_ERR_BAD_ARGUMENT, "null session");
    if (int jrc = join_worker(s)) return jrc;
    int rc = check_solver_opts(o);
    if (rc) return rc;
    s->opts = *o;
    return RBP_OK;
}

// `async`: a grid-wide joint PLANNER stage is handed to the session's worker thread (everything else only enqueues anyway)
static int run_impl(rbp_session* s, int stages, void* stream, bool async) {
    if (!s) return fail(RBP_ERR_BAD_ARGUMENT, "null session");
    if (int jrc = join_worker(s)) return jrc;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(s->device));
    const rbp_solver_opts& o = s->opts;
    if (stages & RBP_STAGE_PLANNER) {
        int rc = ensure_planner_workspace(s, st);
        if (rc) return rc;
        // refusals of a sharded session BEFORE anything is enqueued or any state of the session changes (joint_wide is known now)
        if (s->shard.nranks == 2 && !s->joint_wide)
            return fail(RBP_ERR_BAD_ARGUMENT, "rbp_session_shard_joint: this plan does not run on the grid-wide joint solver (rbp_solver_opts.joint_wide_min_agents), there is no factorisation to share");
        if (s->shard.nranks == 2 && async)
            return fail(RBP_ERR_BAD_ARGUMENT, "rbp_session_run_async: a session sharded over two ranks (rbp_session_shard_joint) calls the exchange hook on the caller's thread: use rbp_session_run");
    }
    s->last_stages = stages;
    s->d.p.polish = o.polish ? 1 : 0;
    s->d.p.far_slack = o.qp_far_slack;
    if (stages & RBP_STAGE_CORRIDOR) {
        int rc = launch_corridor(s->d, st);
        if (rc) return rc;
    }
    if ((stages & RBP_STAGE_PLANNER) && s->joint_wide) {
        // grid-wide joint QP: a launch per phase; the host learns once per interior-point iteration whether any mission is still
        // running, so this call SYNCHRONISES the stream (unlike the batch path, which only enqueues)
        launch_planner_prologue(s->d, st);
        int rc = RBP_OK;
        JointOpts jo;
        jo.corrector = o.joint_corrector ? 1 : 0, jo.schedule = o.joint_schedule;
        if (s->shard.nranks == 2) jo.shard = &s->shard;
        if (async) {
            // rbp_session_run_async: the host loop moves to a thread and a stream of the session's own, ordered after what the caller's
            // stream holds so far (inputs, the CORRIDOR stage, the prologue) by an event
            if (!s->jstream) {
                HIP_TRY(hipStreamCreateWithFlags(&s->jstream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&s->jev_in, hipEventDisableTiming));
            }
            HIP_TRY(hipEventRecord(s->jev_in, st));
            HIP_TRY(hipGetLastError());
            s->worker_rc = RBP_OK;
            s->worker = std::thread([s, jo]() {
                int wrc = RBP_OK;
                hipStream_t js = s->jstream;
                if (hipSetDevice(s->device) != hipSuccess || hipStreamWaitEvent(js, s->jev_in, 0) != hipSuccess) wrc = RBP_ERR_HIP;
                if (!wrc && s->d.p.iteration > 0) wrc = launch_planner_joint(s->d, s->qp_ws, js, &s->joint_stats, jo);
                if (!wrc) {
                    launch_planner_epilogue(s->d, js);
                    if (hipStreamSynchronize(js) != hipSuccess || hipGetLastError() != hipSuccess) wrc = RBP_ERR_HIP;
                }
                s->worker_rc = wrc;
            });
            return RBP_OK;
        }
        if (s->d.p.iteration > 0) rc = launch_planner_joint(s->d, s->qp_ws, st, &s->joint_stats, jo);
        if (rc == RBP_ERR_EXCHANGE) return rc;  // (kernels/jqp.hip has recorded which exchange failed and why)
        if (rc) return fail(rc, "joint QP: HIP error");
        launch_planner_epilogue(s->d, st);
    } else if ((stages & RBP_STAGE_PLANNER) && o.qp_schedule == 2 && !planner_has_phase_split()) {
        return fail(RBP_ERR_BAD_ARGUMENT, "rbp_solver_opts.qp_schedule = 2: the phase-split schedule is not part of the release library (it loses everywhere: DESIGN.md 3.3); "
                                          "the developer build lib/librbp_hip_dev.so (`make dev`) carries it");
    } else if ((stages & RBP_STAGE_PLANNER) && o.qp_schedule == 2) {
        // phase split (kernels/qp_phase.inc): chip-wide row sweeps, one workgroup per mission for the chains; the missions are spread over a
        // few streams whose rounds overlap
        int G = o.qp_groups > 0 ? o.qp_groups : (s->d.K >= 1024 ? 2 : 1);
        G = std::max(1, std::min(G, std::min((int)rbp_session::QP_MAX_GROUPS, s->d.K)));
        while (s->n_gstream < G) {
            const int g = s->n_gstream;
            if (g == 0) HIP_TRY(hipEventCreateWithFlags(&s->gevent[0], hipEventDisableTiming));
            HIP_TRY(hipStreamCreateWithFlags(&s->gstream[g], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&s->gevent[1 + g], hipEventDisableTiming));
            s->n_gstream++;
        }
        launch_planner_phased(s->d, s->qp_ws, s->qp_ws_per_mission, st, s->gstream, s->gevent, G, o.qp_rounds);
    } else if (stages & RBP_STAGE_PLANNER) {
        // one workgroup per mission (qp_batch_kernel; the default: see DESIGN.md 3.3 for the A/B against the phase split).  Two workgroups
        // per CU (the 256-thread build) pay off as soon as there are more missions than CUs: the 512-thread build would need a second round
        // (measured at 300/400/500 missions: +17-21 %)
        const bool w4 = o.qp_variant ? o.qp_variant == 4 : s->d.K > s->n_cu;
        DevSession d = s->d;
        if (!o.qp_block_order) d.qp_order = nullptr, d.qp_cost = nullptr;
        if (w4)
            launch_planner_w4(d, s->qp_ws, s->qp_ws_per_mission, st);
        else
            launch_planner_w2(d, s->qp_ws, s->qp_ws_per_mission, st);
    }
    HIP_TRY(hipGetLastError());
    return RBP_OK;
}

int rbp_session_run(rbp_session* s, int stages, void* stream) { return run_impl(s, stages, stream, false); }
int rbp_session_run_async(rbp_session* s, int stages, void* stream) { return run_impl(s, stages, stream, true); }
int rbp_session_wait(rbp_session* s) {
    if (!s) return fail(RBP_ERR_BAD_ARGUMENT, "null session");
    return join_worker(s);
}

static int shard_joint_impl(rbp_session* s, int32_t rank, int32_t nranks, rbp_exchange_fn exchange, rbp_exchange_stream_fn exchange_stream,
                            rbp_exchange_abort_fn abort_peer, void* user, double timeout_s) {
    if (!s) return fail(RBP_ERR_BAD_ARGUMENT, "null session");
    if (int jrc = join_worker(s)) return jrc;
    if (nranks == 1) {  // undo: this session runs both chains again (the buffers stay until destroy)
        s->shard.nranks = 1, s->shard.rank = 0, s->shard.exchange = nullptr, s->shard.exchange_stream = nullptr, s->shard.abort_peer = nullptr, s->shard.user = nullptr;
        return RBP_OK;
    }
    if (nranks != 2 || rank < 0 || rank > 1 || (!exchange && !exchange_stream))
        return fail(RBP_ERR_BAD_ARGUMENT, "rbp_session_shard_joint: the twisted elimination has two chains -- nranks must be 2 (or 1 to undo), rank 0 or 1, and an exchange hook is needed");
    if (s->param.sequential || s->d.N < 2)
        return fail(RBP_ERR_BAD_ARGUMENT, "rbp_session_shard_joint: only a joint QP (plan/sequential = false) has a factorisation to share; the sequential schedule shards by mission");
    const size_t cap = joint_exchange_bytes(s->d.N, s->d.M, s->d.K);
    if (!s->shard_buf) {
        HIP_TRY(hipSetDevice(s->device));
        const hipError_t e = hipMalloc((void**)&s->shard_buf, 2 * cap + 64);  // (+ the error word of the stream-ordered exchange)
        if (e != hipSuccess)
            return fail(RBP_ERR_HIP, "rbp_session_shard_joint: exchange buffers (2 x " + std::to_string(cap) + " bytes) could not be reserved: " + hipGetErrorString(e));
    }
    s->shard.rank = rank, s->shard.nranks = 2, s->shard.send = (double*)s->shard_buf, s->shard.recv = (double*)(s->shard_buf + cap), s->shard.cap = cap;
    s->shard.xerr = (double*)(s->shard_buf + 2 * cap);
    s->shard.exchange = exchange, s->shard.exchange_stream = exchange_stream, s->shard.abort_peer = abort_peer, s->shard.user = user, s->shard.timeout_s = timeout_s;
    return RBP_OK;
}
int rbp_session_shard_joint(rbp_session* s, int32_t rank, int32_t nranks, rbp_exchange_fn exchange, void* user) {
    return shard_joint_impl(s, rank, nranks, exchange, nullptr, nullptr, user, 0.0);
}
int rbp_session_shard_joint_stream(rbp_session* s, int32_t rank, int32_t nranks, rbp_exchange_stream_fn exchange, rbp_exchange_abort_fn abort_peer, void* user,
                                   double timeout_s) {
    return shard_joint_impl(s, rank, nranks, nullptr, exchange, abort_peer, user, timeout_s);
}

int rbp_session_download(rbp_session* s, rbp_plan* plans, int32_t* status, void* stream) {
    if (!s || !plans) return fail(RBP_ERR_BAD_ARGUMENT, "null argument");
    if (int jrc = join_worker(s)) return jrc;
    hipStream_t st = (hipStream_t)stream;
    const DevSession& d = s->d;
    const int K = d.K, N = d.N, M = d.M, MB = d.max_boxes, P = M + 1, oq = 6 * M;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<int> stat(K);
    std::vector<double> sc((size_t)K * SC_N);
    std::vector<double> boxbuf, timebuf;
#define DN(dst, src, bytes) HIP_TRY(hipMemcpy((dst), (const void*)(src), (bytes), hipMemcpyDeviceToHost))
    DN(stat.data(), d.status, sizeof(int) * K);
    DN(sc.data(), d.scalars, sizeof(double) * K * SC_N);
    int first = 0;
    int bs = 1, biter = 0;
    batch_schedule(s->param, N, &bs, &biter);
    for (int k = 0; k < K; ++k) {
        rbp_plan& p = plans[k];
        const int Mq = s->Mk[k], Pq = Mq + 1, oqq = 6 * Mq, MBq = s->MBk[k];
        if (p.N != N || p.M != Mq || p.max_boxes != MBq) return fail(RBP_ERR_BAD_ARGUMENT, "plan shape differs from the session");
        const double* q = &sc[(size_t)k * SC_N];
        // timeScale (rbp_planner.hpp:250-264) multiplies T, the SFC end times and the RSFC times by time_scale: done here, on
        // the host copies (the same IEEE products), because the device arrays stay unscaled (see rbp_session_reset)
        // ... and only when the last run included the planner: after a CORRIDOR-only run the scalar still holds the factor of
        // an earlier planner run, which has nothing to do with the corridor times just written
        const double ts = ((s->last_stages & RBP_STAGE_PLANNER) && q[SC_TIME_SCALE] > 0) ? q[SC_TIME_SCALE] : 1.0;
        DN(p.T, d.T + (size_t)k * P, sizeof(double) * Pq);
        if (ts != 1.0)
            for (int m = 0; m < Pq; ++m) p.T[m] *= ts;
        if (p.sfc_count) DN(p.sfc_count, d.sfc_count + (size_t)k * N, sizeof(int) * N);
        if (p.sfc_box) {
            if (MBq == MB) {
                DN(p.sfc_box, d.sfc_box + (size_t)k * N * MB * 6, sizeof(double) * (size_t)N * MB * 6);
            } else {
                boxbuf.resize((size_t)N * MB * 6);
                DN(boxbuf.data(), d.sfc_box + (size_t)k * N * MB * 6, sizeof(double) * (size_t)N * MB * 6);
                repack_boxes(p.sfc_box, MBq, boxbuf.data(), MB, N, 6);
            }
        }
        if (p.sfc_time) {
            timebuf.resize((size_t)N * MB);
            DN(timebuf.data(), d.sfc_time + (size_t)k * N * MB, sizeof(double) * (size_t)N * MB);
            std::vector<int> cnt(N);
            DN(cnt.data(), d.sfc_count + (size_t)k * N, sizeof(int) * N);
            if (ts != 1.0)
                for (int a = 0; a < N; ++a)
                    for (int b = 0; b < cnt[a] && b < MB; ++b) timebuf[(size_t)a * MB + b] *= ts;
            repack_boxes(p.sfc_time, MBq, timebuf.data(), MB, N, 1);
        }
        if (p.rsfc_normal) DN(p.rsfc_normal, d.rsfc_normal + (size_t)k * d.npair * M * 3, sizeof(float) * (size_t)d.npair * Mq * 3);
        if (p.rsfc_time) {
            DN(p.rsfc_time, d.rsfc_time + (size_t)k * M, sizeof(double) * Mq);
            if (ts != 1.0)
                for (int m = 0; m < Mq; ++m) p.rsfc_time[m] *= ts;
        }
        // the planner's outputs belong to a run that included the planner: after a CORRIDOR-only run the device still holds the control
        // points / coefficients of an earlier planner run (coef even rescaled by THAT run's time_scale), which do not go with the corridor
        // just written -- they are not handed out, and the solver outcome reads "no QP solved"
        const bool planned = (s->last_stages & RBP_STAGE_PLANNER) != 0;
        if (planned && p.coef) DN(p.coef, d.coef + (size_t)k * N * 3 * oq, sizeof(double) * (size_t)N * 3 * oqq);
        if (planned && p.ctrl) DN(p.ctrl, d.ctrl + (size_t)k * N * 3 * oq, sizeof(double) * (size_t)N * 3 * oqq);
        p.time_scale = ts;
        p.time_scale_alt = ((s->last_stages & RBP_STAGE_PLANNER) && q[SC_TIME_SCALE_ALT] > 0) ? q[SC_TIME_SCALE_ALT] : 1.0;
        p.total_cost = planned ? q[SC_TOTAL_COST] : 0.0;
        p.qp_iterations = planned ? (int)q[SC_IPM_ITERS] : 0;
        p.qp_solves = planned ? (int)q[SC_QP_SOLVED] : 0;
        p.qp_unpolished = planned ? (int)(q[SC_QP_SOLVED] - q[SC_POLISHED]) : 0;
        p.kkt_max = planned ? q[SC_KKT_MAX] : 0.0;
        if (biter > 0) {
            int last = biter - 1, nb = std::min(bs, N - last * bs);
            p.x_size = 3 * nb * oqq;
            p.eq_size = 3 * nb * 3 * (Mq + 1);
            int nf = N - nb;
            p.ineq_size = 2 * p.x_size + (nb * (nb - 1) / 2 + nb * nf) * oqq;
        }
        if (status) status[k] = stat[k];
        if (!first && stat[k]) first = stat[k];
    }
#undef DN
    if (first) g_err = "mission failed with status " + std::to_string(first);
    return first;
}

int rbp_session_device_arrays(rbp_session* s, int32_t mission, rbp_device_arrays* out) {
    if (!s || !out) return fail(RBP_ERR_BAD_ARGUMENT, "null argument");
    const DevSession& d = s->d;
    if (mission < 0 || mission >= d.K) return fail(RBP_ERR_BAD_ARGUMENT, "mission index outside the session");
    const size_t k = (size_t)mission;
    out->sfc_count = d.sfc_count + k * d.N;
    out->sfc_box = d.sfc_box + k * d.N * d.max_boxes * 6;
    out->sfc_time = d.sfc_time + k * d.N * d.max_boxes;
    out->rsfc_normal = d.rsfc_normal + k * d.npair * d.M * 3;
    out->rsfc_time = d.rsfc_time + k * d.M;
    out->status = d.status + k;
    out->N = d.N, out->M = d.M, out->max_boxes = d.max_boxes, out->npair = d.npair, out->device = s->device;
    return RBP_OK;
}

int rbp_session_counters(rbp_session* s, rbp_counters* out, void* stream) {
    if (!s || !out) return fail(RBP_ERR_BAD_ARGUMENT, "null argument");
    if (int jrc = join_worker(s)) return jrc;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    const int K = s->d.K;
    std::vector<unsigned long long> ct((size_t)K * CT_N);
    std::vector<double> sc((size_t)K * SC_N);
    HIP_TRY(hipMemcpy(ct.data(), s->d.counters, sizeof(unsigned long long) * ct.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sc.data(), s->d.scalars, sizeof(double) * sc.size(), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    for (int k = 0; k < K; ++k) {
        out->sfc_samples += (double)ct[(size_t)k * CT_N + CT_SFC_SAMPLES];
        out->qp_flops += sc[(size_t)k * SC_N + SC_FLOPS];
        out->qp_ipm_iters += sc[(size_t)k * SC_N + SC_IPM_ITERS];
        out->qp_solves += sc[(size_t)k * SC_N + SC_QP_SOLVED];
        out->qp_constraint_rows += sc[(size_t)k * SC_N + SC_ROWS];
        out->qp_polished += sc[(size_t)k * SC_N + SC_POLISHED];
        out->qp_row_bytes += sc[(size_t)k * SC_N + SC_ROW_BYTES];
        out->kkt_max = std::max(out->kkt_max, sc[(size_t)k * SC_N + SC_KKT_MAX]);
    }
    return RBP_OK;
}

int rbp_session_scalars(rbp_session* s, double* out, int n, void* stream) {
    if (!s || !out || n <= 0 || n > SC_N) return fail(RBP_ERR_BAD_ARGUMENT, "bad argument");
    if (int jrc = join_worker(s)) return jrc;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::vector<double> sc((size_t)s->d.K * SC_N);
    HIP_TRY(hipMemcpy(sc.data(), s->d.scalars, sizeof(double) * sc.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < s->d.K; ++k) memcpy(out + (size_t)k * n, &sc[(size_t)k * SC_N], sizeof(double) * n);
    return RBP_OK;
}

void rbp_session_destroy(rbp_session* s) {
    if (!s) return;
    (void)join_worker(s);
    if (s->jstream) {
        (void)hipSetDevice(s->device);
        (void)hipStreamDestroy(s->jstream);
        (void)hipEventDestroy(s->jev_in);
    }
    if (s->n_gstream > 0) {
        (void)hipSetDevice(s->device);
        (void)hipEventDestroy(s->gevent[0]);
        for (int g = 0; g < s->n_gstream; ++g) (void)hipStreamDestroy(s->gstream[g]), (void)hipEventDestroy(s->gevent[1 + g]);
    }
    if (s->ws_own) {
        (void)hipSetDevice(s->device);
        (void)hipFree(s->ws_own);
    }
    if (s->shard_buf) {
        (void)hipSetDevice(s->device);
        (void)hipFree(s->shard_buf);
    }
    if (s->ctx) {
        s->ctx->busy = false;  // the arena (and the QP workspace) stay with the context
    } else if (s->arena.base) {
        (void)hipSetDevice(s->device);
        (void)hipFree(s->arena.base);
    }
    delete s;
}

// ---- contexts ---------------------------------------------------------------------------------------------------------
int rbp_ctx_create(rbp_ctx** out, int device) {
    if (!out) return fail(RBP_ERR_BAD_ARGUMENT, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RBP_ERR_NO_DEVICE, "no HIP device: the RBP path has no CPU fallback");
    if (device < 0) (void)hipGetDevice(&device);  // the calling thread's current device
    if (device < 0 || device >= ndev) return fail(RBP_ERR_NO_DEVICE, "device index out of range");
    auto* c = new rbp_ctx();
    c->device = device;
    *out = c;
    return RBP_OK;
}

void rbp_ctx_destroy(rbp_ctx* c) {
    if (!c) return;
    if (c->base || c->ws_base) {
        (void)hipSetDevice(c->device);
        if (c->base) (void)hipFree(c->base);
        if (c->ws_base) (void)hipFree(c->ws_base);
    }
    delete c;
}

// The per-thread default context of the drop-in calls.  A worker thread that exits gives its arena back (hipFree); at process
// exit the HIP runtime may already be gone when thread_local destructors run, so from the first atexit handler on nothing is
// freed any more (the runtime releases device memory itself).  rbp_release_thread_context() frees it on request.
namespace {
bool g_process_exiting = false;
struct Tls {
    rbp_ctx* c = nullptr;
    ~Tls() {
        if (c && !g_process_exiting) rbp_ctx_destroy(c);
        c = nullptr;
    }
};
Tls& tls_ctx() {
    static const int registered = (atexit([] { g_process_exiting = true; }), 0);
    (void)registered;
    thread_local Tls tls;
    return tls;
}
}  // namespace

void rbp_release_thread_context(void) {
    Tls& tls = tls_ctx();
    if (tls.c) rbp_ctx_destroy(tls.c);
    tls.c = nullptr;
}

static int one_shot(rbp_ctx* ctx, const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan,
                    int stages, int agent_begin = 0, int agent_end = -1) {
    if (!mission || !param || !plan) return fail(RBP_ERR_BAD_ARGUMENT, "null argument");
    if (stages == RBP_STAGE_PLANNER) {
        if (!(plan->sfc_count && plan->sfc_box && plan->sfc_time && plan->rsfc_normal && plan->rsfc_time))
            return fail(RBP_ERR_BAD_ARGUMENT, "RBPPlanner::update needs the corridor (plan.sfc_* and plan.rsfc_*) as input");
        for (int a = 0; a < plan->N; ++a)
            if (plan->sfc_count[a] <= 0 || plan->sfc_count[a] > plan->max_boxes)
                return fail(RBP_ERR_BAD_ARGUMENT, "RBPPlanner::update: plan.sfc_count must be in [1, max_boxes] for every agent (run Corridor::update first)");
    }
    rbp_world dummy_world;
    float zero = 0.0f;
    if (!world) {  // planner stage does not read the map
        dummy_world.dim[0] = dummy_world.dim[1] = dummy_world.dim[2] = 1;
        dummy_world.key_min[0] = dummy_world.key_min[1] = dummy_world.key_min[2] = 0;
        dummy_world.res = 1.0, dummy_world.dist = &zero;
        world = &dummy_world;
    }
    // without an explicit context the calling thread's own one is used (created on first use on its current device), so the
    // drop-in calls do not allocate device memory per plan either
    Tls& tls = tls_ctx();
    if (!ctx) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        rbp_solver_opts keep = tls.c ? tls.c->opts : default_solver_opts();
        if (tls.c && tls.c->device != dev) {
            rbp_ctx_destroy(tls.c);
            tls.c = nullptr;
        }
        if (!tls.c) {
            int rc = rbp_ctx_create(&tls.c, dev);
            if (rc) return rc;
            tls.c->opts = keep;
        }
        ctx = tls.c;
    }
    rbp_session* s = nullptr;
    int rc = session_create_impl(&s, ctx->device, 1, world, mission, param, plan, ctx);
    if (rc) return rc;
    if (agent_end >= 0) rc = rbp_session_set_agent_range(s, agent_begin, agent_end);
    if (!rc) rc = rbp_session_run(s, stages, nullptr);
    if (!rc) {
        int32_t st = 0;
        rc = rbp_session_download(s, plan, &st, nullptr);
    }
    rbp_session_destroy(s);
    return rc;
}

int rbp_ctx_set_solver_opts(rbp_ctx* ctx, const rbp_solver_opts* o) {
    int rc = check_solver_opts(o);
    if (rc) return rc;
    if (!ctx) {  // the calling thread's default context
        Tls& tls = tls_ctx();
        if (!tls.c) {
            int ndev = 0, dev = 0;
            if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RBP_ERR_NO_DEVICE, "no HIP device: the RBP path has no CPU fallback");
            (void)hipGetDevice(&dev);
            rc = rbp_ctx_create(&tls.c, dev);
            if (rc) return rc;
        }
        ctx = tls.c;
    }
    ctx->opts = *o;
    return RBP_OK;
}

int rbp_corridor_update(const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan) {
    if (!world) return fail(RBP_ERR_BAD_ARGUMENT, "null world");
    return one_shot(nullptr, world, mission, param, plan, RBP_STAGE_CORRIDOR);
}

int rbp_corridor_update_range(const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan,
                              int32_t agent_begin, int32_t agent_end) {
    if (!world) return fail(RBP_ERR_BAD_ARGUMENT, "null world");
    if (agent_end < 0) return fail(RBP_ERR_BAD_ARGUMENT, "agent range outside [0, N]");
    return one_shot(nullptr, world, mission, param, plan, RBP_STAGE_CORRIDOR, agent_begin, agent_end);
}

int rbp_planner_update(const rbp_mission* mission, const rbp_param* param, rbp_plan* plan) {
    return one_shot(nullptr, nullptr, mission, param, plan, RBP_STAGE_PLANNER);
}

int rbp_ctx_corridor_update(rbp_ctx* ctx, const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan) {
    if (!ctx || !world) return fail(RBP_ERR_BAD_ARGUMENT, "null context / world");
    return one_shot(ctx, world, mission, param, plan, RBP_STAGE_CORRIDOR);
}

int rbp_ctx_planner_update(rbp_ctx* ctx, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan) {
    if (!ctx) return fail(RBP_ERR_BAD_ARGUMENT, "null context");
    return one_shot(ctx, nullptr, mission, param, plan, RBP_STAGE_PLANNER);
}

int rbp_ctx_plan_update(rbp_ctx* ctx, const rbp_world* world, const rbp_mission* mission, const rbp_param* param, rbp_plan* plan) {
    if (!ctx || !world) return fail(RBP_ERR_BAD_ARGUMENT, "null context / world");
    return one_shot(ctx, world, mission, param, plan, RBP_STAGE_ALL);
}

}  // extern "C"
