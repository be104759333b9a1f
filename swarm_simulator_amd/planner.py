"""Host-side mirror of the reference's two stage classes, on top of the C ABI (lib/librbp_hip.so).

    Corridor(world, mission, param).update(log, plan)      <-> SwarmPlanning::Corridor::update    rbp_corridor.hpp:13-26
    RBPPlanner(mission, param).update(log, plan)           <-> SwarmPlanning::RBPPlanner::update  rbp_planner.hpp:21-84

Same argument meaning and error behaviour: `update` returns True/False and mutates the PlanResult in place;
`last_error` carries what the reference would have sent to ROS_ERROR.  `Session` is the device-resident batched
form bench.py times.  The HIP library is mandatory: nothing here falls back to a CPU implementation.
"""
import ctypes as C
import os

from . import _abi as A
from .types import Mission, Param, PlanResult, World

_lib = None

ERROR_TEXT = {
    A.RBP_ERR_OBSTACLE_IN_INIT_TRAJ: "Corridor: Invalid initial trajectory. Obstacle invades initial trajectory.",
    A.RBP_ERR_UNEQUAL_TRAJ_LEN: "Corridor: size of initial trajectories must be equal",
    A.RBP_ERR_INIT_TRAJ_COLLIDE: "Corridor: initial trajectories are collided with each other",
    A.RBP_ERR_SFC_OVERFLOW: "Corridor: more SFC boxes than plan.max_boxes",
    A.RBP_ERR_QP_FAILED: "RBPPlanner: Failed to optimize QP",
    A.RBP_ERR_UNSUPPORTED_DEGREE: "RBPPlanner: n should be 5",
    A.RBP_ERR_BAD_ARGUMENT: "bad argument",
    A.RBP_ERR_NO_DEVICE: "no HIP device (the RBP path has no CPU fallback)",
    A.RBP_ERR_HIP: "HIP runtime error",
    A.RBP_ERR_EXCHANGE: "exchange between the two ranks of a sharded joint solve failed",
}


# rbp_exchange_fn of include/rbp.h: int (*)(void* user, void* send_dev, void* recv_dev, size_t bytes)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
EXCHANGE_STREAM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)   # rbp_exchange_stream_fn (+ the stream)
EXCHANGE_ABORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)                                                    # rbp_exchange_abort_fn


class RbpLibraryMissing(RuntimeError):
    pass


preloaded_hip_runtime = []   # torch's HIP runtime libraries loaded ahead of librbp_hip.so by lib() (empty: none)


def lib():
    """Load lib/librbp_hip.so; fails loudly if it was not built."""
    global _lib
    if _lib is None:
        path = os.environ.get("RBP_HIP_LIB") or os.path.join(A.LIB_DIR, "librbp_hip.so")  # env override: developer A/B builds
        if not os.path.exists(path):
            raise RbpLibraryMissing(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 and must find THEM when it initialises
        # (if this library's /opt/rocm runtime is in the process first, torch.cuda reports "No HIP GPUs are available"); with torch's
        # copies loaded first, both sides share one runtime whatever the import order -- which streams, events and device pointers
        # exchanged between them (bench.py, sharded.py) rely on.
        # RBP_PRELOAD_TORCH_HIP=0 switches the preload off (a process that never imports torch); what was preloaded is recorded in
        # `preloaded_hip_runtime` and a failure is reported, not swallowed.
        global preloaded_hip_runtime
        if os.environ.get("RBP_PRELOAD_TORCH_HIP", "1") != "0":
            import importlib.util
            sp = importlib.util.find_spec("torch")
            if sp is not None and sp.submodule_search_locations:
                tl = os.path.join(list(sp.submodule_search_locations)[0], "lib")
                for n in ("libhsa-runtime64.so", "libamdhip64.so"):
                    f = os.path.join(tl, n)
                    if os.path.exists(f):
                        try:
                            C.CDLL(f, mode=C.RTLD_GLOBAL)
                            preloaded_hip_runtime.append(f)
                        except OSError as e:
                            import warnings
                            warnings.warn(f"swarm_simulator_amd: could not preload torch's {n} ({e}); torch.cuda may not see the GPU "
                                          f"if it is imported after this library")
        L = C.CDLL(path)
        P = C.POINTER
        L.rbp_version.restype = C.c_char_p
        # the structs of include/rbp.h are written by the library: refuse a library built from another header
        if os.environ.get("RBP_HIP_LIB") and not hasattr(L, "rbp_abi_version"):
            pass  # (developer A/B builds of older commits predate the check)
        else:
            L.rbp_abi_version.restype = C.c_int
            L.rbp_sizeof.restype = C.c_size_t
            L.rbp_sizeof.argtypes = [C.c_int]
            if L.rbp_abi_version() != A.RBP_ABI_VERSION:
                raise RbpLibraryMissing(f"{path}: ABI version {L.rbp_abi_version()}, this binding expects {A.RBP_ABI_VERSION} (rebuild the library)")
            L.rbp_session_device_arrays.argtypes = [C.c_void_p, C.c_int32, P(A.rbp_device_arrays)]
            for which, t in enumerate((A.rbp_world, A.rbp_mission, A.rbp_param, A.rbp_plan, A.rbp_counters, A.rbp_device_arrays, A.rbp_solver_opts)):
                if L.rbp_sizeof(which) != C.sizeof(t):
                    raise RbpLibraryMissing(f"{path}: sizeof({t.__name__}) is {L.rbp_sizeof(which)} in the library, {C.sizeof(t)} in this binding")
            L.rbp_release_thread_context.restype = None
        L.rbp_last_error.restype = C.c_char_p
        L.rbp_device_count.restype = C.c_int
        L.rbp_param_defaults.argtypes = [P(A.rbp_param)]
        L.rbp_param_defaults.restype = None
        L.rbp_corridor_update.argtypes = [P(A.rbp_world), P(A.rbp_mission), P(A.rbp_param), P(A.rbp_plan)]
        L.rbp_corridor_update_range.argtypes = [P(A.rbp_world), P(A.rbp_mission), P(A.rbp_param), P(A.rbp_plan), C.c_int32, C.c_int32]
        L.rbp_planner_update.argtypes = [P(A.rbp_mission), P(A.rbp_param), P(A.rbp_plan)]
        L.rbp_session_set_agent_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.rbp_session_create.argtypes = [P(C.c_void_p), C.c_int, C.c_int, P(A.rbp_world), P(A.rbp_mission), P(A.rbp_param),
                                         P(A.rbp_plan)]
        L.rbp_session_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rbp_session_run_async.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rbp_session_wait.argtypes = [C.c_void_p]
        L.rbp_session_shard_joint.argtypes = [C.c_void_p, C.c_int32, C.c_int32, EXCHANGE_FN, C.c_void_p]
        L.rbp_session_shard_joint_stream.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.rbp_session_download.argtypes = [C.c_void_p, P(A.rbp_plan), A.c_int32_p, C.c_void_p]
        L.rbp_session_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.rbp_session_counters.argtypes = [C.c_void_p, P(A.rbp_counters), C.c_void_p]
        L.rbp_session_scalars.argtypes = [C.c_void_p, A.c_double_p, C.c_int, C.c_void_p]
        L.rbp_session_destroy.argtypes = [C.c_void_p]
        L.rbp_session_destroy.restype = None
        try:  # (developer A/B builds of older commits loaded through RBP_HIP_LIB may predate the GPU distance grid)
            L.rbp_edt_dims.argtypes = [C.c_double, C.c_double * 3, C.c_double * 3, C.c_int32 * 3, C.c_int32 * 3]
            L.rbp_edt_build.argtypes = [A.c_int32_p, C.c_int64, C.c_double, C.c_double * 3, C.c_double * 3, C.c_double, A.c_float_p]
        except AttributeError:
            if not os.environ.get("RBP_HIP_LIB"):
                raise
        L.rbp_solver_opts_defaults.argtypes = [P(A.rbp_solver_opts)]
        L.rbp_solver_opts_defaults.restype = None
        L.rbp_session_set_solver_opts.argtypes = [C.c_void_p, P(A.rbp_solver_opts)]
        L.rbp_ctx_set_solver_opts.argtypes = [C.c_void_p, P(A.rbp_solver_opts)]
        L.rbp_session_reserve_workspace.argtypes = [C.c_void_p, C.c_void_p]
        L.rbp_session_workspace_bytes.argtypes = [C.c_void_p]
        L.rbp_session_workspace_bytes.restype = C.c_size_t
        L.rbp_ctx_create.argtypes = [P(C.c_void_p), C.c_int]
        L.rbp_ctx_destroy.argtypes = [C.c_void_p]
        L.rbp_ctx_destroy.restype = None
        L.rbp_ctx_corridor_update.argtypes = [C.c_void_p, P(A.rbp_world), P(A.rbp_mission), P(A.rbp_param), P(A.rbp_plan)]
        L.rbp_ctx_planner_update.argtypes = [C.c_void_p, P(A.rbp_mission), P(A.rbp_param), P(A.rbp_plan)]
        L.rbp_ctx_plan_update.argtypes = [C.c_void_p, P(A.rbp_world), P(A.rbp_mission), P(A.rbp_param), P(A.rbp_plan)]
        L.rbp_session_create_in.argtypes = [C.c_void_p, P(C.c_void_p), C.c_int, P(A.rbp_world), P(A.rbp_mission), P(A.rbp_param),
                                            P(A.rbp_plan)]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "rbp_param_defaults", "rbp_corridor_update", "rbp_corridor_update_range", "rbp_planner_update", "rbp_session_create",
    "rbp_session_run", "rbp_session_run_async", "rbp_session_wait", "rbp_session_shard_joint", "rbp_session_shard_joint_stream", "rbp_session_set_agent_range",
    "rbp_session_download", "rbp_session_reset", "rbp_session_destroy", "rbp_session_counters", "rbp_session_scalars",
    "rbp_session_device_arrays",
    "rbp_version", "rbp_abi_version", "rbp_sizeof", "rbp_release_thread_context",
    "rbp_last_error", "rbp_device_count",
    "rbp_ctx_create", "rbp_ctx_destroy", "rbp_ctx_corridor_update", "rbp_ctx_planner_update", "rbp_ctx_plan_update",
    "rbp_session_create_in",
    "rbp_solver_opts_defaults", "rbp_session_set_solver_opts", "rbp_ctx_set_solver_opts", "rbp_session_workspace_bytes", "rbp_session_reserve_workspace",
    "rbp_edt_dims", "rbp_edt_build",
]


def last_error():
    return lib().rbp_last_error().decode()


def solver_opts(**kw):
    """rbp_solver_opts (include/rbp.h) with the library's defaults, fields overridden by keyword: polish, joint_wide_min_agents,
    joint_corrector, joint_schedule (0 auto / 1 look-ahead / 2 bulk / 3 bulk with two pivot tiles per pass), qp_schedule (0 auto / 1 one workgroup per mission / 2 phase split),
    qp_variant (0 / 2 / 4), qp_block_order, qp_groups, qp_rounds, qp_far_slack (metres; <= 0: off).  The library reads no environment variables: these are the switches."""
    o = A.rbp_solver_opts()
    lib().rbp_solver_opts_defaults(C.byref(o))
    for k, v in kw.items():
        if k == "size" or not hasattr(o, k):
            raise TypeError(f"rbp_solver_opts has no field {k!r}")
        setattr(o, k, float(v) if k == "qp_far_slack" else int(v))
    return o


def set_default_solver_opts(opts=None, **kw):
    """solver options of the calling thread's default context: what Corridor / RBPPlanner without an explicit Context use."""
    o = opts if opts is not None else solver_opts(**kw)
    rc = lib().rbp_ctx_set_solver_opts(None, C.byref(o))
    if rc:
        raise RuntimeError(f"rbp_ctx_set_solver_opts rc={rc}: {last_error()}")


class Context:
    """rbp_ctx: device memory kept across plans (include/rbp.h).  `device=None` = the calling thread's current device.
    `opts`: rbp_solver_opts (see solver_opts()) of every session / one-shot call made in the context."""

    def __init__(self, device=None, opts=None):
        self._h = C.c_void_p()
        rc = lib().rbp_ctx_create(C.byref(self._h), -1 if device is None else int(device))
        if rc:
            raise RuntimeError(f"rbp_ctx_create failed rc={rc}: {ERROR_TEXT.get(rc, '')} | {last_error()}")
        if opts is not None:
            self.set_solver_opts(opts)

    def set_solver_opts(self, opts=None, **kw):
        o = opts if opts is not None else solver_opts(**kw)
        rc = lib().rbp_ctx_set_solver_opts(self._h, C.byref(o))
        if rc:
            raise RuntimeError(f"rbp_ctx_set_solver_opts rc={rc}: {last_error()}")

    def plan_update(self, world: World, mission: Mission, param: Param, plan: PlanResult) -> int:
        """Corridor::update && RBPPlanner::update in one call; returns the C ABI's code (0 = both true)."""
        w, m, p, pl = world.c_struct(), mission.c_struct(), param.c_struct(), plan.c_struct()
        rc = lib().rbp_ctx_plan_update(self._h, C.byref(w), C.byref(m), C.byref(p), C.byref(pl))
        plan.sync_from(pl)
        return rc

    def close(self):
        if self._h:
            lib().rbp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Corridor:
    """rbp_corridor.hpp:11-26.  `ctx`: optional Context whose device arena the call reuses (default: the library's
    per-thread context)."""

    def __init__(self, world: World, mission: Mission, param: Param, ctx: Context = None):
        self.world, self.mission, self.param, self.ctx = world, mission, param, ctx
        self.last_error = ""
        self.rc = 0

    def update(self, log: bool, plan: PlanResult) -> bool:
        w, m, p, pl = self.world.c_struct(), self.mission.c_struct(), self.param.c_struct(), plan.c_struct()
        if self.ctx is not None:
            self.rc = lib().rbp_ctx_corridor_update(self.ctx._h, C.byref(w), C.byref(m), C.byref(p), C.byref(pl))
        else:
            self.rc = lib().rbp_corridor_update(C.byref(w), C.byref(m), C.byref(p), C.byref(pl))
        self.last_error = "" if self.rc == 0 else ERROR_TEXT.get(self.rc, str(self.rc)) + " | " + last_error()
        return self.rc == 0

    def update_range(self, plan: PlanResult, agent_begin: int, agent_end: int) -> bool:
        """one shard of an agent-sharded update (include/rbp.h rbp_corridor_update_range): only the SFC of agents
        [agent_begin, agent_end) and the RSFC rows of pairs (qi, qj) with qi in that range are valid afterwards."""
        w, m, p, pl = self.world.c_struct(), self.mission.c_struct(), self.param.c_struct(), plan.c_struct()
        self.rc = lib().rbp_corridor_update_range(C.byref(w), C.byref(m), C.byref(p), C.byref(pl), agent_begin, agent_end)
        self.last_error = "" if self.rc == 0 else ERROR_TEXT.get(self.rc, str(self.rc)) + " | " + last_error()
        return self.rc == 0


class RBPPlanner:
    """rbp_planner.hpp:19-84"""

    def __init__(self, mission: Mission, param: Param, ctx: "Context" = None):
        self.mission, self.param, self.ctx = mission, param, ctx
        self.last_error = ""
        self.rc = 0

    def update(self, log: bool, plan: PlanResult) -> bool:
        m, p, pl = self.mission.c_struct(), self.param.c_struct(), plan.c_struct()
        if self.ctx is not None:
            self.rc = lib().rbp_ctx_planner_update(self.ctx._h, C.byref(m), C.byref(p), C.byref(pl))
        else:
            self.rc = lib().rbp_planner_update(C.byref(m), C.byref(p), C.byref(pl))
        plan.sync_from(pl)
        self.last_error = "" if self.rc == 0 else ERROR_TEXT.get(self.rc, str(self.rc)) + " | " + last_error()
        return self.rc == 0


class Session:
    """K independent missions resident in HBM (e.g. the 50-map sweep of swarm_traj_planner_rbp_test_all.cpp:49-103).
    The missions share N; every plan keeps its own M (= ECBS makespan + 2) and max_boxes."""

    def __init__(self, worlds, missions, param: Param, plans, device=0, opts=None):
        K = len(plans)
        assert len(worlds) == K and len(missions) == K
        self.K, self.plans, self.param, self.device = K, plans, param, device
        self._keep = (worlds, missions)
        self._w = (A.rbp_world * K)(*[w.c_struct() for w in worlds])
        self._m = (A.rbp_mission * K)(*[m.c_struct() for m in missions])
        self._p = param.c_struct()
        self._pl = (A.rbp_plan * K)(*[p.c_struct() for p in plans])
        self._h = C.c_void_p()
        rc = lib().rbp_session_create(C.byref(self._h), device, K, self._w, self._m, C.byref(self._p), self._pl)
        if rc:
            raise RuntimeError(f"rbp_session_create failed rc={rc}: {ERROR_TEXT.get(rc, '')} | {last_error()}")
        if opts is not None:
            self.set_solver_opts(opts)

    def set_solver_opts(self, opts=None, **kw):
        """rbp_solver_opts of this session (before its next run): Session.set_solver_opts(qp_schedule=2) or an object from solver_opts()"""
        o = opts if opts is not None else solver_opts(**kw)
        rc = lib().rbp_session_set_solver_opts(self._h, C.byref(o))
        if rc:
            raise RuntimeError(f"rbp_session_set_solver_opts rc={rc}: {last_error()}")

    def run(self, stages=A.RBP_STAGE_ALL, stream=None):
        self._xchg_error = None   # (an exchange hook's exception belongs to the run it happened in)
        rc = lib().rbp_session_run(self._h, stages, C.c_void_p(stream or 0))
        if rc:
            cause = getattr(self, "_xchg_error", None)
            raise RuntimeError(f"rbp_session_run rc={rc}: {last_error()}" + (f" ({cause!r})" if cause is not None else ""))

    def run_async(self, stages=A.RBP_STAGE_ALL, stream=None):
        """`run` that returns at once for a grid-wide joint session too (the solve proceeds on a library thread; `wait`, `download`
        and every other call on the session wait for it): include/rbp.h rbp_session_run_async"""
        rc = lib().rbp_session_run_async(self._h, stages, C.c_void_p(stream or 0))
        if rc:
            raise RuntimeError(f"rbp_session_run_async rc={rc}: {last_error()}")

    def wait(self):
        rc = lib().rbp_session_wait(self._h)
        if rc:
            raise RuntimeError(f"rbp_session_wait rc={rc}: {last_error()}")

    def shard_joint(self, dist, group=None):
        """make this session one rank of a PAIR that shares a joint solve (include/rbp.h rbp_session_shard_joint): rank 0 of `group`
        (default: the whole process group, which must have two ranks) eliminates the lower chain of the knots, rank 1 the upper one; the
        three exchanges per interior-point iteration are all-gathers on the library's device buffers -- RCCL over xGMI under the "nccl"
        backend, through host copies under "gloo" (tests: two ranks on one GPU).  dist=None undoes it."""
        import torch
        if dist is None:
            rc = lib().rbp_session_shard_joint(self._h, 0, 1, EXCHANGE_FN(), None)
            self._xchg = None
            self._xchg_error = None
            if rc:
                raise RuntimeError(f"rbp_session_shard_joint rc={rc}: {last_error()}")
            return
        ws, rank = dist.get_world_size(group), dist.get_rank(group)
        if ws != 2:
            raise ValueError(f"a joint solve is shared by TWO ranks (the twisted elimination has two chains); the group has {ws}")
        gloo = dist.get_backend(group) == "gloo"
        sess = self
        dev = torch.device("cuda", self.device)
        self._xchg_error = None

        class _View:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

        def hook(user, send_ptr, recv_ptr, nbytes):
            try:
                n = nbytes // 8
                send = torch.as_tensor(_View(send_ptr, n), device=dev)   # (the library's buffers live on the session's device)
                recv = torch.as_tensor(_View(recv_ptr, n), device=dev)
                if gloo:
                    outs = [torch.empty(n, dtype=torch.float64) for _ in range(2)]
                    dist.all_gather(outs, send.cpu(), group=group)
                    recv.copy_(outs[1 - rank])
                else:
                    out = torch.empty(2 * n, dtype=torch.float64, device=send.device)
                    dist.all_gather_into_tensor(out, send, group=group)
                    recv.copy_(out[(1 - rank) * n:(2 - rank) * n])
                torch.cuda.synchronize(dev)
                sess.exchanges += 1
                sess.exchange_bytes += nbytes
                return 0
            except BaseException as e:  # (an exception must not unwind through the C frames of the library)
                sess._xchg_error = e
                return 1

        self.exchanges, self.exchange_bytes = 0, 0
        self._xchg = EXCHANGE_FN(hook)  # (kept alive as long as the session may call it)
        rc = lib().rbp_session_shard_joint(self._h, rank, 2, self._xchg, None)
        if rc:
            raise RuntimeError(f"rbp_session_shard_joint rc={rc}: {last_error()}")

    def set_agent_range(self, agent_begin: int, agent_end: int):
        rc = lib().rbp_session_set_agent_range(self._h, agent_begin, agent_end)
        if rc:
            raise RuntimeError(f"rbp_session_set_agent_range rc={rc}: {last_error()}")

    def reset(self, stream=None):
        rc = lib().rbp_session_reset(self._h, C.c_void_p(stream or 0))
        if rc:
            raise RuntimeError(f"rbp_session_reset rc={rc}: {last_error()}")

    def download(self, stream=None):
        """copies outputs into the PlanResult objects; returns per-mission status list."""
        import numpy as np
        st = np.zeros(self.K, np.int32)
        lib().rbp_session_download(self._h, self._pl, A.ptr(st, A.c_int32_p), C.c_void_p(stream or 0))
        for p, c in zip(self.plans, self._pl):
            p.sync_from(c)
        return st.tolist()

    def reserve_workspace(self, stream=None):
        """reserve the QP workspace now (otherwise the first PLANNER run does)"""
        rc = lib().rbp_session_reserve_workspace(self._h, C.c_void_p(stream or 0))
        if rc:
            raise RuntimeError(f"rbp_session_reserve_workspace rc={rc}: {last_error()}")

    def workspace_bytes_per_mission(self):
        return int(lib().rbp_session_workspace_bytes(self._h))

    def counters(self, stream=None):
        ct = A.rbp_counters()
        rc = lib().rbp_session_counters(self._h, C.byref(ct), C.c_void_p(stream or 0))
        if rc:
            raise RuntimeError(f"rbp_session_counters rc={rc}: {last_error()}")
        return {k: getattr(ct, k) for k, _ in ct._fields_}

    def device_arrays(self, mission=0):
        """torch tensors over mission `mission`'s corridor arrays in the session's HBM arena (no copy): dict with sfc_count [N] i32,
        sfc_box [N][MB][6] f64, sfc_time [N][MB] f64, rsfc_normal [npair][M][3] f32, rsfc_time [M] f64 (MB, M: session strides)."""
        import torch
        da = A.rbp_device_arrays()
        rc = lib().rbp_session_device_arrays(self._h, mission, C.byref(da))
        if rc:
            raise RuntimeError(f"rbp_session_device_arrays rc={rc}: {last_error()}")

        class _View:  # __cuda_array_interface__ holder (torch on ROCm reads it like on CUDA)
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}

        dev = torch.device("cuda", da.device)
        def view(ptr, shape, typestr):
            return torch.as_tensor(_View(ptr, shape, typestr), device=dev)
        N, M, MB, npair = da.N, da.M, da.max_boxes, da.npair
        return {"sfc_count": view(da.sfc_count, (N,), "<i4"), "sfc_box": view(da.sfc_box, (N, MB, 6), "<f8"),
                "sfc_time": view(da.sfc_time, (N, MB), "<f8"), "rsfc_normal": view(da.rsfc_normal, (max(npair, 1), M, 3), "<f4")[:npair],
                "rsfc_time": view(da.rsfc_time, (M,), "<f8"), "status": view(da.status, (1,), "<i4"), "_keepalive": self}

    def scalars(self, n=24, stream=None):
        import numpy as np
        out = np.zeros((self.K, n))
        rc = lib().rbp_session_scalars(self._h, A.ptr(out, A.c_double_p), n, C.c_void_p(stream or 0))
        if rc:
            raise RuntimeError(f"rbp_session_scalars rc={rc}: {last_error()}")
        return out

    def close(self):
        if self._h:
            lib().rbp_session_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_world(keys, res, param: Param, max_dist=1.0):
    """The distance grid of a world ON THE GPU (rbp_edt_build): DynamicEDTOctomap(maxDist, tree, world_min, world_max, false).update()
    (swarm_traj_planner_rbp_test_all.cpp:57-63) + getDistance on every voxel centre.  Same result, bit for bit, as host.build_world."""
    import numpy as np
    from .types import World
    keys = np.ascontiguousarray(keys, np.int32).reshape(-1, 4)
    lo = (C.c_double * 3)(param.world_x_min, param.world_y_min, param.world_z_min)
    hi = (C.c_double * 3)(param.world_x_max, param.world_y_max, param.world_z_max)
    dim, kmin = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    rc = lib().rbp_edt_dims(res, lo, hi, dim, kmin)
    if rc:
        raise ValueError(f"rbp_edt_dims rc={rc}: {last_error()}")
    dist = np.zeros(tuple(dim), np.float32)
    rc = lib().rbp_edt_build(A.ptr(keys, A.c_int32_p), len(keys), res, lo, hi, max_dist, A.ptr(dist, A.c_float_p))
    if rc:
        raise RuntimeError(f"rbp_edt_build rc={rc}: {last_error()}")
    return World(dist, tuple(kmin), res)
