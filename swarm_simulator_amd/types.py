"""Host-side data types mirroring the reference's Mission / Param / PlanResult.

reference: swarm_planner/include/mission.hpp:10-20, param.hpp:7-42, sp_const.hpp:16-28.
Each object owns numpy buffers and hands out the flat C structs of include/rbp.h.
"""
from dataclasses import dataclass, field

import numpy as np

from . import _abi as A


@dataclass
class Mission:
    """mission.hpp:13-15: qn, startState/goalState (9 each), quad_size, max_vel, max_acc."""
    start: np.ndarray      # [N][9]
    goal: np.ndarray       # [N][9]
    radius: np.ndarray     # [N]   quad_size
    max_vel: np.ndarray    # [N][3]
    max_acc: np.ndarray    # [N][3]
    speed: np.ndarray = None

    def __post_init__(self):
        self.start = A.as_f64(self.start)
        self.goal = A.as_f64(self.goal)
        self.radius = A.as_f64(self.radius)
        self.max_vel = A.as_f64(self.max_vel)
        self.max_acc = A.as_f64(self.max_acc)

    @property
    def qn(self):
        return int(self.start.shape[0])

    N = qn

    def c_struct(self):
        m = A.rbp_mission()
        m.N = self.qn
        m.start = A.ptr(self.start, A.c_double_p)
        m.goal = A.ptr(self.goal, A.c_double_p)
        m.radius = A.ptr(self.radius, A.c_double_p)
        m.max_vel = A.ptr(self.max_vel, A.c_double_p)
        m.max_acc = A.ptr(self.max_acc, A.c_double_p)
        return m

    def subset(self, idx):
        idx = np.asarray(idx)
        return Mission(self.start[idx], self.goal[idx], self.radius[idx], self.max_vel[idx], self.max_acc[idx])


@dataclass
class Param:
    """param.hpp:44-70 — same names, same defaults."""
    log: bool = False
    world_x_min: float = -5
    world_y_min: float = -5
    world_z_min: float = 0
    world_x_max: float = 5
    world_y_max: float = 5
    world_z_max: float = 2.5
    ecbs_w: float = 1.3
    grid_xy_res: float = 0.3
    grid_z_res: float = 0.6
    grid_margin: float = 0.2
    box_xy_res: float = 0.1
    box_z_res: float = 0.1
    time_scale: bool = True
    time_step: float = 1
    downwash: float = 2.0
    n: int = 5
    phi: int = 3
    sequential: bool = False
    batch_size: int = 4
    batch_iter: int = 0
    iteration: int = 1
    timescale_rule: int = 0   # NOT a key of the reference: include/rbp.h RBP_TIMESCALE_* (0 = every real root, 1 = roots_derivative's first two eigenvalues)

    @classmethod
    def random_forest(cls, **kw):
        """launch/plan_rbp_random_forest.launch:29-66 argument defaults."""
        d = dict(world_z_min=0.3, ecbs_w=1.3, grid_xy_res=0.5, grid_z_res=1.0, grid_margin=0.2,
                 sequential=True, batch_size=4, batch_iter=-1, iteration=1)
        d.update(kw)
        return cls(**d)

    @classmethod
    def test_sweep(cls, **kw):
        """launch/plan_rbp_test.launch:27-59 (the 50-map sweep): as random_forest but ecbs_w=1.5."""
        d = dict(ecbs_w=1.5)
        d.update(kw)
        return cls.random_forest(**d)

    def c_struct(self):
        p = A.rbp_param()
        p.world_min[:] = [self.world_x_min, self.world_y_min, self.world_z_min]
        p.world_max[:] = [self.world_x_max, self.world_y_max, self.world_z_max]
        p.box_xy_res, p.box_z_res = self.box_xy_res, self.box_z_res
        p.downwash, p.time_step = self.downwash, self.time_step
        p.ecbs_w, p.grid_xy_res, p.grid_z_res, p.grid_margin = self.ecbs_w, self.grid_xy_res, self.grid_z_res, self.grid_margin
        p.n, p.phi = self.n, self.phi
        p.sequential, p.batch_size, p.batch_iter, p.iteration = int(self.sequential), self.batch_size, self.batch_iter, self.iteration
        p.time_scale, p.log = int(self.time_scale), int(self.log)
        p.timescale_rule = int(self.timescale_rule)
        return p


@dataclass
class World:
    """The distance grid DynamicEDTOctomap serves (include/rbp.h rbp_world)."""
    dist: np.ndarray            # [nx][ny][nz] float32 metres
    key_min: tuple
    res: float

    def __post_init__(self):
        self.dist = A.as_f32(self.dist)

    def c_struct(self):
        w = A.rbp_world()
        w.dim[:] = list(self.dist.shape)
        w.key_min[:] = list(self.key_min)
        w.res = self.res
        w.dist = A.ptr(self.dist, A.c_float_p)
        return w

    def c_buf(self):
        w = A.rbp_world_buf()
        w.dim[:] = list(self.dist.shape)
        w.key_min[:] = list(self.key_min)
        w.res = self.res
        w.dist = A.ptr(self.dist, A.c_float_p)
        return w


class PlanResult:
    """sp_const.hpp:21-28 as flat arrays (include/rbp.h rbp_plan)."""

    def __init__(self, init_traj, T, max_boxes=None):
        self.init_traj = A.as_f32(init_traj)          # [N][M+1][3]
        self.T = A.as_f64(T).copy()                   # [M+1]
        N, M1, _ = self.init_traj.shape
        assert M1 == self.T.shape[0]
        self.N, self.M = N, M1 - 1
        M = self.M
        self.max_boxes = int(max_boxes or M)
        self.sfc_count = np.zeros(N, np.int32)
        self.sfc_box = np.zeros((N, self.max_boxes, 6), np.float64)
        self.sfc_time = np.zeros((N, self.max_boxes), np.float64)
        self.rsfc_normal = np.zeros((A.npair(N), M, 3), np.float32)
        self.rsfc_time = np.zeros(M, np.float64)
        self.coef = np.zeros((N, 3, 6 * M), np.float64)
        self.ctrl = np.zeros((N, 3, 6 * M), np.float64)
        self.time_scale = 1.0
        self.time_scale_alt = 1.0   # the other rule's factor (rbp_param.timescale_rule)
        self.total_cost = 0.0
        self.x_size = self.eq_size = self.ineq_size = 0
        self.qp_iterations = 0
        self.qp_solves = self.qp_unpolished = 0
        self.kkt_max = 0.0
        self._c = None

    def c_struct(self):
        p = A.rbp_plan()
        p.N, p.M = self.N, self.M
        p.T = A.ptr(self.T, A.c_double_p)
        p.init_traj = A.ptr(self.init_traj, A.c_float_p)
        p.max_boxes = self.max_boxes
        p.sfc_count = A.ptr(self.sfc_count, A.c_int32_p)
        p.sfc_box = A.ptr(self.sfc_box, A.c_double_p)
        p.sfc_time = A.ptr(self.sfc_time, A.c_double_p)
        p.rsfc_normal = A.ptr(self.rsfc_normal, A.c_float_p)
        p.rsfc_time = A.ptr(self.rsfc_time, A.c_double_p)
        p.coef = A.ptr(self.coef, A.c_double_p)
        p.ctrl = A.ptr(self.ctrl, A.c_double_p)
        p.time_scale = self.time_scale
        p.total_cost = self.total_cost
        self._c = p
        return p

    def sync_from(self, p):
        """copy the scalar outputs back from the C struct after a call."""
        self.time_scale, self.total_cost = p.time_scale, p.total_cost
        self.x_size, self.eq_size, self.ineq_size = p.x_size, p.eq_size, p.ineq_size
        self.qp_iterations = p.qp_iterations
        self.qp_solves, self.qp_unpolished, self.kkt_max = p.qp_solves, p.qp_unpolished, p.kkt_max
        self.time_scale_alt = p.time_scale_alt

    def clone_inputs(self):
        return PlanResult(self.init_traj.copy(), self.T.copy(), self.max_boxes)

    def clone(self):
        q = self.clone_inputs()
        for k in ("sfc_count", "sfc_box", "sfc_time", "rsfc_normal", "rsfc_time", "coef", "ctrl"):
            getattr(q, k)[...] = getattr(self, k)
        q.time_scale, q.total_cost = self.time_scale, self.total_cost
        return q

    # ---- reference-shaped views ----------------------------------------------------------------
    def SFC(self, qi):
        """list of (box[6], end_time) like PlanResult::SFC[qi] (sp_const.hpp:17)."""
        return [(self.sfc_box[qi, b].copy(), float(self.sfc_time[qi, b])) for b in range(int(self.sfc_count[qi]))]

    def RSFC(self, qi, qj):
        """list of (normal[3] float32, time) like PlanResult::RSFC[qi][qj] (sp_const.hpp:18)."""
        p = A.pair_index(self.N, qi, qj)
        return [(self.rsfc_normal[p, m].copy(), float(self.rsfc_time[m])) for m in range(self.M)]

    def msgs_traj_info(self):
        """[N, n, T_0..T_M]   rbp_planner.hpp:270-274"""
        return np.concatenate([[self.N, 5], self.T])

    def msgs_traj_coef(self, qi):
        """data of msgs_traj_coef[qi]: column-major (6M x 3)  rbp_planner.hpp:286-289"""
        return self.coef[qi].reshape(-1)
