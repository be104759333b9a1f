"""ctypes mirror of include/rbp.h and include/rbp_host.h (struct layouts only; no compute here).

The product libraries (lib/librbp_hip.so, lib/librbp_host.so) take these structs; the test-suite's CPU checker
uses the same layouts, so a test can hand the *same* buffers to both.
"""
import ctypes as C
import os

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")

# error codes (include/rbp.h)
RBP_OK = 0
RBP_ERR_OBSTACLE_IN_INIT_TRAJ = 1
RBP_ERR_UNEQUAL_TRAJ_LEN = 2
RBP_ERR_INIT_TRAJ_COLLIDE = 3
RBP_ERR_SFC_OVERFLOW = 4
RBP_ERR_QP_FAILED = 10
RBP_ERR_UNSUPPORTED_DEGREE = 11
RBP_ERR_BAD_ARGUMENT = 20
RBP_ERR_NO_DEVICE = 30
RBP_ERR_HIP = 31
RBP_ERR_EXCHANGE = 32

RBP_ABI_VERSION = 6  # include/rbp.h

RBP_TIMESCALE_ALL_REAL_ROOTS = 0     # rbp_param.timescale_rule (include/rbp.h)
RBP_TIMESCALE_FIRST_EIGENVALUES = 1

RBP_STAGE_CORRIDOR = 1
RBP_STAGE_PLANNER = 2
RBP_STAGE_ALL = 3


class rbp_world(C.Structure):
    _fields_ = [("dim", C.c_int32 * 3), ("key_min", C.c_int32 * 3), ("res", C.c_double), ("dist", c_float_p)]


class rbp_mission(C.Structure):
    _fields_ = [("N", C.c_int32), ("start", c_double_p), ("goal", c_double_p), ("radius", c_double_p),
                ("max_vel", c_double_p), ("max_acc", c_double_p)]


class rbp_param(C.Structure):
    _fields_ = [("world_min", C.c_double * 3), ("world_max", C.c_double * 3),
                ("box_xy_res", C.c_double), ("box_z_res", C.c_double),
                ("downwash", C.c_double), ("time_step", C.c_double),
                ("ecbs_w", C.c_double), ("grid_xy_res", C.c_double), ("grid_z_res", C.c_double),
                ("grid_margin", C.c_double),
                ("n", C.c_int32), ("phi", C.c_int32), ("sequential", C.c_int32), ("batch_size", C.c_int32),
                ("batch_iter", C.c_int32), ("iteration", C.c_int32), ("time_scale", C.c_int32), ("log", C.c_int32),
                ("timescale_rule", C.c_int32)]


class rbp_plan(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("T", c_double_p), ("init_traj", c_float_p),
                ("max_boxes", C.c_int32), ("sfc_count", c_int32_p), ("sfc_box", c_double_p), ("sfc_time", c_double_p),
                ("rsfc_normal", c_float_p), ("rsfc_time", c_double_p),
                ("coef", c_double_p), ("ctrl", c_double_p),
                ("time_scale", C.c_double), ("total_cost", C.c_double),
                ("x_size", C.c_int32), ("eq_size", C.c_int32), ("ineq_size", C.c_int32), ("qp_iterations", C.c_int32),
                ("qp_solves", C.c_int32), ("qp_unpolished", C.c_int32), ("kkt_max", C.c_double), ("time_scale_alt", C.c_double)]


class rbp_counters(C.Structure):
    _fields_ = [("sfc_samples", C.c_double), ("qp_flops", C.c_double), ("qp_ipm_iters", C.c_double),
                ("qp_solves", C.c_double), ("qp_constraint_rows", C.c_double), ("qp_polished", C.c_double),
                ("qp_row_bytes", C.c_double), ("kkt_max", C.c_double)]


class rbp_device_arrays(C.Structure):
    _fields_ = [("sfc_count", C.c_void_p), ("sfc_box", C.c_void_p), ("sfc_time", C.c_void_p), ("rsfc_normal", C.c_void_p),
                ("rsfc_time", C.c_void_p), ("status", C.c_void_p), ("N", C.c_int32), ("M", C.c_int32), ("max_boxes", C.c_int32), ("npair", C.c_int32),
                ("device", C.c_int32)]


class rbp_solver_opts(C.Structure):
    _fields_ = [("size", C.c_int32), ("polish", C.c_int32), ("joint_wide_min_agents", C.c_int32), ("joint_corrector", C.c_int32),
                ("joint_schedule", C.c_int32), ("qp_schedule", C.c_int32), ("qp_variant", C.c_int32), ("qp_block_order", C.c_int32),
                ("qp_groups", C.c_int32), ("qp_rounds", C.c_int32), ("qp_far_slack", C.c_double)]


class rbp_mission_buf(C.Structure):
    _fields_ = [("N", C.c_int32), ("start", c_double_p), ("goal", c_double_p), ("radius", c_double_p),
                ("speed", c_double_p), ("max_vel", c_double_p), ("max_acc", c_double_p)]


class rbp_octomap_buf(C.Structure):
    _fields_ = [("res", C.c_double), ("n_occupied", C.c_int64), ("keys", c_int32_p), ("n_nodes", C.c_int64)]


class rbp_world_buf(C.Structure):
    _fields_ = [("dim", C.c_int32 * 3), ("key_min", C.c_int32 * 3), ("res", C.c_double), ("dist", c_float_p)]


class rbp_init_traj_buf(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("T", c_double_p), ("init_traj", c_float_p),
                ("makespan", C.c_int32), ("sum_cost", C.c_int32),
                ("high_level_expanded", C.c_int64), ("low_level_expanded", C.c_int64)]


def ptr(a, typ):
    """pointer to a C-contiguous numpy array (caller keeps `a` alive)."""
    if a is None:
        return C.cast(None, typ)
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(typ)


def npair(N):
    return N * (N - 1) // 2


def pair_index(N, qi, qj):
    """index of pair (qi<qj) in the order RSFC[qi][qj] is filled (rbp_corridor.hpp:342-344)."""
    return qi * N - qi * (qi + 1) // 2 + (qj - qi - 1)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
