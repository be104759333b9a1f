"""Writer / reader of the flat case files of tests/abi_c (layout: flatfile.h).  Test infrastructure."""
import struct

import numpy as np

MAGIC = 0x46504252


def param_vector(p):
    return np.array([p.world_x_min, p.world_y_min, p.world_z_min, p.world_x_max, p.world_y_max, p.world_z_max,
                     p.box_xy_res, p.box_z_res, p.downwash, p.time_step, p.ecbs_w, p.grid_xy_res, p.grid_z_res, p.grid_margin,
                     p.n, p.phi, int(p.sequential), p.batch_size, p.batch_iter, p.iteration, int(p.time_scale), int(p.log)], dtype=np.float64)


def write_case(path, world, mission, param, plan):
    N, M, MB = mission.qn, plan.M, plan.max_boxes
    with open(path, "wb") as f:
        f.write(np.array([MAGIC, N, M, MB, *world.dist.shape, *world.key_min], dtype=np.int32).tobytes())
        f.write(struct.pack("<d", world.res))
        f.write(param_vector(param).tobytes())
        f.write(np.ascontiguousarray(world.dist, dtype=np.float32).tobytes())
        for a in (mission.start, mission.goal, mission.radius, mission.max_vel, mission.max_acc, plan.T):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(plan.init_traj, dtype=np.float32).tobytes())


def read_result(path):
    b = open(path, "rb").read()
    head = np.frombuffer(b, dtype=np.int32, count=8)
    assert head[0] == MAGIC
    rc_c, rc_p, N, M, MB, solves, unpol = (int(x) for x in head[1:])
    off = 32
    out = dict(rc_corridor=rc_c, rc_planner=rc_p, N=N, M=M, max_boxes=MB, qp_solves=solves, qp_unpolished=unpol)

    def take(name, dtype, shape):
        nonlocal off
        n = int(np.prod(shape))
        out[name] = np.frombuffer(b, dtype=dtype, count=n, offset=off).reshape(shape).copy()
        off += n * np.dtype(dtype).itemsize

    take("sfc_count", np.int32, (N,))
    take("sfc_box", np.float64, (N, MB, 6))
    take("sfc_time", np.float64, (N, MB))
    take("rsfc_normal", np.float32, (N * (N - 1) // 2, M, 3))
    take("rsfc_time", np.float64, (M,))
    take("coef", np.float64, (N, 3, 6 * M))
    take("ctrl", np.float64, (N, 3, 6 * M))
    take("T", np.float64, (M + 1,))
    take("time_scale", np.float64, (1,))
    take("total_cost", np.float64, (1,))
    assert off == len(b)
    return out
