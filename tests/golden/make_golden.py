"""Generate the committed golden vectors  tests/golden/*.npz.

The reference holds no golden vectors for this path and cannot be built here (CPLEX, Eigen, octomap,
dynamicEDT3D, ROS absent), so these vectors are produced by the CPU oracle (oracle/) from the reference's own
input files (data/missions/*.json, data/worlds/*.bt) and pin it against regressions; each file also stores the
oracle's KKT certificate of the stored solution.  PARITY UNPINNED applies (see oracle/README.md).

Run from the repo root:   python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swarm_simulator_amd import host  # noqa: E402
from swarm_simulator_amd.types import Param, PlanResult  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def c1_init_traj():
    """C1 (BASELINE.json configs[0]): mission_4agents_15 geometry, hand-made conflict-free 4-waypoint roundabout,
    M = 3 (ECBS would give M ~ 18; a head-on swap through the origin is rejected at rbp_corridor.hpp:385)."""
    wp = np.array([
        [[4, 0, 1], [2, 2, 1], [-2, 2, 1], [-4, 0, 1]],
        [[0, 4, 1], [-2, 2, 1], [-2, -2, 1], [0, -4, 1]],
        [[-4, 0, 1], [-2, -2, 1], [2, -2, 1], [4, 0, 1]],
        [[0, -4, 1], [2, -2, 1], [2, 2, 1], [0, 4, 1]],
    ], np.float32)
    return wp, np.arange(4, dtype=np.float64)


CASES = [
    # name, mission file, agent subset (None = all), world, param kwargs, init traj source
    ("c1_4agents_empty_joint", "mission_4agents_15.json", None, "empty.bt", dict(sequential=False), "c1"),
    ("c1_4agents_empty_seq2", "mission_4agents_15.json", None, "empty.bt", dict(sequential=True, batch_size=2), "c1"),
    ("s4_map1_joint", "mission_64agents_15.json", [0, 16, 32, 48], "map1.bt", dict(sequential=False), "ecbs"),
    ("s4_map1_seq2", "mission_64agents_15.json", [0, 16, 32, 48], "map1.bt", dict(sequential=True, batch_size=2), "ecbs"),
    ("s8_map5_seq4", "mission_8agents_15.json", None, "map5.bt", dict(), "ecbs"),
    ("s8_map5_seq4_partial", "mission_8agents_15.json", None, "map5.bt", dict(batch_iter=1), "ecbs"),
    ("s8_map5_seq4_iter2", "mission_8agents_15.json", None, "map5.bt", dict(iteration=2), "ecbs"),
    ("c2_16agents_map3", "mission_16agents_15.json", None, "map3.bt", dict(), "ecbs"),
    ("c3_64agents_map1", "mission_64agents_15.json", None, "map1.bt", dict(), "ecbs"),
]


def grid_hash(w):
    return hashlib.sha256(np.ascontiguousarray(w.dist).tobytes()).hexdigest()


def make(name, mission_file, subset, world_file, pkw, src):
    p = Param.test_sweep(**pkw)
    m = host.load_mission(mission_file)
    if subset is not None:
        m = m.subset(subset)
    w = host.load_world(world_file, p)
    if src == "c1":
        traj, T = c1_init_traj()
        pr = PlanResult(traj, T)
    else:
        pr = host.ecbs_plan(w, m, p)
    init_traj, T0 = pr.init_traj.copy(), pr.T.copy()
    rc, nsamples = O.corridor_update(w, m, p, pr)
    assert rc == 0, (name, rc)
    sfc_time0, rsfc_time0 = pr.sfc_time.copy(), pr.rsfc_time.copy()
    rc, rep = O.planner_update(m, p, pr)
    assert rc == 0, (name, rc, rep)
    obj, veq, vbox, vrs = O.evaluate_ctrl(m, pr)
    big = pr.N >= 32
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        mission_file=mission_file, subset=np.array(subset if subset is not None else [], np.int32), world_file=world_file,
        param_keys=np.array(list(pkw.keys())), param_vals=np.array([float(v) for v in pkw.values()]),
        grid_sha256=grid_hash(w), init_traj=init_traj, T0=T0,
        sfc_count=pr.sfc_count, sfc_box=pr.sfc_box, sfc_time0=sfc_time0, rsfc_time0=rsfc_time0,
        rsfc_normal=(np.zeros(0, np.float32) if big else pr.rsfc_normal),
        rsfc_sha256=hashlib.sha256(pr.rsfc_normal.tobytes()).hexdigest(),
        n_samples=nsamples, ctrl=pr.ctrl, coef=pr.coef, T=pr.T, time_scale=pr.time_scale, total_cost=pr.total_cost,
        sizes=np.array([pr.x_size, pr.eq_size, pr.ineq_size]),
        kkt=np.array([rep["kkt_stationarity"], rep["kkt_primal_eq"], rep["kkt_primal_ineq"], rep["kkt_compl"],
                      rep["duality_gap_rel"]]),
        n_polished=rep["n_polished"], n_qp=rep["n_qp"], evaluate=np.array([obj, veq, vbox, vrs]))
    print(f"{name}: N={pr.N} M={pr.M} samples={nsamples} cost={pr.total_cost:.9f} time_scale={pr.time_scale:.6f} "
          f"polished {rep['n_polished']}/{rep['n_qp']} iters {rep['iters_total']} kkt {rep['kkt_stationarity']:.1e}")


if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        if only and c[0] not in only:
            continue
        make(*c)
