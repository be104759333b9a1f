"""Oracle answers of the 64-agent JOINT QP (plan/sequential = false) on all 50 maps of the reference's sweep: tests/golden/joint64_sweep.npz.

The oracle needs ~5.5 min per map on one core (a process pool over the maps: ~40 min on 8 cores), so the answers are committed in compact form:
per map the objective, the iteration count, whether the oracle's own polish certified the point, M, a hash of initTraj, and the control points
of four of the 64 agents (0, 21, 42, 63) -- enough to pin the GPU's control points without 16 MB of vectors.  PARITY UNPINNED applies as for
every oracle vector (oracle/README.md).

Run from the repo root:   python tests/golden/make_joint_sweep_golden.py [workers]
"""
import hashlib
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
AGENTS = [0, 21, 42, 63]


def one(map_id):
    from swarm_simulator_amd import host
    from swarm_simulator_amd.types import Param
    from tests import oracle_lib as O
    p = Param.test_sweep(sequential=False)
    m = host.load_mission("mission_64agents_15.json")
    w = host.load_world(f"map{map_id}.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref = init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    t = time.time()
    rc, rep = O.planner_update(m, p, ref)
    dt = time.time() - t
    print(f"map{map_id}: rc={rc} M={ref.M} cost={ref.total_cost:.12f} iters={rep['iters_total']} polished={rep['n_polished']} {dt:.0f}s", flush=True)
    return dict(map=map_id, rc=rc, M=ref.M, cost=ref.total_cost, iters=rep["iters_total"], polished=rep["n_polished"],
                sha=hashlib.sha256(np.ascontiguousarray(init.init_traj).tobytes()).hexdigest(), ctrl=ref.ctrl[AGENTS].copy())


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    with ProcessPoolExecutor(workers) as ex:
        res = list(ex.map(one, range(1, 51)))
    Mmax = max(r["M"] for r in res)
    ctrl = np.full((50, len(AGENTS), 3, 6 * Mmax), np.nan)
    for i, r in enumerate(res):
        ctrl[i, :, :, :6 * r["M"]] = r["ctrl"].reshape(len(AGENTS), 3, -1)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "joint64_sweep.npz"), agents=np.array(AGENTS),
                        rc=np.array([r["rc"] for r in res]), M=np.array([r["M"] for r in res]), cost=np.array([r["cost"] for r in res]),
                        iters=np.array([r["iters"] for r in res]), polished=np.array([r["polished"] for r in res]),
                        init_traj_sha256=np.array([r["sha"] for r in res]), ctrl=ctrl)
