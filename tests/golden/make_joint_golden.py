"""Golden vectors of the JOINT QP (plan/sequential = false, the reference's code default: param.hpp:67, rbp_planner.hpp:857-859) at 32 and 64
agents: tests/golden/joint32_map7.npz, joint32_map21.npz, joint64_map3.npz, joint64_map12.npz, joint64_map30.npz.

The oracle's joint solve takes 26 s (32 agents) / 300 s (64 agents) on one core, too long for the test suites, so its answer is committed:
control points of the certified optimum (oracle_qp_report: polished, stationarity < 1e-10), the objective, and a hash of the inputs the
GPU test rebuilds (mission file, map, the repository's ECBS initTraj).  PARITY UNPINNED applies as for every oracle vector
(oracle/README.md).

Run from the repo root:   python tests/golden/make_joint_golden.py [agents:map ...]      (default: every case below)
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from swarm_simulator_amd import host  # noqa: E402
from swarm_simulator_amd.types import Param  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = [(32, 7), (32, 21), (64, 3), (64, 12), (64, 30)]


def make(n, map_id):
    p = Param.test_sweep(sequential=False)
    m = host.load_mission(f"mission_{n}agents_15.json")
    w = host.load_world(f"map{map_id}.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref = init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    t = time.time()
    rc, rep = O.planner_update(m, p, ref)
    assert rc == 0 and rep["n_qp"] == 1 and rep["n_polished"] == 1, (rc, rep)
    print(f"joint N={n} map{map_id}: M={ref.M} cost={ref.total_cost:.12f} iters={rep['iters_total']} {time.time() - t:.1f}s")
    np.savez_compressed(os.path.join(OUT, f"joint{n}_map{map_id}.npz"), ctrl=ref.ctrl, total_cost=ref.total_cost, M=ref.M,
                        iters=rep["iters_total"], kkt_stationarity=rep["kkt_stationarity"],
                        init_traj_sha256=hashlib.sha256(np.ascontiguousarray(init.init_traj).tobytes()).hexdigest())


if __name__ == "__main__":
    for n, map_id in ([tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or CASES):
        make(n, map_id)
