#!/usr/bin/env python
"""joint (sequential=false) QP of a whole mission on the GPU vs the oracle: sup-error, cost, time.
usage: tools/gpu_joint.py [n_agents] [map_id] [--no-oracle]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mid = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = Param.test_sweep(sequential=False)
m = host.load_mission(f"mission_{n}agents_15.json")
w = host.load_world(f"map{mid}.bt", p)
init = host.ecbs_plan(w, m, p)
ref, gpu = init.clone_inputs(), init.clone_inputs()
assert planner.Corridor(w, m, p).update(False, gpu)
pl = planner.RBPPlanner(m, p)
for rep in range(2):
    g2 = gpu.clone_inputs() if rep == 0 else gpu
    if rep == 0:
        planner.Corridor(w, m, p).update(False, g2)
    t = time.time(); ok = pl.update(False, g2); dt = time.time() - t
    print(f"gpu joint N={n} M={g2.M}: ok={ok} {dt:.3f}s cost={g2.total_cost:.9f} {pl.last_error if not ok else ''}")
if "--no-oracle" not in sys.argv:
    assert O.corridor_update(w, m, p, ref)[0] == 0
    t = time.time(); rc, rep = O.planner_update(m, p, ref); dt = time.time() - t
    print(f"oracle: rc={rc} {dt:.2f}s cost={ref.total_cost:.9f} iters={rep['iters_total']} polished={rep['n_polished']}")
    print("ctrl sup-err", np.abs(ref.ctrl - gpu.ctrl).max(), "rel cost", abs(ref.total_cost - gpu.total_cost) / abs(ref.total_cost))
    print("feas", O.evaluate_ctrl(m, gpu))
