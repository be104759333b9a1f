#!/usr/bin/env python
"""joint QP (plan/sequential=false) of the N-agent mission on maps [first, first+count) in ONE session: outcome per map.
usage: tools/gpu_joint_sweep.py [n_agents] [first_map] [count]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from swarm_simulator_amd import host, planner, _abi as A
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
count = int(sys.argv[3]) if len(sys.argv) > 3 else 8
p = Param.test_sweep(sequential=False)
m = host.load_mission(f"mission_{n}agents_15.json")
worlds = [host.load_world(f"map{i}.bt", p) for i in range(first, first + count)]
plans = [host.ecbs_plan(w, m, p) for w in worlds]
sess = planner.Session(worlds, [m] * count, p, plans)
for rep in range(int(os.environ.get("REPS", "2"))):
    sess.reset()
    t = time.time(); sess.run(A.RBP_STAGE_ALL); st = sess.download(); dt = time.time() - t
sc = sess.scalars(12)
print(f"{count} missions in {dt:.3f}s = {count * n / dt:.0f} agent-traj/s")
for i, g in enumerate(plans):
    feas = O.evaluate_ctrl(m, g) if st[i] == 0 else None
    print(f"map{first + i}: status {st[i]} M {g.M} iters {g.qp_iterations} unpolished {g.qp_unpolished} kkt {g.kkt_max:.2e} cost {g.total_cost:.9f} "
          f"why {sc[i][8]:.0f} reason {sc[i][9]:.0f} it {sc[i][10]:.0f} feas {feas}")
sess.close()
