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
ids = [((first - 1 + i) % 50) + 1 for i in range(count)]
cache = {}
for i in ids:
    if i not in cache:
        w_ = host.load_world(f"map{i}.bt", p)
        cache[i] = (w_, host.ecbs_plan(w_, m, p))
worlds = [cache[i][0] for i in ids]
plans = [cache[i][1].clone_inputs() for i in ids]
sess = planner.Session(worlds, [m] * count, p, plans, opts=planner.solver_opts(joint_schedule=int(os.environ.get("JOINT_SCHEDULE", "0"))))
for rep in range(int(os.environ.get("REPS", "2"))):
    sess.reset()
    t = time.time(); sess.run(A.RBP_STAGE_ALL); st = sess.download(); dt = time.time() - t
sc = sess.scalars(12)
print(f"{count} missions in {dt:.3f}s = {count * n / dt:.0f} agent-traj/s")
for i, g in enumerate(plans):
    feas = O.evaluate_ctrl(m, g) if st[i] == 0 else None
    print(f"map{ids[i]}: status {st[i]} M {g.M} iters {g.qp_iterations} unpolished {g.qp_unpolished} kkt {g.kkt_max:.2e} cost {g.total_cost:.9f} "
          f"why {sc[i][8]:.0f} reason {sc[i][9]:.0f} it {sc[i][10]:.0f} feas {feas}")
first_of = {}
mism = 0
for i, g in enumerate(plans):
    j = first_of.setdefault(ids[i], i)
    if j != i and not np.array_equal(plans[j].ctrl.view(np.uint64), g.ctrl.view(np.uint64)):
        mism += 1
print(f"copies of a map that differ from the first copy (bitwise): {mism}; polished {sum(1 for g in plans if g.qp_unpolished == 0)} of {count}; failed {sum(1 for x in st if x)}")
sess.close()
