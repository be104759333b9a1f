"""the double-step tile sweep (joint_schedule 3) against the one-pivot bulk schedule (2) and the look-ahead schedule (1), one mission each:
control points, objective, iterations.  python tools/experiments/r05_joint_double_step_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swarm_simulator_amd import host, planner  # noqa: E402
from swarm_simulator_amd.types import Param  # noqa: E402

for n, map_id in [(8, 5), (16, 3), (32, 7), (64, 7), (64, 45)]:
    p = Param.test_sweep(sequential=False)
    m = host.load_mission(f"mission_{n}agents_15.json")
    w = host.load_world(f"map{map_id}.bt", p)
    init = host.ecbs_plan(w, m, p)
    res = {}
    for sched in (1, 2, 3):
        ctx = planner.Context(opts=planner.solver_opts(joint_wide_min_agents=2, joint_schedule=sched))
        g = init.clone_inputs()
        assert planner.Corridor(w, m, p, ctx).update(False, g)
        pl = planner.RBPPlanner(m, p, ctx)
        assert pl.update(False, g), pl.last_error
        ctx.close()
        res[sched] = g
    a = res[2]
    for sched in (1, 3):
        b = res[sched]
        print(f"N={n:3d} map{map_id:<2d} schedule {sched} vs 2: |dctrl| {np.abs(a.ctrl - b.ctrl).max():.3e}  dcost {abs(a.total_cost - b.total_cost):.3e}  "
              f"iterations {b.qp_iterations} vs {a.qp_iterations}  unpolished {b.qp_unpolished}/{a.qp_unpolished}  kkt {b.kkt_max:.2e}/{a.kkt_max:.2e}", flush=True)
