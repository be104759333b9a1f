#!/usr/bin/env python
"""grid-wide joint QP (kernels/jqp.hip, rbp_solver_opts.joint_wide_min_agents) vs the one-workgroup joint path and (optionally) the oracle.
usage: tools/gpu_joint_wide.py [n_agents] [map_id] [--oracle] [--no-wg] [--reps R]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 16
mid = int(args[1]) if len(args) > 1 else 3
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
p = Param.test_sweep(sequential=False)
if n == 256:  # BASELINE config C4 (tools/make_mission_256.py): the world is x in [-5, 15]
    p = Param.test_sweep(sequential=False, world_x_min=-5, world_y_min=-5, world_x_max=15, world_y_max=5)
    m = host.load_mission("mission_256agents_c4.json")
else:
    m = host.load_mission(f"mission_{n}agents_15.json")
w = host.load_world(f"map{mid}.bt", p)
init = host.ecbs_plan(w, m, p)
base = init.clone_inputs()
assert planner.Corridor(w, m, p).update(False, base)
res = {}
for mode in (["1"] if "--no-wg" in sys.argv else ["1", "0"]):
    ctx = planner.Context(opts=planner.solver_opts(joint_wide_min_agents=2 if mode == "1" else 0, joint_schedule=int(os.environ.get("JOINT_SCHEDULE", "0"))))
    pl = planner.RBPPlanner(m, p, ctx)
    for rep in range(reps):
        g = base.clone()
        t = time.time(); ok = pl.update(False, g); dt = time.time() - t
        print(f"wide={mode} N={n} M={g.M}: ok={ok} {dt:.3f}s cost={g.total_cost:.9f} iters={g.qp_iterations} unpolished={g.qp_unpolished} "
              f"kkt={g.kkt_max:.2e} {pl.last_error if not ok else ''}", flush=True)
    res[mode] = g
    print("   feas (obj, eq, box, rsfc):", O.evaluate_ctrl(m, g))
if "0" in res and "1" in res:
    print("wide vs one-workgroup: ctrl sup-diff", np.abs(res["0"].ctrl - res["1"].ctrl).max(), "rel cost", abs(res["0"].total_cost - res["1"].total_cost) / abs(res["0"].total_cost))
if "--oracle" in sys.argv:
    ref = init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    t = time.time(); rc, rep = O.planner_update(m, p, ref); dt = time.time() - t
    print(f"oracle: rc={rc} {dt:.2f}s cost={ref.total_cost:.9f} iters={rep['iters_total']} polished={rep['n_polished']}")
    for mode, g in res.items():
        print(f"wide={mode} vs oracle: ctrl sup-err", np.abs(ref.ctrl - g.ctrl).max(), "rel cost", abs(ref.total_cost - g.total_cost) / abs(ref.total_cost))
