"""Developer tool (GPU box): ONE joint mission (mission file, world file), e.g. with the developer library and RBP_JOINT_TRACE=1:
RBP_HIP_LIB=$PWD/swarm_simulator_amd/lib/librbp_hip_dev.so RBP_JOINT_TRACE=1 python tools/experiments/r05_joint_one.py mission_64agents_20.json map19.bt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep(sequential=False)
m = host.load_mission(sys.argv[1])
w = host.load_world(sys.argv[2], p)
g = host.ecbs_plan(w, m, p).clone_inputs()
assert planner.Corridor(w, m, p).update(False, g)
pl = planner.RBPPlanner(m, p)
t = time.time(); ok = pl.update(False, g); dt = time.time() - t
print(f"{sys.argv[1]} {sys.argv[2]}: ok={ok} {dt:.3f}s M={g.M} cost={g.total_cost:.9f} iters {g.qp_iterations} unpolished {g.qp_unpolished} kkt {g.kkt_max:.2e} {pl.last_error}")
