import os, sys
sys.path.insert(0, os.getcwd())
from swarm_simulator_amd import host, planner, _abi as A
from swarm_simulator_amd.types import Param
p = Param.test_sweep(sequential=False)
m = host.load_mission("mission_64agents_15.json")
ids = [int(x) for x in sys.argv[2:]]
sched = int(sys.argv[1])
worlds = [host.load_world(f"map{i}.bt", p) for i in ids]
plans = [host.ecbs_plan(w, m, p).clone_inputs() for w in worlds]
s = planner.Session(worlds, [m] * len(ids), p, plans, opts=planner.solver_opts(joint_schedule=sched))
s.run(A.RBP_STAGE_ALL); st = s.download()
for i, g in zip(ids, plans):
    print(f"map{i}: iters {g.qp_iterations} unpolished {g.qp_unpolished} kkt {g.kkt_max:.2e} cost {g.total_cost:.9f}")
