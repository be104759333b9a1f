import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep()
K = int(sys.argv[1]); var = int(sys.argv[2])
m, worlds, plans = bench.build_inputs(bench.shard_missions(K, 0, 1), 64, p)
s = planner.Session(worlds, [m] * K, p, plans, opts=planner.solver_opts(qp_variant=var))
for rep in range(2):
    s.reset(); s.run(); st = s.download()
    print("K", K, "variant", var, "rep", rep, "status", sorted(set(st)), "iters", sum(g.qp_iterations for g in plans), flush=True)
s.close()
