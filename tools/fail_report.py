"""Developer tool: which missions of the 50-map sweep fail, and where (scalars SC_PROF0+1/+2: batch*1000 + reason, detail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep(batch_size=int(os.environ.get("BS", "4")))
m, worlds, plans = bench.build_inputs(list(range(1, 51)), int(os.environ.get("AGENTS", "64")), p)
for var in os.environ.get("VARIANTS", "w2,w4").split(","):
    s = planner.Session(worlds, [m] * 50, p, plans, opts=planner.solver_opts(qp_variant=int(var[1])))
    s.run(); st = s.download(); sc = s.scalars(28)
    print(var, "failed:", [(i + 1, st[i], plans[i].M, sc[i, 9], sc[i, 10], sc[i, 3]) for i in range(50) if st[i]])
    print("   ipm iters total", sc[:, 2].sum(), "per QP", sc[:, 2].sum() / max(sc[:, 3].sum(), 1), "unpolished", [(i + 1, pl.qp_unpolished, sc[i, 8]) for i, pl in enumerate(plans) if pl.qp_unpolished])
    s.close()
