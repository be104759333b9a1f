"""Developer tool: which batch QPs of the 50-map sweep keep the interior-point answer, and why (scalars slot SC_PROF0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep(batch_size=int(os.environ.get("BS", "4")))
m, worlds, plans = bench.build_inputs(list(range(1, 51)), int(os.environ.get("AGENTS", "64")), p)
for var in ("w2", "w4"):
    s = planner.Session(worlds, [m] * 50, p, plans, opts=planner.solver_opts(qp_variant=int(var[1])))
    s.run(); st = s.download(); sc = s.scalars(28)
    print(var, "status", [x for x in st if x], "unpolished:", [(i + 1, pl.qp_unpolished, sc[i, 8], f"{pl.kkt_max:.1e}") for i, pl in enumerate(plans) if pl.qp_unpolished])
    print("   ipm iters total", sc[:, 2].sum(), "per QP", sc[:, 2].sum() / sc[:, 3].sum())
    s.close()
