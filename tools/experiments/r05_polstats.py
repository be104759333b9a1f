import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep()
m, worlds, plans = bench.build_inputs(list(range(1, 51)), 64, p)
s = planner.Session(worlds, [m] * 50, p, plans)
s.run(); st = s.download(); sc = s.scalars(32)
print("failed", sum(1 for x in st if x), "QPs", sc[:, 3].sum(), "iters", sc[:, 2].sum(), "per QP", sc[:, 2].sum() / sc[:, 3].sum())
print("try1 attempts", sc[:, 25].sum(), "accepted", sc[:, 26].sum(), "| try2 attempts", sc[:, 27].sum(), "accepted", sc[:, 29].sum(), "| final attempts", sc[:, 30].sum(), "accepted", sc[:, 31].sum())
s.close()
