"""Developer tool: where sfc_kernel spends its time (library built with EXTRA=-DSFC_PROFILE): wall_clock64 ticks (100 MHz) summed over
the agents of a mission -- key lists (axis_keys), sample tests, everything after the mask, the mask build."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from swarm_simulator_amd import planner, _abi as A
from swarm_simulator_amd.types import Param
K = int(os.environ.get("K", "50"))
p = Param.test_sweep()
m, worlds, plans = bench.build_inputs(bench.shard_missions(K, 0, 1), 64, p)
s = planner.Session(worlds, [m] * K, p, plans)
s.run(A.RBP_STAGE_CORRIDOR); st = s.download()
sc = s.scalars()
n = 64.0
for i, name in enumerate(["key lists", "sample tests", "after mask (total per agent)", "mask build (lane 0's wave)"]):
    print(f"{name:32s} {sc[:, 20 + i].mean() / n / 1e8 * 1e6:9.1f} us per agent")
