"""Developer tool (GPU): the same missions solved several times in one session (and over several runs) must agree bit for bit.
usage: K=512 REPS=3 python tools/determinism_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
K = int(os.environ.get("K", "512")); REPS = int(os.environ.get("REPS", "3"))
p = Param.test_sweep()
m, worlds, plans = bench.build_inputs(bench.shard_missions(K, 0, 1), 64, p)
s = planner.Session(worlds, [m] * K, p, plans)
ref_it = None
for rep in range(REPS):
    s.reset(); s.run(); st = s.download(); sc = s.scalars(28)
    it = sc[:, 2].copy()
    # copies of the same map inside the session
    bad_in = [k for k in range(50, K) if it[k] != it[k % 50]]
    msg = f"rep {rep}: iterations total {it.sum():.0f}; missions whose count differs from their first copy: {len(bad_in)} {bad_in[:10]}"
    if ref_it is not None:
        msg += f"; differs from rep 0: {int((it != ref_it).sum())}"
    else:
        ref_it = it
    print(msg, "failed:", int(np.count_nonzero(st)))
