"""Developer tool (GPU box): control points of K missions of the headline workload through the library RBP_HIP_LIB names -> <out>.npy
(two libraries that must agree bit for bit: run twice, compare with numpy).  usage: python tools/experiments/r05_dump_ctrl.py <out> [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
p = Param.test_sweep()
m, worlds, plans = bench.build_inputs(list(range(1, K + 1)), 64, p)
s = planner.Session(worlds, [m] * K, p, plans)
s.run(); st = s.download(); s.close()
Mmax = max(g.M for g in plans)
out = np.zeros((K, 64, 3, 6 * Mmax))
for i, g in enumerate(plans):
    out[i, :, :, :6 * g.M] = g.ctrl
np.save(sys.argv[1], out)
print("status", sorted(set(st)), "iters", sum(g.qp_iterations for g in plans), "unpolished", sum(g.qp_unpolished for g in plans))
