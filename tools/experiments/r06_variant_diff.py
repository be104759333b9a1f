"""Developer tool (GPU box): the 50-map sweep through the library RBP_HIP_LIB names with one build of the QP kernel pinned -> per-map interior-point
iterations + control points (<out>.npz); two libraries are compared with `python tools/experiments/r06_variant_diff.py cmp a.npz b.npz`.
usage: RBP_HIP_LIB=... python tools/experiments/r06_variant_diff.py run <out.npz> [variant 2|4] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a["ctrl"] - b["ctrl"]).reshape(len(a["iters"]), -1).max(1)
    print("iterations", int(a["iters"].sum()), int(b["iters"].sum()), "maps with another count:", [(i + 1, int(x), int(y)) for i, (x, y) in enumerate(zip(a["iters"], b["iters"])) if x != y])
    print("control points: bit-identical on", int((d == 0).sum()), "of", len(d), "maps; largest difference %.3e m" % d.max())
    sys.exit(0)
import bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = int(sys.argv[4]) if len(sys.argv) > 4 else 50
p = Param.test_sweep()
m, worlds, plans = bench.build_inputs(list(range(1, K + 1)), 64, p)
s = planner.Session(worlds, [m] * K, p, plans, opts=planner.solver_opts(qp_variant=variant))
s.run(); st = s.download(); s.close()
Mmax = max(g.M for g in plans)
out = np.zeros((K, 64, 3, 6 * Mmax))
for i, g in enumerate(plans):
    out[i, :, :, :6 * g.M] = g.ctrl
np.savez(sys.argv[2], ctrl=out, iters=np.array([g.qp_iterations for g in plans]))
print("status", sorted(set(st)), "iters", sum(g.qp_iterations for g in plans), "unpolished", sum(g.qp_unpolished for g in plans))
