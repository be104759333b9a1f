"""Developer tool (GPU box): the reduced row set of the interior-point phase (QP_FAR_SLACK builds, RBP_HIP_LIB) on the 50-map sweep:
control points against the library's default build (every row near), iterations, algorithmic bytes, batch QPs solved twice.
usage: RBP_HIP_LIB=... python tools/experiments/r05_far_check.py <ref.npy | --write ref.npy>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep()
m, worlds, plans = bench.build_inputs(list(range(1, 51)), 64, p)
s = planner.Session(worlds, [m] * 50, p, plans)
s.run(); st = s.download(); sc = s.scalars(32); ct = s.counters(); s.close()
Mmax = max(g.M for g in plans)
out = np.zeros((50, 64, 3, 6 * Mmax))
for i, g in enumerate(plans):
    out[i, :, :, :6 * g.M] = g.ctrl
print(f"status {sorted(set(st))} QPs {sc[:, 3].sum():.0f} iters {sc[:, 2].sum():.0f} unpolished {sum(g.qp_unpolished for g in plans)} kkt_max {max(g.kkt_max for g in plans):.2e} "
      f"solved twice {sc[:, 11].sum():.0f} algorithmic GB {ct['qp_row_bytes'] / 1e9:.2f} rows swept {ct['qp_constraint_rows']:.3e}")
if sys.argv[1] == "--write":
    np.save(sys.argv[2], out)
else:
    ref = np.load(sys.argv[1])
    print(f"   vs every-row build: ctrl sup-diff {np.abs(ref - out).max():.3e} m")
