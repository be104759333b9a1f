"""Developer tool: per-phase cycle breakdown of qp_batch_kernel (library built with EXTRA=-DQP_PROFILE)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from swarm_simulator_amd import planner, _abi as A
from swarm_simulator_amd.types import Param
K = int(os.environ.get("K", "4"))
p = Param.test_sweep(batch_size=int(os.environ.get("BS", "4")), iteration=int(os.environ.get("ITER", "1")), sequential=os.environ.get("JOINT", "0") != "1")
NA = int(os.environ.get("AGENTS", "64"))
m, worlds, plans = bench.build_inputs(bench.shard_missions(K, 0, 1), NA, p)
s = planner.Session(worlds, [m] * K, p, plans)
s.run(); st = s.download()
sc = s.scalars(32)
if os.environ.get("JSON"):  # bench.py's sweep_phase_gbs leg: one line, nothing else
    import json
    sweeps = [1, 7, 9, 10, 11]  # BUILD, AFF, STEP, NBHD, UPDATE
    print(json.dumps({"missions": K, "failed": int(np.count_nonzero(st)), "sweep_bytes": float(sc[:, 28].sum()),
                      "sweep_ticks_100mhz": float(sc[:, [8 + i for i in sweeps]].sum()), "kernel_ticks_100mhz": float(sc[:, 8:20].sum())}))
    sys.exit(0)
names = ["polish (+tail)", "BUILD sweep", "grad+FT+norms", "assemble", "factor", "rhs glue", "solve", "AFF sweep", "batch setup", "STEP sweep", "NBHD sweeps", "UPDATE sweep"]
tot = sc[:, 8:20].sum(1)
print("missions", len(st), "failed", int(np.count_nonzero(st)), "IPM iterations per mission", sc[:, 2].mean())
for i, n in enumerate(names):
    print(f"{n:18s} {sc[:, 8 + i].mean() / 1e8 * 1e3:9.2f} ms (100 MHz clock)  {100 * sc[:, 8 + i].sum() / tot.sum():5.1f} %")
print("total", tot.mean() / 1e8 * 1e3, "ms per mission")
for i, n in enumerate(["polish: candidates+K0+gradient", "polish: V columns + S", "polish: dual active-set (wave 0)", "polish: primal step/verify"]):
    print(f"  {n:34s} {sc[:, 20 + i].mean() / 1e8 * 1e3:9.2f} ms")
print(f"  left chain: MFMA update {sc[:, 25].mean() / 1e5:.2f} ms, waiting for block assembly {sc[:, 26].mean() / 1e5:.2f} ms, knot work {sc[:, 27].mean() / 1e5:.2f} ms")
if os.environ.get("LHSTATS"):
    print(f"  LH calls {sc[:, 27].mean():.1f}  appends {sc[:, 25].mean():.1f}  gradient passes {sc[:, 24].mean():.1f} (slot shared with the byte counter: subtract it)  inner solves {sc[:, 26].mean():.1f} per mission")
if os.environ.get("SOLVE_TIMERS"):  # library built with EXTRA="-DQP_SOLVE_TIMERS" (the slots of the chain-side timers then hold the substitutions' instead)
    sc = s.scalars(36)
    for who, a, b in (("left chain ", 25, 26), ("right chain", 33, 34), ("staging wave", 29, 30)):
        print(f"  substitutions, {who}: work {sc[:, a].mean() / 1e5:.2f} ms, at the step barrier {sc[:, b].mean() / 1e5:.2f} ms")
