"""Oracle answers of JOINT QPs (plan/sequential = false) on cases the solver constants of kernels/jqp.hip were never tuned on:

    tests/golden/joint_heldout.npz   mission_64agents_12.json, mission_64agents_20.json, mission_32agents_12.json x 10 maps of the
                                     sweep, and mission_64agents_15.json on ICRA2020_64agents_presentation.bt
    tests/golden/joint32_sweep.npz   mission_32agents_15.json on all 50 maps (the 32-agent counterpart of joint64_sweep.npz)

Compact form as in make_joint_sweep_golden.py: per case the objective, the iteration count, whether the oracle's own polish certified
the point, M, a hash of initTraj and the control points of four agents (0, N/3, 2N/3, N-1).  The oracle needs ~5.5 min per 64-agent
case and ~30 s per 32-agent case on one core.  PARITY UNPINNED applies as for every oracle vector (oracle/README.md).

Run from the repo root:   python tests/golden/make_joint_heldout_golden.py [workers] [heldout|sweep32|both]
"""
import hashlib
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HELD_MAPS = [2, 7, 13, 19, 23, 29, 31, 37, 43, 47]
HELDOUT = ([("mission_64agents_12.json", f"map{i}.bt") for i in HELD_MAPS] + [("mission_64agents_20.json", f"map{i}.bt") for i in HELD_MAPS] +
           [("mission_32agents_12.json", f"map{i}.bt") for i in HELD_MAPS] + [("mission_64agents_15.json", "ICRA2020_64agents_presentation.bt")])
SWEEP32 = [("mission_32agents_15.json", f"map{i}.bt") for i in range(1, 51)]


def agents_of(n):
    return [0, n // 3, (2 * n) // 3, n - 1]


CACHE = os.environ.get("JOINT_GOLDEN_CACHE", "/tmp/joint_golden_cache")  # finished cases survive an interrupted run


def one(case):
    import pickle
    os.makedirs(CACHE, exist_ok=True)
    cf = os.path.join(CACHE, f"{case[0]}__{case[1]}.pkl")
    if os.path.exists(cf):
        return pickle.load(open(cf, "rb"))
    r = one_uncached(case)
    pickle.dump(r, open(cf + ".tmp", "wb"))
    os.replace(cf + ".tmp", cf)
    return r


def one_uncached(case):
    mission, world = case
    from swarm_simulator_amd import host
    from swarm_simulator_amd.types import Param
    from tests import oracle_lib as O
    p = Param.test_sweep(sequential=False)
    m = host.load_mission(mission)
    w = host.load_world(world, p)
    try:
        init = host.ecbs_plan(w, m, p)
    except RuntimeError as e:
        print(f"{mission} {world}: ECBS found no initial trajectory ({e})", flush=True)
        return dict(mission=mission, world=world, rc=-1, M=0, cost=0.0, iters=0, polished=0, sha="", ctrl=np.zeros((4, 3, 0)), N=m.qn)
    ref = init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    t = time.time()
    rc, rep = O.planner_update(m, p, ref)
    dt = time.time() - t
    print(f"{mission} {world}: rc={rc} M={ref.M} cost={ref.total_cost:.12f} iters={rep['iters_total']} polished={rep['n_polished']} {dt:.0f}s", flush=True)
    return dict(mission=mission, world=world, rc=rc, M=ref.M, cost=ref.total_cost, iters=rep["iters_total"], polished=rep["n_polished"],
                sha=hashlib.sha256(np.ascontiguousarray(init.init_traj).tobytes()).hexdigest(), ctrl=ref.ctrl[agents_of(m.qn)].copy(), N=m.qn)


def save(name, res):
    Mmax = max(max(r["M"] for r in res), 1)
    ctrl = np.full((len(res), 4, 3, 6 * Mmax), np.nan)
    for i, r in enumerate(res):
        if r["M"] > 0:
            ctrl[i, :, :, :6 * r["M"]] = r["ctrl"].reshape(4, 3, -1)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name), mission=np.array([r["mission"] for r in res]),
                        world=np.array([r["world"] for r in res]), N=np.array([r["N"] for r in res]),
                        rc=np.array([r["rc"] for r in res]), M=np.array([r["M"] for r in res]), cost=np.array([r["cost"] for r in res]),
                        iters=np.array([r["iters"] for r in res]), polished=np.array([r["polished"] for r in res]),
                        init_traj_sha256=np.array([r["sha"] for r in res]), ctrl=ctrl)


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    what = sys.argv[2] if len(sys.argv) > 2 else "both"
    with ProcessPoolExecutor(workers) as ex:
        if what in ("heldout", "both"):
            # (the long 64-agent cases first: the pool drains evenly)
            save("joint_heldout.npz", list(ex.map(one, HELDOUT)))
        if what in ("sweep32", "both"):
            save("joint32_sweep.npz", list(ex.map(one, SWEEP32)))
