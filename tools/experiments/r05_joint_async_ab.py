"""rbp_session_run_async on the 50-map joint sweep (64 agents): one session / two sessions of 25 missions in flight at once / four of
12-13.  A session's host loop synchronises once per interior-point round; with several sessions in flight one session's kernels fill the
other's synchronisation gaps.  Run on a GPU box from the repo root:   python tools/experiments/r05_joint_async_ab.py [agents]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from swarm_simulator_amd import _abi as A  # noqa: E402
from swarm_simulator_amd import planner  # noqa: E402
from swarm_simulator_amd.types import Param  # noqa: E402

agents = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = Param.test_sweep(sequential=False)
m, worlds, plans = bench.build_inputs(bench.shard_missions(50, 0, 1), agents, p)
for parts in (1, 2, 4):
    cuts = [round(50 * i / parts) for i in range(parts + 1)]
    sess = [planner.Session(worlds[a:b], [m] * (b - a), p, plans[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    best = 1e30
    for rep in range(3):
        for s in sess:
            s.reset()
        t0 = time.perf_counter()
        for s in sess:
            s.run_async(A.RBP_STAGE_ALL)
        t_call = time.perf_counter() - t0
        for s in sess:
            s.wait()
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    st = sum((s.download() for s in sess), [])
    unpol = sum(s.counters()["qp_unpolished"] for s in sess) if "qp_unpolished" in sess[0].counters() else -1
    print(f"{parts} session(s) in flight: {1e3 * best:8.1f} ms per sweep = {50 * agents / best:8.1f} agent-traj/s   calls returned after {1e3 * t_call:.2f} ms   "
          f"failed {sum(1 for x in st if x)}  unpolished {unpol}", flush=True)
    for s in sess:
        s.close()
