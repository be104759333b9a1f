"""A/B: the 2000-mission step as ONE session on one stream against G sessions of 2000 / G missions on G streams (the corridor stage of one
group overlaps the planner stage of another, the tail of one group's qp_batch_kernel is filled by the next group's workgroups).
usage: python tools/ab_two_sessions.py [--groups 2] [--steps 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from swarm_simulator_amd import planner, _abi as A
from swarm_simulator_amd.types import Param

ap = argparse.ArgumentParser()
ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4])
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--missions", type=int, default=2000)
ap.add_argument("--interleave", action="store_true", help="group g takes missions g, g + G, ... instead of a contiguous slice")
args = ap.parse_args()
param = Param.test_sweep(batch_size=4, iteration=1, sequential=True)
maps = bench.shard_missions(args.missions, 0, 1)
mission, worlds, plans = bench.build_inputs(maps, 64, param)
K = len(plans)
out = {}
for G in args.groups:
    idx = [list(range(g, K, G)) if args.interleave else list(range(g * K // G, (g + 1) * K // G)) for g in range(G)]
    sess = [planner.Session([worlds[i] for i in ix], [mission] * len(ix), param, [plans[i] for i in ix]) for ix in idx]
    streams = [torch.cuda.Stream() for _ in range(G)] if G > 1 else [torch.cuda.current_stream()]
    for s, st in zip(sess, streams):
        s.reserve_workspace(st.cuda_stream)

    def step():
        for s, st in zip(sess, streams):
            s.reset(st.cuda_stream)
            s.run(A.RBP_STAGE_ALL, st.cuda_stream)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ok = all(not any(s.download(st.cuda_stream)) for s, st in zip(sess, streams))
    out[G] = {"ms_per_step": 1e3 * dt, "value": K * 64 / dt, "ok": ok}
    for s in sess:
        s.close()
    print(json.dumps({"groups": G, **out[G]}), flush=True)
