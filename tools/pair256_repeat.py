"""The 256-agent joint mission of BASELINE config C4 shared by a PAIR of sessions (rbp_session_shard_joint / _stream), N times in a row on
one GPU, under a watchdog -- the case that "did not return once in four pytest runs" in round 5 and was never reproduced.  Every pass
builds fresh sessions, runs the two ranks on two threads (and, with --stream, on two streams with the stream-ordered exchange), requires
both ranks to come back within --timeout seconds and to carry the bits of the unsharded solve, and prints its time and exchange count; a
rank that does not come back is reported with both ranks' hook-call indices instead of hanging the run.

usage: python tools/pair256_repeat.py [--n 50] [--stream] [--agents 256] [--timeout 240]
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests.test_gpu_joint_shard import _Pair, _StreamPair, _same_bits
import ctypes as C


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50)
    ap.add_argument("--agents", type=int, default=256)
    ap.add_argument("--stream", action="store_true")
    ap.add_argument("--timeout", type=float, default=240.0)
    args = ap.parse_args()
    if args.agents == 256:
        p = Param.test_sweep(world_x_min=-5, world_y_min=-5, world_x_max=15, world_y_max=5, sequential=False)
        m = host.load_mission("mission_256agents_c4.json")
    else:
        p = Param.test_sweep(sequential=False)
        m = host.load_mission(f"mission_{args.agents}agents_15.json")
    w = host.load_world("map1.bt", p)
    init = host.ecbs_plan(w, m, p)
    assert planner.Corridor(w, m, p).update(False, init)
    alone = init.clone()
    s = planner.Session([w], [m], p, [alone])
    t0 = time.perf_counter()
    s.run(A.RBP_STAGE_PLANNER)
    assert s.download() == [0]
    s.close()
    print(f"unsharded: {time.perf_counter() - t0:.2f} s, {alone.qp_iterations} iterations, unpolished {alone.qp_unpolished}", flush=True)
    L = planner.lib()
    ok = 0
    for rep in range(args.n):
        pair = _StreamPair() if args.stream else _Pair()
        plans = [init.clone() for _ in range(2)]
        sessions = [planner.Session([w], [m], p, [plans[r]]) for r in range(2)]
        for r, ss in enumerate(sessions):
            if args.stream:
                rc = L.rbp_session_shard_joint_stream(ss._h, r, 2, C.cast(pair.hooks[r], C.c_void_p), None, None, args.timeout)
            else:
                rc = L.rbp_session_shard_joint(ss._h, r, 2, pair.hooks[r], None)
            assert rc == 0, planner.last_error()
        out = [None, None]

        def work(r):
            stream = pair.streams[r].cuda_stream if args.stream else None
            rc = L.rbp_session_run(sessions[r]._h, A.RBP_STAGE_PLANNER, C.c_void_p(stream or 0))
            out[r] = (rc, planner.last_error() if rc else "")
            if rc:
                pair.barrier.abort()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(2)]
        [t.start() for t in th]
        [t.join(timeout=args.timeout) for t in th]
        dt = time.perf_counter() - t0
        if any(t.is_alive() for t in th):
            pair.barrier.abort()
            print(f"pass {rep}: A RANK DID NOT RETURN within {args.timeout} s; hook calls so far {pair.calls}, results {out}", flush=True)
            sys.exit(3)
        if out[0][0] or out[1][0]:
            print(f"pass {rep}: FAILED {out}; hook calls {pair.calls}", flush=True)
            sys.exit(2)
        sts = [ss.download(pair.streams[r].cuda_stream if args.stream else None) for r, ss in enumerate(sessions)]
        [ss.close() for ss in sessions]
        same = sts == [[0], [0]] and _same_bits(plans[0], alone) and _same_bits(plans[1], alone)
        print(f"pass {rep}: {dt:.2f} s, {pair.calls[0]} exchanges, {pair.bytes[0] / 1e9:.2f} GB each way, bits of the unsharded solve: {same}", flush=True)
        if not same:
            sys.exit(4)
        ok += 1
    print(f"{ok} of {args.n} passes ok ({'stream-ordered' if args.stream else 'synchronous'} exchange)")


if __name__ == "__main__":
    main()
