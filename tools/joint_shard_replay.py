"""What ONE rank of a sharded joint solve costs on a GPU of its own -- measured on a one-GPU box by record and replay.

A pair of sessions shares a joint solve (rbp_session_shard_joint); here both sit on the same GPU (two threads), so the pair's wall time
says nothing.  But the solve is deterministic: the bytes rank r receives in its k-th exchange are the same in every run.  So
  1. run the pair once and RECORD, per rank, every buffer it received (256 agents: 907 exchanges, 4.0 GB -- it stays in HBM);
  2. run rank r ALONE, its exchange hook replaying the recording (a device-to-device copy instead of the transfer over xGMI).
Step 2 is exactly the work -- kernels, host loop, synchronisations, hook calls -- rank r does on a GPU of its own; what is missing is the
wire: add bytes / link bandwidth + a collective's latency per exchange (printed as a model, NOT a measurement).  The answer of each
replayed rank is checked against the unsharded solve bit for bit.

usage: python tools/joint_shard_replay.py [--agents 256|64|32|16] [--map 1]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param


class _View:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def dev(ptr, nbytes):
    return torch.as_tensor(_View(ptr, nbytes // 8), device="cuda")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=256)
    ap.add_argument("--map", type=int, default=1)
    ap.add_argument("--xgmi-gbs", type=float, default=100.0, help="model only: bytes/s one direction of the pair's link sustains (a link's peak is ~153 GB/s)")
    ap.add_argument("--xchg-us", type=float, default=60.0, help="model only: latency of one two-rank collective on top of the hook's own cost")
    ap.add_argument("--stream", action="store_true", help="replay through the STREAM-ORDERED exchange (rbp_session_shard_joint_stream): the hook enqueues the "
                                                          "copy of the recording on the run's stream, nothing synchronises per exchange")
    args = ap.parse_args()
    if args.agents == 256:
        p = Param.test_sweep(world_x_min=-5, world_y_min=-5, world_x_max=15, world_y_max=5, sequential=False)
        m = host.load_mission("mission_256agents_c4.json")
    else:
        p = Param.test_sweep(sequential=False)
        m = host.load_mission(f"mission_{args.agents}agents_15.json")
    w = host.load_world(f"map{args.map}.bt", p)
    init = host.ecbs_plan(w, m, p)
    assert planner.Corridor(w, m, p).update(False, init)
    L = planner.lib()

    def session():
        pl = init.clone()
        return planner.Session([w], [m], p, [pl]), pl

    def timed_run(s):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(A.RBP_STAGE_PLANNER)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # unsharded (second run of a warm session)
    s, alone = session()
    timed_run(s)
    s.reset()
    t_alone = timed_run(s)
    assert s.download() == [0]
    s.close()

    # 1. the pair on this one GPU, recording what each rank receives
    barrier = threading.Barrier(2, timeout=600)
    posted = [None, None]
    rec = [[], []]

    def make_pair_hook(r):
        def hook(user, send_ptr, recv_ptr, nbytes):
            try:
                posted[r] = (send_ptr, nbytes)
                barrier.wait()
                peer_ptr, peer_bytes = posted[1 - r]
                assert peer_bytes == nbytes
                got = dev(peer_ptr, nbytes).clone()
                dev(recv_ptr, nbytes).copy_(got)
                rec[r].append(got)
                torch.cuda.synchronize()
                barrier.wait()
                return 0
            except BaseException as e:
                print("hook failed:", repr(e), file=sys.stderr)
                barrier.abort()
                return 1
        return hook

    hooks = [planner.EXCHANGE_FN(make_pair_hook(r)) for r in range(2)]
    pair = [session() for _ in range(2)]
    for r, (s, _) in enumerate(pair):
        assert L.rbp_session_shard_joint(s._h, r, 2, hooks[r], None) == 0, planner.last_error()
    t_pair = [0.0, 0.0]

    def work(r):
        t_pair[r] = timed_run(pair[r][0])

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for s, _ in pair:
        assert s.download() == [0]
        s.close()

    def same(a, b):
        return bool(np.array_equal(a.ctrl.view(np.uint64), b.ctrl.view(np.uint64)) and a.total_cost == b.total_cost and a.qp_iterations == b.qp_iterations)

    assert same(pair[0][1], alone) and same(pair[1][1], alone)

    # 2. each rank alone, the peer replayed
    out = {"agents": m.qn, "M": int(alone.M), "map": args.map, "iterations": int(alone.qp_iterations), "unsharded_s": t_alone,
           "pair_on_one_gpu_s": max(t_pair), "exchanges": len(rec[0]), "exchange_bytes_each_way": int(sum(t.numel() * 8 for t in rec[0]))}
    for r in range(2):
        pos = [0]
        hook_s = [0.0]

        def replay(user, send_ptr, recv_ptr, nbytes, r=r, pos=pos, hook_s=hook_s):
            t0 = time.perf_counter()
            got = rec[r][pos[0]]
            pos[0] += 1
            if got.numel() * 8 != nbytes:
                return 3
            dev(recv_ptr, nbytes).copy_(got)
            torch.cuda.synchronize()
            hook_s[0] += time.perf_counter() - t0
            return 0

        def replay_stream(user, send_ptr, recv_ptr, nbytes, stream_ptr, r=r, pos=pos, hook_s=hook_s):
            t0 = time.perf_counter()
            got = rec[r][pos[0]]
            pos[0] += 1
            if got.numel() * 8 != nbytes:
                return 3
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream_ptr)) if stream_ptr else torch.cuda.default_stream()):
                dev(recv_ptr, nbytes).copy_(got, non_blocking=True)
            hook_s[0] += time.perf_counter() - t0
            return 0

        s, pl = session()
        if args.stream:
            import ctypes as C
            hk = planner.EXCHANGE_STREAM_FN(replay_stream)
            assert L.rbp_session_shard_joint_stream(s._h, r, 2, C.cast(hk, C.c_void_p), None, None, 600.0) == 0
        else:
            hk = planner.EXCHANGE_FN(replay)
            assert L.rbp_session_shard_joint(s._h, r, 2, hk, None) == 0
        timed_run(s)          # warm-up (also replays)
        s.reset()
        pos[0] = 0
        hook_s[0] = 0.0
        t = timed_run(s)
        assert s.download() == [0] and pos[0] == len(rec[r])
        s.close()
        out[f"rank{r}_alone_replayed_s"] = t
        out[f"rank{r}_hook_s"] = hook_s[0]
        out[f"rank{r}_same_bits_as_unsharded"] = same(pl, alone)
    wire = out["exchange_bytes_each_way"] / (args.xgmi_gbs * 1e9) + out["exchanges"] * args.xchg_us * 1e-6
    slow = max(out["rank0_alone_replayed_s"], out["rank1_alone_replayed_s"])
    out["exchange"] = "stream-ordered (one host synchronisation per interior-point round)" if args.stream else "synchronous hook (a host synchronisation per exchange)"
    if args.stream:
        args.xchg_us = min(args.xchg_us, 20.0)  # (no host round trip per exchange: what is left is the collective's own latency on the stream)
    wire = out["exchange_bytes_each_way"] / (args.xgmi_gbs * 1e9) + out["exchanges"] * args.xchg_us * 1e-6
    out["model"] = {"what": "slower replayed rank + bytes / link bandwidth + exchanges x collective latency: an ESTIMATE of the two-GPU time, not a measurement",
                    "xgmi_gbs_assumed": args.xgmi_gbs, "collective_latency_us_assumed": args.xchg_us, "wire_s": wire,
                    "two_gpu_s_estimated": slow + wire, "speedup_estimated": t_alone / (slow + wire)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
