#!/usr/bin/env python
"""bench.py — agent-trajectories/sec of the RBP plan path (SFC + RSFC + QP) on MI355X.

A "step" is one pass of the hot path (Corridor::update + RBPPlanner::update equivalents) over one batch of
missions: the reference's own benchmark shape, the map sweep of swarm_traj_planner_rbp_test_all.cpp:49-103
(64-agent mission x worlds/map*.bt, launch/plan_rbp_test.launch:27-59: sequential=true, batch_size=4).  The ECBS
front-end, map loading and the EDT are NOT in the metric and run once, untimed; inputs (distance
grids, initTraj, mission) are resident in HBM when the timed region starts.

Multi-GPU: missions are independent, so ranks take disjoint slices of the sweep with no data-path collective
(weak scaling: every rank gets --missions-per-gpu missions); value = missions of all ranks * N / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD public spec for MI355X FP64 matrix (not in the local guides)


def shard_missions(n_per_rank: int, rank: int, world_size: int, n_maps: int = 50):
    """weak scaling: rank r plans maps (r*n_per_rank + i) mod n_maps, i < n_per_rank  (1-based map ids)."""
    return [((rank * n_per_rank + i) % n_maps) + 1 for i in range(n_per_rank)]


def aggregate(n_agents_local: int, seconds_local: float, dist=None):
    """whole-job throughput: sum of agent-trajectories over ranks / max time over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return n_agents_local, seconds_local
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds_local], dtype=torch.float64, device=dev)
    n = torch.tensor([float(n_agents_local)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return int(round(n.item())), float(t.item())


def build_inputs(map_ids, n_agents, param):
    from swarm_simulator_amd import host
    m = host.load_mission(f"mission_{n_agents}agents_15.json")
    worlds, plans, cache = [], [], {}
    for mid in map_ids:
        if mid not in cache:
            w = host.load_world(f"map{mid}.bt", param)
            cache[mid] = (w, host.ecbs_plan(w, m, param))
        w, pr = cache[mid]
        worlds.append(w)
        plans.append(pr.clone_inputs())
    # every map keeps its own M = ECBS makespan + 2 (ecbs_planner.hpp:41-43): the session is ragged, nothing is padded
    out = plans
    return m, worlds, out


def cpu_baseline(mission, param, world, plan, budget_s=20.0):
    """the CPU oracle (a port: CPLEX is proprietary and absent) timed on this box's host cores, one thread,
    on ONE mission of the same workload (SFC + RSFC + QP), bounded to ~budget_s."""
    from tests import oracle_lib as O
    pr = plan.clone_inputs()
    t0 = time.perf_counter()
    rc, ns = O.corridor_update(world, mission, param, pr)
    t1 = time.perf_counter()
    rc2, rep = O.planner_update(mission, param, pr)
    t2 = time.perf_counter()
    ok = rc == 0 and rc2 == 0
    return {"value": (mission.qn / (t2 - t0)) if ok else None, "unit": "agent-trajectories/s", "cores": 1, "kind": "port",
            "sample": f"1 mission ({mission.qn} agents, M={pr.M}): corridor {t1 - t0:.3f}s + planner {t2 - t1:.3f}s "
                      f"({rep['n_qp']} batch QPs, {rep['iters_total']} IPM iterations, own IPM+active-set in place of CPLEX)",
            "host_cores_available": os.cpu_count()}, ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--agents", type=int, default=64, help="mission_<N>agents_15.json (64 = headline, 16 = C2)")
    ap.add_argument("--missions-per-gpu", type=int, default=2000,
                    help="missions resident per step on each GPU: 2000 = forty passes of the reference's 50-map sweep (one "
                         "workgroup per mission, two resident per CU, the rest handed out by the dispatcher as slots free up: "
                         "four rounds keep the tail short); 50 = exactly one sweep")
    ap.add_argument("--batch-size", type=int, default=4, help="plan/batch_size (4 = plan_rbp_test.launch; 8 = BASELINE config C5)")
    ap.add_argument("--iteration", type=int, default=1, help="plan/iteration: Gauss-Seidel passes over all batches (C5: 50)")
    ap.add_argument("--joint", action="store_true", help="plan/sequential=false: one QP over all agents of a mission")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the RBP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)

    from swarm_simulator_amd import planner
    from swarm_simulator_amd import _abi as A
    from swarm_simulator_amd.types import Param

    param = Param.test_sweep(batch_size=args.batch_size, iteration=args.iteration, sequential=not args.joint)
    map_ids = shard_missions(args.missions_per_gpu, rank, world_size)
    mission, worlds, plans = build_inputs(map_ids, args.agents, param)
    K, N, M = len(plans), mission.qn, plans[0].M
    sess = planner.Session(worlds, [mission] * K, param, plans, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        sess.reset(stream)
        sess.run(A.RBP_STAGE_ALL, stream)

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    status = sess.download(stream)
    variant = os.environ.get("RBP_QP_VARIANT", "auto")
    if any(status) and variant == "auto":
        # the library picks the 128-VGPR build (two workgroups per CU) for this many missions; if that build ever fails a
        # mission the bench falls back to the 256-VGPR build rather than reporting nothing (and says so in `config`)
        os.environ["RBP_QP_VARIANT"] = "w2"
        variant = "w2 (fallback: the 128-VGPR build failed a mission)"
        step()
        torch.cuda.synchronize()
        status = sess.download(stream)
    if any(status):
        raise SystemExit(f"rank {rank}: missions failed with status {[x for x in status if x][:8]}")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # per-stage device time with HIP events on the launch stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        sess.reset(stream)
        ev[3 * i].record()
        sess.run(A.RBP_STAGE_CORRIDOR, stream)
        ev[3 * i + 1].record()
        sess.run(A.RBP_STAGE_PLANNER, stream)
        ev[3 * i + 2].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    n_total, secs = aggregate(K * N * args.steps, t1 - t0, dist)
    corridor_ms = float(np.mean([ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(args.steps)]))
    planner_ms = float(np.mean([ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(args.steps)]))
    ct = sess.counters(stream)  # of the last step
    status = sess.download(stream)

    if rank == 0:
        value = n_total / secs
        # dominant kernel: the batch QP kernel, ONE launch per step (every workgroup runs its mission's whole batch
        # schedule).  Algorithmic work per launch = flops of the dense block factorisations/solves it logs (SURVEY.md 8d:
        # F = sum_factor 7/3 nk^3 per knot + sum_solve 4 nk^2 per knot); achieved = flops per launch / planner-stage time
        # (HIP events on the launch stream; the stage is the QP launch plus three sub-millisecond helper kernels).
        qp_tflops = ct["qp_flops"] / (planner_ms * 1e-3) / 1e12
        # HBM-side bytes per qp_batch_kernel launch from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE runs of this same command, FETCH_SIZE doubled per the gfx950 correction): profiles/r01_pmc.json
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
            if pmc.get("missions_per_gpu") == K and N == 64 and args.batch_size == 4 and args.iteration == 1 and not args.joint:
                traffic = pmc["kernels"]["qp_batch_kernel"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        sfc_bytes = 4.0 * ct["sfc_samples"]
        out = {
            "metric": "agent-trajectories/sec (RBP plan: SFC+RSFC+QP)", "value": value, "unit": "agent-trajectories/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * secs / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{N}-agent random_forest mission (mission_{N}agents_15.json) on worlds/map1..50.bt, {K} "
                                   f"missions in flight per GPU ({K / 50:g} passes of the 50-map sweep), sequential={str(not args.joint).lower()} "
                                   f"batch_size={args.batch_size} iteration={args.iteration} (plan_rbp_test.launch keys)",
                       "agents": N, "segments": M, "missions_per_gpu": K, "parallelism": f"missions sharded over {world_size} GPU(s)",
                       "all_missions_ok": not any(status), "qp_kernel_variant": variant},
            "stage_ms": {"corridor": corridor_ms, "planner": planner_ms},
            "roofline": {"bound": "mfma", "kernel": "qp_batch_kernel", "achieved": qp_tflops, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": qp_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "flops_per_step": ct["qp_flops"], "ipm_iterations_per_step": ct["qp_ipm_iters"],
                         "constraint_rows_swept_per_step": ct["qp_constraint_rows"],
                         "batch_qps_per_step": ct["qp_solves"], "batch_qps_polished_per_step": ct["qp_polished"]},
            "roofline_sfc": {"bound": "hbm", "kernel": "sfc_kernel", "achieved": sfc_bytes / (corridor_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s (reference-equivalent sample bytes, not physical)",
                             "frac": sfc_bytes / (corridor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "samples_per_step": ct["sfc_samples"]},
        }
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], _ = cpu_baseline(mission, param, worlds[0], plans[0])
            except Exception as e:  # the oracle is optional equipment of the bench
                out["cpu_baseline"] = {"value": None, "unit": "agent-trajectories/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    sess.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
