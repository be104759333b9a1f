#!/usr/bin/env python
"""bench.py — agent-trajectories/sec of the RBP plan path (SFC + RSFC + QP) on MI355X.

A "step" is one pass of the hot path (Corridor::update + RBPPlanner::update equivalents) over one batch of
missions: the reference's own benchmark shape, the map sweep of swarm_traj_planner_rbp_test_all.cpp:49-103
(64-agent mission x worlds/map*.bt, launch/plan_rbp_test.launch:27-59: sequential=true, batch_size=4).  Every map keeps
its own M = ECBS makespan + 2 (ecbs_planner.hpp:41-43): the session is ragged, nothing is padded.  The ECBS
front-end, map loading and the EDT are NOT in the metric and run once, untimed; inputs (distance
grids, initTraj, mission) are resident in HBM when the timed region starts.

Multi-GPU: missions are independent, so ranks take disjoint slices of the sweep with no data-path collective
(weak scaling: every rank gets --missions-per-gpu missions); value = missions of all ranks * N / max-over-ranks time.
`--gpus N` without a torchrun environment re-executes itself under torch.distributed.run with N ranks.

`--config c4` times ONE 256-agent mission whose Corridor::update is sharded by agent over the ranks (one fused RCCL
all-gather) followed by the planner sweep (BASELINE.json config 4).  With `--joint` the mission is ONE joint QP on the grid-wide solver,
and on an even number of ranks PAIRS of ranks share its knot elimination (rbp_session_shard_joint; swarm_simulator_amd/sharded.py) --
`--gpus 2 --backend gloo` runs that pair on a one-GPU box as a correctness line (both ranks on the one device: its time means nothing).

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD public spec for MI355X FP64 matrix (not in the local guides)

CONFIGS = {  # BASELINE.json configs -> flags (C1 is the CPU plumbing case of the test-suite)
    "c2": dict(agents=16), "c3": dict(agents=64), "c5": dict(agents=64, batch_size=8, iteration=50, missions_per_gpu=250),
    "c4": dict(agents=256),
}


PMC_FILE = "profiles/r06_pmc.json"   # HBM-side bytes per launch, collected by tools/collect_profiles.sh on the sources hashed below


def kernel_source_sha(variant="auto"):
    """hash of everything that decides what the device executes: the HIP sources of the kernels AND of the ABI layer (which picks
    the launched build), the Makefile (flags, QP_WAVES_PER_EU of the two builds) and the variant override in force.  Ties a
    profiles/*_pmc.json to the code it measured."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "swarm_simulator_amd", "csrc")
    files = [os.path.join(base, "Makefile")]
    for sub in ("kernels", "abi"):
        d = os.path.join(base, sub)
        # (kernels/jqp*: the grid-wide joint solver, a separate set of kernels that the batch workload never launches)
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if not f.startswith("jqp")]
    for f in files:
        h.update(os.path.relpath(f, base).encode())
        h.update(open(f, "rb").read())
    h.update(("variant=" + variant).encode())
    return h.hexdigest()[:16]


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
        plans.append(pr.clone_inputs())   # every map keeps its own M = makespan + 2: a ragged session, nothing is padded
    return m, worlds, plans


# ---- CPU baseline: the oracle (a port: CPLEX is proprietary and absent), timed on this box's host cores -------------------------
def _cpu_one(args):
    """one mission of the workload on one core: corridor + planner of the oracle.  Returns (agents, seconds, stage split, report)."""
    mid, n_agents, pkw = args
    from swarm_simulator_amd import host
    from swarm_simulator_amd.types import Param
    from tests import oracle_lib as O
    p = Param.test_sweep(**pkw)
    m = host.load_mission(f"mission_{n_agents}agents_15.json")
    w = host.load_world(f"map{mid}.bt", p)
    pr = host.ecbs_plan(w, m, p)
    t0 = time.perf_counter()
    rc, _ = O.corridor_update(w, m, p, pr)
    t1 = time.perf_counter()
    rc2, rep = O.planner_update(m, p, pr)
    t2 = time.perf_counter()
    return m.qn, t2 - t0, t1 - t0, t2 - t1, rc == 0 and rc2 == 0, rep["n_qp"], rep["iters_total"], pr.M


def cplex_probe():
    """is there a CPLEX installation whose Concert path could be timed beside the port?  Looks where the reference's
    CMakeLists.txt:38-47 looks (CPLEX_PREFIX_DIR) and in IBM's default prefix.  Reports only; nothing is built."""
    cands = [os.environ.get("CPLEX_PREFIX_DIR"), "/opt/ibm/ILOG", "/opt/ibm", os.path.expanduser("~/ibm/ILOG")]
    for c in cands:
        if not c or not os.path.isdir(c):
            continue
        for root, _, files in os.walk(c):
            if "ilocplex.h" in files:
                return f"found {os.path.join(root, 'ilocplex.h')} (Concert path not built: reference sources / Eigen / octomap absent)"
    return "no CPLEX installation (CPLEX_PREFIX_DIR unset, /opt/ibm absent): CPU baseline is the port only"


def cpu_baseline(n_agents, pkw, budget_s=25.0):
    """measured twice: ONE thread on one mission of the workload, and ALL host cores with one mission per core
    (the missions of the sweep are independent: a process pool over maps), bounded to about `budget_s` of wall time."""
    from concurrent.futures import ProcessPoolExecutor
    one = _cpu_one((1, n_agents, pkw))
    n1, s1, sc, sp, ok, nqp, its, M = one
    out = {"value": (n1 / s1) if ok else None, "unit": "agent-trajectories/s", "cores": 1, "kind": "port",
           "sample": f"1 mission (map1, {n1} agents, M={M}): corridor {sc:.3f}s + planner {sp:.3f}s ({nqp} batch QPs, {its} IPM "
                     f"iterations, own IPM+active-set in place of CPLEX)"}
    # the cores this process may actually run on (a container's CPU set / quota can be far below os.cpu_count())
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except Exception:
        pass
    workers = max(1, min(cores, 64))   # bounded sample: at most 64 processes, one mission each
    n_missions = workers
    try:
        t0 = time.perf_counter()
        with ProcessPoolExecutor(max_workers=workers) as ex:
            res = list(ex.map(_cpu_one, [((i % 50) + 1, n_agents, pkw) for i in range(n_missions)], chunksize=1))
        dt = time.perf_counter() - t0
        good = [r for r in res if r[4]]
        # the metric's stages only (Corridor::update + RBPPlanner::update inside each worker, all workers running concurrently):
        # agents of all workers / the slowest worker's stage time.  The pool's wall time also holds interpreter start, grid build
        # and ECBS of every worker -- stages the GPU figure excludes too -- and is kept as wall_incl_setup.
        slowest = max(r[1] for r in good) if good else float("nan")
        out["all_cores"] = {"value": sum(r[0] for r in good) / slowest, "unit": "agent-trajectories/s", "cores": workers,
                            # an UPPER BOUND of the CPU rate, not a measurement of it: it assumes every worker's stage time overlaps
                            # the others' perfectly (the pool staggers their starts); the measured figure is wall_incl_setup
                            "value_is": "upper bound (perfect overlap of the workers' stage times assumed)",
                            "wall_incl_setup": sum(r[0] for r in good) / dt,
                            "sample": f"{len(good)} missions (maps 1..{min(50, n_missions)}, one per process, {workers} processes running "
                                      f"concurrently): corridor+planner of the slowest worker {slowest:.2f}s; pool wall time {dt:.1f}s "
                                      f"incl. process start, grid build and ECBS of each worker"}
    except Exception as e:  # the pool is optional equipment of the bench
        out["all_cores"] = {"value": None, "cores": workers, "sample": f"failed: {e}"}
    out["cplex_probe"] = cplex_probe()
    out["host_cores_available"] = cores
    out["os_cpu_count"] = os.cpu_count()
    return out


def single_mission_latency(mission, param, world, plan, reps=3):
    """the reference's call shape: ONE mission through the two synchronous host-buffer calls (upload + kernels + download each),
    device memory kept in a context.  Returns milliseconds (min over reps) for corridor, planner and the fused one-call form."""
    from swarm_simulator_amd import planner
    ctx = planner.Context()
    best = [1e30, 1e30, 1e30]
    for _ in range(reps):
        pr = plan.clone_inputs()
        t0 = time.perf_counter()
        ok1 = planner.Corridor(world, mission, param, ctx).update(False, pr)
        t1 = time.perf_counter()
        ok2 = planner.RBPPlanner(mission, param, ctx).update(False, pr)
        t2 = time.perf_counter()
        pr2 = plan.clone_inputs()
        rc = ctx.plan_update(world, mission, param, pr2)
        t3 = time.perf_counter()
        if not (ok1 and ok2 and rc == 0):
            return None
        best = [min(best[0], 1e3 * (t1 - t0)), min(best[1], 1e3 * (t2 - t1)), min(best[2], 1e3 * (t3 - t2))]
    ctx.close()
    return {"corridor_update_ms": best[0], "planner_update_ms": best[1], "two_calls_ms": best[0] + best[1], "fused_plan_update_ms": best[2],
            "note": "one mission (map1) alone on the GPU, host buffers in and out (H2D/D2H included), min of %d" % reps}


def single_sweep_rate(mission_file_agents, param, steps=3):
    """the reference's real call shape at scale: ONE pass of the 50-map sweep (swarm_traj_planner_rbp_test_all.cpp:49-51) resident, i.e.
    50 workgroups on 256 CUs.  Returns agent-trajectories/s and ms per sweep (device-resident inputs, like the headline)."""
    import torch
    from swarm_simulator_amd import planner
    from swarm_simulator_amd import _abi as A
    m, worlds, plans = build_inputs(shard_missions(50, 0, 1), mission_file_agents, param)
    sess = planner.Session(worlds, [m] * 50, param, plans)
    stream = torch.cuda.current_stream().cuda_stream
    sess.reset(stream), sess.run(A.RBP_STAGE_ALL, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sess.reset(stream)
        sess.run(A.RBP_STAGE_ALL, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = not any(sess.download(stream))
    sess.close()
    return {"value": 50 * m.qn / dt if ok else None, "unit": "agent-trajectories/s", "ms_per_sweep": 1e3 * dt,
            "note": "50 missions resident (map1..50, one workgroup each): one pass of the reference's own sweep"}


def joint_leg(agents=64):
    """The JOINT QP (plan/sequential = false, the reference's code default param.hpp:67; kernels/jqp.hip) under the driver's clock:
    one 64-agent mission through the two synchronous calls (map1), and one pass of the 50-map sweep resident in a session.  Returns the
    top-level scalars and the roofline_joint block (bound mfma: flops the solver logs for its tile sweeps and substitutions / planner-stage
    time from HIP events; the dominant kernel's own rate from the committed rocprofv3 summary, tied to the sources by hash)."""
    import torch
    from swarm_simulator_amd import planner
    from swarm_simulator_amd import _abi as A
    from swarm_simulator_amd.types import Param
    p = Param.test_sweep(sequential=False)
    m, worlds, plans = build_inputs(shard_missions(50, 0, 1), agents, p)
    ctx = planner.Context()
    best = 1e30
    for _ in range(2):
        pr = plans[0].clone_inputs()
        t0 = time.perf_counter()
        ok = planner.Corridor(worlds[0], m, p, ctx).update(False, pr) and planner.RBPPlanner(m, p, ctx).update(False, pr)
        best = min(best, 1e3 * (time.perf_counter() - t0))
        if not ok:
            return {"error": "joint mission failed"}
    single = {"two_calls_ms": best, "iterations": pr.qp_iterations, "unpolished": pr.qp_unpolished, "kkt_max": pr.kkt_max}
    ctx.close()
    sess = planner.Session(worlds, [m] * 50, p, plans)
    stream = torch.cuda.current_stream().cuda_stream
    sess.run(A.RBP_STAGE_ALL, stream)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    sess.reset(stream)
    t0 = time.perf_counter()
    ev[0].record(); sess.run(A.RBP_STAGE_CORRIDOR, stream); ev[1].record(); sess.run(A.RBP_STAGE_PLANNER, stream); ev[2].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    planner_ms = ev[1].elapsed_time(ev[2])
    ct = sess.counters(stream)
    status = sess.download(stream)
    ws = sess.workspace_bytes_per_mission()
    sess.close()
    # the same sweep as two sessions of 25 missions in flight at once (rbp_session_run_async: each solve on its session's own thread and
    # stream; one session's kernels fill the other's once-per-round synchronisation gaps)
    halves = [planner.Session(worlds[a:b], [m] * 25, p, plans[a:b]) for a, b in ((0, 25), (25, 50))]
    asyn = None
    for rep in range(2):
        for s2 in halves:
            s2.reset(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s2 in halves:
            s2.run_async(A.RBP_STAGE_ALL, stream)
        t_call = time.perf_counter() - t0
        for s2 in halves:
            s2.wait()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        st2 = sum((s2.download(stream) for s2 in halves), [])
        asyn = {"sessions_in_flight": 2, "value": (50 * m.qn / t_all) if not any(st2) else None, "ms": 1e3 * t_all, "calls_return_after_ms": 1e3 * t_call}
    for s2 in halves:
        s2.close()
    tflops = ct["qp_flops"] / (planner_ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": "jq_update(_bulk) + jq_panel (joint QP, kernels/jqp.hip)", "achieved": tflops, "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "flops_per_sweep": ct["qp_flops"],
            "ipm_iterations_per_sweep": ct["qp_ipm_iters"], "qps_polished": ct["qp_polished"], "kkt_max": ct["kkt_max"],
            "note": "whole planner stage of 50 resident 64-agent joint missions (sweeps, factorisations, substitutions, polish): a lower bound "
                    "of the update kernel's own rate, which kernel_profiled carries"}
    try:
        jk = json.load(open(os.path.join(ROOT, "profiles", "r06_joint_kernel.json")))
        base = os.path.join(ROOT, "swarm_simulator_amd", "csrc", "kernels")
        h = hashlib.sha256()
        for f in sorted(os.listdir(base)):
            if f.startswith("jqp"):
                h.update(f.encode()), h.update(open(os.path.join(base, f), "rb").read())
        if jk["joint_source_sha"] == h.hexdigest()[:16]:
            roof["kernel_profiled"] = {"kernel": jk["kernel"], "achieved": jk["tflops"], "unit": "TFLOP/s", "frac": jk["frac_of_fp64_mfma_peak"],
                                       "share_of_logged_flops": jk["share_of_logged_flops_in_this_kernel"], "missions_per_gpu": jk["missions_per_gpu"],
                                       "source": "profiles/r06_joint_kernel.json"}
            # HBM-side bytes per launch of that kernel (FETCH_SIZE / WRITE_SIZE passes of the same command, gfx950 correction applied by
            # tools/joint_kernel_json.py): a committed constant tied to the joint solver's sources by the hash above, like traffic_profiled
            if jk.get("traffic"):
                roof["traffic"] = jk["traffic"]["hbm_bytes_per_launch"]
                roof["traffic_note"] = (f"per launch of {jk['kernel']} at {jk['missions_per_gpu']} resident missions (profiled constant, "
                                        "profiles/r06_joint_kernel.json); this run's sweep keeps 50 resident")
    except Exception:
        pass
    return {"joint_single_mission_ms": single["two_calls_ms"], "joint_single_mission": single,
            "joint_sweep_value": (50 * m.qn / dt) if not any(status) else None, "joint_sweep_ms": 1e3 * dt,
            "joint_sweep_async": asyn, "joint_workspace_bytes_per_mission": ws, "roofline_joint": roof}


def sweep_phase_rate(K, agents, resident_wgs=512):
    """roofline.sweep_phase_gbs: the streaming part of qp_batch_kernel judged on its own.  A subprocess runs the SAME workload through the
    developer build of the library that carries the in-kernel phase timers (lib/librbp_hip_prof.so, `make prof`; 100 MHz wall clock per
    workgroup, tools/qp_phase_profile.py) and reports, summed over the missions, the bytes the row sweeps stream (rbp_dev.h
    SC_SWEEP_BYTES: the sweep part of the algorithmic bytes) and the time the workgroups spend inside their sweep phases.  With
    `resident_wgs` workgroups on the chip at a time (two per CU), chip-level rate = resident * bytes / time."""
    import subprocess
    lib = os.path.join(ROOT, "swarm_simulator_amd", "lib", "librbp_hip_prof.so")
    if not os.path.exists(lib):
        return None
    env = dict(os.environ, RBP_HIP_LIB=lib, K=str(K), AGENTS=str(agents), BS="4", ITER="1", JSON="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "qp_phase_profile.py")], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode or not line:
        return None
    d = json.loads(line[-1])
    if d["failed"] or d["sweep_ticks_100mhz"] <= 0:
        return None
    res = min(resident_wgs, d["missions"])
    d["resident_workgroups"] = res
    d["gbs"] = res * d["sweep_bytes"] / (d["sweep_ticks_100mhz"] / 1e8) / 1e9
    d["sweep_share_of_kernel_time"] = d["sweep_ticks_100mhz"] / d["kernel_ticks_100mhz"]
    return d


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` outside a torchrun environment: run the same command line with N ranks, one per GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="BASELINE.json configuration shortcut (sets the flags below)")
    ap.add_argument("--agents", type=int, default=64, help="mission_<N>agents_15.json (64 = headline, 16 = C2)")
    ap.add_argument("--missions-per-gpu", type=int, default=None,
                    help="missions resident per step on each GPU: 2000 = forty passes of the reference's 50-map sweep (one "
                         "workgroup per mission, two resident per CU, the rest handed out by the dispatcher as slots free up: "
                         "four rounds keep the tail short); 50 = exactly one sweep.  Default 2000; with --joint at 16 agents or more 200 "
                         "(the grid-wide joint solver keeps every knot's explicit inverse: ~0.35 GB of workspace per 64-agent mission)")
    ap.add_argument("--batch-size", type=int, default=4, help="plan/batch_size (4 = plan_rbp_test.launch; 8 = BASELINE config C5)")
    ap.add_argument("--iteration", type=int, default=1, help="plan/iteration: Gauss-Seidel passes over all batches (C5: 50)")
    ap.add_argument("--joint", action="store_true", help="plan/sequential=false: one QP over all agents of a mission")
    ap.add_argument("--qp-schedule", choices=["auto", "mono", "phase"], default="auto",
                    help="rbp_solver_opts.qp_schedule: one workgroup per mission runs a mission's whole schedule (mono = the default) / the "
                         "phase-split schedule with chip-wide row sweeps (kernels/qp_phase.inc: developer build only, RBP_HIP_LIB=.../librbp_hip_dev.so); A/B runs")
    ap.add_argument("--qp-groups", type=int, default=0, help="rbp_solver_opts.qp_groups (phase split: streams)")
    ap.add_argument("--qp-variant", choices=["auto", "w2", "w4"], default="auto", help="rbp_solver_opts.qp_variant (A/B runs)")
    ap.add_argument("--qp-far-slack", type=float, default=None, help="rbp_solver_opts.qp_far_slack [m] (A/B runs; default: the library's 0.7)")
    ap.add_argument("--joint-schedule", type=int, default=0, help="rbp_solver_opts.joint_schedule (A/B runs: 2 = bulk, 3 = bulk with two pivot tiles per pass)")
    ap.add_argument("--plain-order", action="store_true", help="rbp_solver_opts.qp_block_order = 0 (A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-mission latency leg (profiling runs: keeps the kernel list clean)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for the CPU test of the launcher)")
    ap.add_argument("--dry-run", action="store_true", help="launcher test: initialise the ranks, print the JSON skeleton, plan nothing")
    ap.add_argument("--native-pair", action="store_true",
                    help="--config c4 --joint on an even number of ranks: every rank runs lib/rbp_c4_joint_rank (plain C++ on the C ABIs: no Python, no torch "
                         "in the solve's process) and the rank pairs {2k, 2k+1} share the joint solve through the STREAM-ORDERED RCCL exchange of "
                         "lib/librbp_rccl.so; torch.distributed only collects the ranks' times")
    args = ap.parse_args()
    if args.config:
        for k, v in CONFIGS[args.config].items():
            setattr(args, k, v)
    if args.missions_per_gpu is None:
        args.missions_per_gpu = 200 if (args.joint and args.agents >= 16) else 2000

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_under_torchrun(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world_size)
    n_ranks = dist.get_world_size() if dist is not None else 1   # what the collective library actually sees
    if args.gpus != n_ranks and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but {n_ranks} rank(s) were launched; reporting n_gpus={n_ranks}", file=sys.stderr)
    if args.dry_run:
        n_total, secs = aggregate(args.missions_per_gpu * args.agents, 1.0, dist)
        extra = {}
        if args.config == "c4":
            extra = dry_run_c4(args, rank, n_ranks, dist)
        if rank == 0:
            print(json.dumps({"metric": "agent-trajectories/sec (RBP plan: SFC+RSFC+QP)", "value": None, "n_gpus": n_ranks, "dry_run": True,
                              "agents_all_ranks": n_total, "maps_rank0": shard_missions(min(args.missions_per_gpu, 4), rank, n_ranks), **extra}))
        if dist is not None:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the RBP path has no CPU fallback)")
    if args.backend != "nccl":   # (functional test of the launcher / of a rank pair on a one-GPU box: ranks share devices; timings then mean nothing)
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)

    from swarm_simulator_amd import planner
    from swarm_simulator_amd import _abi as A
    from swarm_simulator_amd.types import Param

    pkw = dict(batch_size=args.batch_size, iteration=args.iteration, sequential=not args.joint)
    param = Param.test_sweep(**pkw)
    if args.config == "c4" and args.native_pair and args.joint:
        return bench_c4_native(args, rank, n_ranks, local_rank, dist)
    if args.config == "c4":
        return bench_c4(args, param, pkw, rank, n_ranks, local_rank, dist)
    map_ids = shard_missions(args.missions_per_gpu, rank, n_ranks)
    mission, worlds, plans = build_inputs(map_ids, args.agents, param)
    K, N = len(plans), mission.qn
    Ms = sorted({p.M for p in plans})
    opts = planner.solver_opts(qp_schedule={"auto": 0, "mono": 1, "phase": 2}[args.qp_schedule], qp_groups=args.qp_groups,
                               qp_variant={"auto": 0, "w2": 2, "w4": 4}[args.qp_variant], qp_block_order=0 if args.plain_order else 1)
    if args.qp_far_slack is not None:
        opts.qp_far_slack = args.qp_far_slack
    opts.joint_schedule = args.joint_schedule
    sess = planner.Session(worlds, [mission] * K, param, plans, device=local_rank, opts=opts)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        sess.reset(stream)
        sess.run(A.RBP_STAGE_ALL, stream)

    sess.reserve_workspace(stream)   # (the QP workspace is otherwise reserved by the first planner run: not part of a plan's time)
    # the session's FIRST run has no history for the block order (DevSession::qp_order: longest mission of the previous run first): it
    # is timed on its own and reported beside `value` (value_first_run) -- the reference's sweep plans every map once
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first_run_s = time.perf_counter() - tf0
    for _ in range(max(args.warmup, 1) - 1):
        step()
    torch.cuda.synchronize()
    status = sess.download(stream)
    variant = args.qp_variant if args.qp_schedule != "phase" else "phase"  # (what kernel_source_sha ties a PMC file to)
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
    unpolished = int(sum(p.qp_unpolished for p in plans))

    if rank == 0:
        value = n_total / secs
        # dominant kernel: the batch QP kernel, ONE launch per step (every workgroup runs its mission's whole batch schedule).
        # Two roofs are reported for it:
        #  * HBM (the binding one: arithmetic intensity ~0.3 flop/B): ALGORITHMIC bytes per launch = the row state / row constants
        #    every sweep streams and the knot blocks every factorisation / substitution reads and writes, counted by the kernel
        #    itself (rbp_counters.qp_row_bytes), / planner-stage time (HIP events on the launch stream);
        #  * FP64 MFMA: flops of the dense block factorisations / solves it logs.
        # `traffic` = HBM-side bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
        # command (PMC_FILE); printed only when that file was collected from the kernel sources of this build.
        qp_tflops = ct["qp_flops"] / (planner_ms * 1e-3) / 1e12
        qp_gbs = ct["qp_row_bytes"] / (planner_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
            same = (pmc.get("missions_per_gpu") == K and pmc.get("kernel_source_sha") == kernel_source_sha(variant) and N == 64 and
                    args.batch_size == 4 and args.iteration == 1 and not args.joint)
            if same:
                traffic = pmc["kernels"]["qp_batch_kernel"]["hbm_bytes_per_launch"]
                traffic_src = PMC_FILE
        except Exception:
            pass
        sfc_bytes = 4.0 * ct["sfc_samples"]
        joint_wide = args.joint and N >= 16  # (rbp_solver_opts.joint_wide_min_agents, default)
        out = {
            "metric": "agent-trajectories/sec (RBP plan: SFC+RSFC+QP)", "value": value, "unit": "agent-trajectories/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * secs / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "the reference's own mission JSONs and octomap worlds (data/, committed copies of swarm_planner/missions and worlds); "
                    "initTraj from the repository's own ECBS front-end; no padding",
            "config": {"workload": f"{N}-agent random_forest mission (mission_{N}agents_15.json) on worlds/map1..50.bt, {K} "
                                   f"missions in flight per GPU ({K / 50:g} passes of the 50-map sweep), every map with its own "
                                   f"M = makespan + 2 (M in {Ms}, no padding), sequential={str(not args.joint).lower()} "
                                   f"batch_size={args.batch_size} iteration={args.iteration} (plan_rbp_test.launch keys)",
                       "agents": N, "segments": Ms, "missions_per_gpu": K, "parallelism": f"missions sharded over {n_ranks} GPU(s)",
                       "all_missions_ok": not any(status), "qp_kernel_variant": variant, "baseline_config": args.config or "c3",
                       "qp_schedule": args.qp_schedule,
                       "block_order": ("plain (rbp_solver_opts.qp_block_order = 0)" if args.plain_order else
                                       "longest mission of the session's previous run first (the warm-up steps supply the history; the first "
                                       "run of a session uses plain order; results do not depend on the order)")},
            "value_first_run": K * N / first_run_s,
            "stage_ms": {"corridor": corridor_ms, "planner": planner_ms},
            "roofline": {"bound": "hbm", "kernel": "qp_batch_kernel", "achieved": qp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": qp_gbs / HBM_PEAK_GBS, "hbm_frac": qp_gbs / HBM_PEAK_GBS,
                         # SURVEY.md 8(d)'s figure for the QP -- logged factorisation / solve flops against the FP64 matrix peak --
                         # at the top level next to the HBM one (the driver's summary keeps top-level keys only)
                         "mfma_frac": qp_tflops / FP64_MFMA_PEAK_TFLOPS, "mfma_achieved_tflops": qp_tflops,
                         "mfma_peak_tflops": FP64_MFMA_PEAK_TFLOPS,
                         "bound_note": "frac = hbm_frac = ALGORITHMIC bytes (row state + row constants streamed by the three sweeps of an "
                                       "interior-point iteration, knot blocks written / read by factorisation and substitutions; counted by "
                                       "the kernel, DESIGN.md 3.3) / kernel time / 8 TB/s.  achieved is algorithmic GB/s, NOT measured "
                                       "traffic (that is `traffic`, PMC).  mfma_frac = logged block-factorisation flops / kernel time / "
                                       "78.6 TFLOP/s (SURVEY 8d); the block-tridiagonal structure is exploited, so it stays small",
                         "algorithmic_bytes_per_launch": ct["qp_row_bytes"],
                         # HBM-side bytes per launch of a SEPARATE rocprofv3 --pmc collection (not of this run): profiled, tied to the
                         # kernel sources by hash; `traffic` keeps the contract's key
                         "traffic": traffic, "traffic_profiled": traffic, "traffic_source": traffic_src,
                         "traffic_ratio": (traffic / ct["qp_row_bytes"]) if traffic and ct["qp_row_bytes"] else None,
                         "mfma": {"achieved": qp_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": qp_tflops / FP64_MFMA_PEAK_TFLOPS,
                                  "flops_per_launch": ct["qp_flops"]},
                         "ipm_iterations_per_step": ct["qp_ipm_iters"], "constraint_rows_swept_per_step": ct["qp_constraint_rows"],
                         "batch_qps_per_step": ct["qp_solves"], "batch_qps_polished_per_step": ct["qp_polished"],
                         "batch_qps_unpolished_per_step": unpolished, "kkt_max": ct["kkt_max"]},
            "roofline_sfc": {"bound": "hbm", "kernel": "sfc_kernel", "achieved": sfc_bytes / (corridor_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s (reference-equivalent sample bytes, not physical)",
                             "frac": sfc_bytes / (corridor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "samples_per_step": ct["sfc_samples"]},
        }
        if joint_wide:
            # the grid-wide joint QP (kernels/jqp.hip): the dominant kernels are the rank-64 tile updates of the sweep inversion
            # (jq_update / jq_panel, v_mfma_f64_16x16x4_f64); SURVEY 8(d)'s roof for the QP is FP64 MFMA.  achieved = the MFMA flops the
            # solver logs (tile updates + panels of every factorisation) / planner-stage time (HIP events on the launch stream; the
            # stage also holds sweeps, substitutions and the polish, so this is a lower bound of the kernels' own rate -- the rocprofv3
            # summary under profiles/ has their durations)
            out["roofline"] = {"bound": "mfma", "kernel": "jq_update + jq_panel (joint QP, kernels/jqp.hip)", "achieved": qp_tflops,
                               "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": qp_tflops / FP64_MFMA_PEAK_TFLOPS,
                               "mfma_frac": qp_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                               "flops_per_step": ct["qp_flops"], "ipm_iterations_per_step": ct["qp_ipm_iters"],
                               "qps_per_step": ct["qp_solves"], "qps_polished_per_step": ct["qp_polished"], "kkt_max": ct["kkt_max"]}
            out["config"]["qp_kernel_variant"] = "grid-wide joint (jqp)"
            # the dominant kernel's own rate, from the committed rocprofv3 summary of this very command (a constant tied to the joint
            # solver's sources by hash, like traffic_profiled above; tools/collect_joint_profiles.sh + profiles/README.md say how)
            try:
                jk = json.load(open(os.path.join(ROOT, "profiles", "r06_joint_kernel.json")))
                base = os.path.join(ROOT, "swarm_simulator_amd", "csrc", "kernels")
                h = hashlib.sha256()
                for f in sorted(os.listdir(base)):
                    if f.startswith("jqp"):
                        h.update(f.encode()), h.update(open(os.path.join(base, f), "rb").read())
                if jk["joint_source_sha"] == h.hexdigest()[:16] and jk["missions_per_gpu"] == K and jk["agents"] == N:
                    out["roofline"]["kernel_profiled"] = {"kernel": jk["kernel"], "achieved": jk["tflops"], "unit": "TFLOP/s",
                                                          "frac": jk["frac_of_fp64_mfma_peak"], "source": "profiles/r06_joint_kernel.json"}
            except Exception:
                pass
        # the single-mission latency and the CPU baseline are rank-0, N = 1 legs (the other ranks would only wait for them)
        if world_size == 1 and N == 64 and args.batch_size == 4 and args.iteration == 1 and not args.joint and not args.no_latency:
            try:
                out["latency_ms_single_mission"] = single_mission_latency(mission, param, worlds[0], plans[0])
            except Exception as e:
                out["latency_ms_single_mission"] = {"error": str(e)}
            try:
                out["single_sweep"] = single_sweep_rate(args.agents, param)
            except Exception as e:
                out["single_sweep"] = {"error": str(e)}
            # top-level scalars (summaries keep top-level keys only)
            lat, ss = out["latency_ms_single_mission"], out["single_sweep"]
            try:  # the sweeps' own rate against the streaming roof (6.3 TB/s measured copy rate, 8 TB/s nominal)
                sp = sweep_phase_rate(K, N)
            except Exception:
                sp = None
            out["roofline"]["sweep_phase_gbs"] = sp["gbs"] if sp else None
            out["roofline"]["sweep_phase"] = sp and {
                "gbs": sp["gbs"], "frac_of_hbm_peak": sp["gbs"] / HBM_PEAK_GBS, "sweep_bytes_per_launch": sp["sweep_bytes"],
                "sweep_share_of_kernel_time": sp["sweep_share_of_kernel_time"], "resident_workgroups": sp["resident_workgroups"],
                "note": "algorithmic bytes of the row sweeps (BUILD, AFF, STEP, NBHD, UPDATE) / the time the workgroups spend in those "
                        "phases (in-kernel 100 MHz timers of the profiling build, same workload, separate process), times the workgroups "
                        "resident at a time; the timers add a barrier per phase, so this is a slight under-estimate"}
            try:  # the joint QP (plan/sequential = false) under the same clock: VERDICT r04 item 3
                out.update(joint_leg(args.agents))
            except Exception as e:
                out["roofline_joint"] = {"error": str(e)}
            out["single_mission_two_calls_ms"] = lat.get("two_calls_ms") if isinstance(lat, dict) else None
            out["single_sweep_value"] = ss.get("value") if isinstance(ss, dict) else None
        if world_size == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.agents, pkw)
            except Exception as e:  # the oracle is optional equipment of the bench
                out["cpu_baseline"] = {"value": None, "unit": "agent-trajectories/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    sess.close()
    if dist is not None:
        dist.destroy_process_group()


def dry_run_c4(args, rank, n_ranks, dist):
    """--config c4 --dry-run (no GPU): the exchange of the agent-sharded corridor with the real partition and the real fused all-gather
    (swarm_simulator_amd/sharded.py) on a synthetic 256-agent plan -- every rank fills ONLY its shard with values that name their agent /
    pair row, the rest is poison; after the gather every rank must hold every agent's values.  Returns what rank 0 prints."""
    import numpy as np
    from swarm_simulator_amd import sharded
    from swarm_simulator_amd.types import PlanResult
    N, M = args.agents, 5
    slices = sharded.agent_slices(N, n_ranks)
    plan = PlanResult(np.zeros((N, M + 1, 3), np.float32), np.arange(M + 1, dtype=np.float64))
    plan.sfc_count[:] = -7
    plan.sfc_box[:] = np.nan
    plan.sfc_time[:] = np.nan
    plan.rsfc_normal[:] = np.nan
    b, e = slices[rank]
    o0, o1 = sharded.pair_offset(N, b), sharded.pair_offset(N, e)
    plan.sfc_count[b:e] = 1 + np.arange(b, e) % M
    plan.sfc_box[b:e] = np.arange(b, e, dtype=np.float64)[:, None, None] + 0.25
    plan.sfc_time[b:e] = np.arange(b, e, dtype=np.float64)[:, None] + 0.5
    plan.rsfc_normal[o0:o1] = -np.arange(o0, o1, dtype=np.float32)[:, None, None]   # (row 0 is -0.0: the sign must survive)
    if dist is not None:
        sharded.gather_corridor(dist, plan, N, slices)
    npair = N * (N - 1) // 2
    ok = bool(np.array_equal(plan.sfc_count, 1 + np.arange(N) % M) and np.array_equal(plan.sfc_box[:, 0, 0], np.arange(N) + 0.25)
              and np.array_equal(plan.sfc_time[:, -1], np.arange(N) + 0.5) and np.array_equal(plan.rsfc_normal[:, 0, 0], -np.arange(npair, dtype=np.float32))
              and np.signbit(plan.rsfc_normal[0, 0, 0]))
    oks = [ok]
    if dist is not None:
        import torch
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        oks = [bool(t.item())]
    pairs = sharded.pair_group(dist) is not None if (dist is not None and args.joint) else False
    return {"c4_agent_slices": slices, "c4_gather_ok_on_every_rank": oks[0], "c4_joint_rank_pairs": pairs}


def bench_c4_native(args, rank, n_ranks, local_rank, dist):
    """--config c4 --joint --native-pair: the solve of every rank runs in lib/rbp_c4_joint_rank (csrc/rccl/c4_joint_rank.cpp), a C++ program on
    the C ABIs alone; with an even number of ranks the pairs {2k, 2k+1} share it through rbp_session_shard_joint_stream +
    rbp_rccl_exchange_stream (grouped ncclSend / ncclRecv enqueued on the run's stream).  This process only launches it and collects the times."""
    import subprocess
    import torch
    from swarm_simulator_amd import _abi as A
    exe = os.path.join(A.LIB_DIR, "rbp_c4_joint_rank")
    if not os.path.exists(exe):
        raise SystemExit("lib/rbp_c4_joint_rank is not built (needs RCCL headers: __graft_entry__.build())")
    paired = n_ranks >= 2 and n_ranks % 2 == 0
    idfile = f"/tmp/rbp_pair_{os.environ.get('MASTER_PORT', '0')}_{rank // 2}_{os.getpid() if not paired else 0}.id"
    if paired and rank % 2 == 0 and os.path.exists(idfile):
        os.remove(idfile)
    if dist is not None:
        dist.barrier()
    cmd = [exe, str(rank % 2 if paired else 0), "2" if paired else "1", str(local_rank), idfile, os.path.join(ROOT, "data"), str(args.steps)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
    if out.returncode != 0:
        raise SystemExit(f"rank {rank}: rbp_c4_joint_rank failed ({out.returncode}): {out.stderr[-800:]}")
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    secs = res["ms_per_step"] * 1e-3 * args.steps
    if dist is not None:
        t = torch.tensor([secs], dtype=torch.float64, device=torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "agent-trajectories/sec (RBP plan: SFC+RSFC+QP)", "value": res["agents"] * args.steps / secs, "unit": "agent-trajectories/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": 1, "ms_per_step": 1e3 * secs / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64",
            "data": "256-agent mission derived from the reference's 64-agent pattern (tools/make_mission_256.py; no such file upstream), worlds/map1.bt",
            "config": {"workload": f"ONE {res['agents']}-agent mission (M={res['segments']}) as ONE joint QP (plan/sequential=false), PLANNER stage timed; "
                                   + ("its knot elimination shared by rank pairs {2k, 2k+1}: " if paired else "whole solve per rank: ") + res["exchange"],
                       "agents": res["agents"], "segments": res["segments"], "parallelism": "joint factorisation shared by rank pairs" if paired else "replicas",
                       "process": "lib/rbp_c4_joint_rank (C++ on include/rbp.h, rbp_host.h, rbp_rccl.h; no Python / torch in the solve)",
                       "qp_iterations": res["qp_iterations"], "qp_unpolished": res["qp_unpolished"], "kkt_max": res["kkt_max"], "baseline_config": "c4",
                       "all_missions_ok": True}}))
    if dist is not None:
        dist.destroy_process_group()


def bench_c4(args, param, pkw, rank, n_ranks, local_rank, dist):
    """BASELINE.json config 4: ONE 256-agent mission, Corridor::update sharded by agent over the ranks with one fused all-gather
    (swarm_simulator_amd/sharded.py), then the planner sweep.  value = 256 agents * steps / wall time (max over ranks)."""
    import torch
    from swarm_simulator_amd import host, sharded
    from swarm_simulator_amd.types import Param
    # the 256-agent mission (tools/make_mission_256.py) flies in the world x in [-5, 15]: forest around x = 0, open ground around x = 10
    param = Param.test_sweep(world_x_min=-5, world_y_min=-5, world_x_max=15, world_y_max=5, **pkw)
    m = host.load_mission("mission_256agents_c4.json")
    w = host.load_world("map1.bt", param)
    init = host.ecbs_plan(w, m, param)
    dev = torch.device("cuda", local_rank)

    xstats = {}

    def step():
        pr = init.clone_inputs()
        t0 = time.perf_counter()
        ok, err = sharded.plan_sharded_device(w, m, param, pr, dist, dev, stats=xstats)
        torch.cuda.synchronize()
        return ok, err, time.perf_counter() - t0, pr

    for _ in range(max(args.warmup, 1)):
        ok, err, _, _ = step()
        if not ok:
            raise SystemExit(f"rank {rank}: C4 mission failed: {err}")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ok, err, _, pr = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    secs = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([secs], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        secs = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "agent-trajectories/sec (RBP plan: SFC+RSFC+QP)", "value": m.qn * args.steps / secs, "unit": "agent-trajectories/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * secs / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "256-agent mission derived from the reference's 64-agent pattern (tools/make_mission_256.py; no such file upstream), worlds/map1.bt",
            "config": {"workload": f"ONE {m.qn}-agent mission (M={pr.M}), Corridor::update sharded by agent over {n_ranks} rank(s) + one fused "
                                   f"all-gather, " + ("RBPPlanner JOINT QP (plan/sequential=false: knot blocks of order 9 N, grid-wide solver "
                                                      "kernels/jqp.hip), " + (f"its twisted knot elimination shared by PAIRS of ranks (rank 2k the lower chain, "
                                                      f"2k+1 the upper one; {xstats.get('exchanges', 0)} exchanges, {xstats.get('exchange_bytes', 0) / 1e6:.0f} MB each "
                                                      "way per solve), sweeps / polish replicated" if xstats.get("exchanges") else "replicated on every rank")
                                                      if args.joint else
                                                      f"RBPPlanner sweep sequential batch_size={args.batch_size}") + "; host buffers in and out",
                       "agents": m.qn, "segments": pr.M, "parallelism": f"agents sharded over {n_ranks} GPU(s) (corridor), " +
                       ("joint factorisation shared by rank pairs" if xstats.get("exchanges") else "planner per rank"),
                       "backend": "single process" if dist is None else dist.get_backend(),
                       "joint_exchanges_per_solve": xstats.get("exchanges", 0), "joint_exchange_bytes_per_solve": xstats.get("exchange_bytes", 0),
                       "qp_iterations": pr.qp_iterations,
                       "baseline_config": "c4", "qp_unpolished": pr.qp_unpolished, "kkt_max": pr.kkt_max, "all_missions_ok": bool(ok)}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
