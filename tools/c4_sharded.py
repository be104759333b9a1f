#!/usr/bin/env python
"""BASELINE.json config C4: ONE 256-agent mission, agents sharded over the ranks (one process per GPU, RCCL all-gather of
the corridor shards), then the QP sweep on every rank.  Launch like bench.py:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/c4_sharded.py [--map 1]
(G = 1 works without a launcher).  Prints one JSON line on rank 0 with the stage times."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from swarm_simulator_amd import host
from swarm_simulator_amd.sharded import ShardedCorridor, agent_slices
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param

ap = argparse.ArgumentParser()
ap.add_argument("--map", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
rank, ws, lr = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(lr)
dist = None
if ws > 1 or "RANK" in os.environ:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=torch.device("cuda", lr))
p = Param.test_sweep(world_x_min=-5, world_y_min=-5, world_x_max=15, world_y_max=5)
m = host.load_mission("mission_256agents_c4.json")
w = host.load_world(f"map{args.map}.bt", p)
init = host.ecbs_plan(w, m, p)
dev = torch.device("cuda", lr)
tc, tq = [], []
for rep in range(args.reps + 1):
    plan = init.clone_inputs()
    cor = ShardedCorridor(w, m, p, dist, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert cor.update(False, plan), cor.last_error
    t1 = time.perf_counter()
    pl = planner.RBPPlanner(m, p)
    assert pl.update(False, plan), pl.last_error
    t2 = time.perf_counter()
    if rep:
        tc.append(t1 - t0), tq.append(t2 - t1)
if rank == 0:
    print(json.dumps({"config": "C4: 256 agents, agents sharded for Corridor::update + all-gather, QP sweep replicated",
                      "n_gpus": ws, "agents": m.qn, "segments": plan.M, "slices": agent_slices(m.qn, ws),
                      "corridor_shard_plus_allgather_s": float(np.median(tc)), "planner_s": float(np.median(tq)),
                      "total_cost": plan.total_cost, "note": "one-shot C ABI calls: include upload/download of the mission"}))
if dist is not None:
    dist.destroy_process_group()
