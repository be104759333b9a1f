"""N > 1 path of bench.py on CPU: world_size-2 gloo processes shard the map sweep and aggregate the metric."""
import os
import subprocess
import sys
import textwrap

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


def test_shards_are_disjoint_and_cover_the_sweep():
    for ws in (1, 2, 4, 8):
        per = 50 // ws
        maps = [bench.shard_missions(per, r, ws) for r in range(ws)]
        flat = [m for s in maps for m in s]
        assert len(flat) == len(set(flat)) == per * ws and min(flat) == 1 and max(flat) <= 50
    assert bench.shard_missions(50, 1, 2)[0] == 1  # weak scaling beyond 50 maps wraps around the sweep


def test_two_rank_gloo_aggregate(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import torch.distributed as dist
        import bench
        dist.init_process_group("gloo")
        r = dist.get_rank()
        maps = bench.shard_missions(5, r, dist.get_world_size())
        n, t = bench.aggregate(len(maps) * 64, 1.0 + r, dist)   # rank 1 is slower
        if r == 0:
            print(json.dumps({{"n": n, "t": t, "maps": maps}}))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n"] == 2 * 5 * 64 and res["t"] == 2.0 and res["maps"] == [1, 2, 3, 4, 5]


def test_agent_slices_partition():
    from swarm_simulator_amd.sharded import agent_slices, pair_offset
    for n in (1, 4, 5, 16, 64, 256):
        for ws in (1, 2, 3, 8):
            sl = agent_slices(n, ws)
            assert sl[0][0] == 0 and sl[-1][1] == n and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert max(e - b for b, e in sl) - min(e - b for b, e in sl) <= 1
            # pair rows of the slices tile the upper triangle
            assert sum(pair_offset(n, e) - pair_offset(n, b) for b, e in sl) == n * (n - 1) // 2


def test_two_rank_gloo_sharded_corridor(tmp_path):
    """agent-sharded Corridor::update (BASELINE config C4 path) with 2 gloo ranks: each rank fills ONLY its shard (the CPU
    checker stands in for the HIP shard kernel, everything else is poisoned), the all-gather must reproduce the unsharded
    corridor bit for bit, including the sign of zero normals."""
    script = tmp_path / "s.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch.distributed as dist
        from swarm_simulator_amd import host
        from swarm_simulator_amd.types import Param
        from swarm_simulator_amd.sharded import ShardedCorridor, pair_offset
        from tests import oracle_lib as O
        dist.init_process_group("gloo")
        p = Param.test_sweep()
        m = host.load_mission("mission_16agents_15.json")
        w = host.load_world("map7.bt", p)
        init = host.ecbs_plan(w, m, p)
        full = init.clone_inputs()
        assert O.corridor_update(w, m, p, full)[0] == 0
        full.rsfc_normal[0, 0, 2] = -0.0            # a signed zero must survive the exchange
        def shard_kernel(plan, b, e):               # writes agents [b, e) and their pair rows only
            plan.sfc_count[:] = -7; plan.sfc_box[:] = np.nan; plan.sfc_time[:] = np.nan; plan.rsfc_normal[:] = np.nan
            plan.sfc_count[b:e] = full.sfc_count[b:e]; plan.sfc_box[b:e] = full.sfc_box[b:e]; plan.sfc_time[b:e] = full.sfc_time[b:e]
            o0, o1 = pair_offset(m.qn, b), pair_offset(m.qn, e)
            plan.rsfc_normal[o0:o1] = full.rsfc_normal[o0:o1]
            plan.rsfc_time[:] = full.rsfc_time
            return True
        mine = init.clone_inputs()
        ok = ShardedCorridor(w, m, p, dist, "cpu", compute=shard_kernel).update(False, mine)
        same = (ok and np.array_equal(mine.sfc_count, full.sfc_count) and np.array_equal(mine.sfc_box, full.sfc_box)
                and np.array_equal(mine.sfc_time, full.sfc_time)
                and np.array_equal(mine.rsfc_normal.view(np.uint32), full.rsfc_normal.view(np.uint32)))
        # a failing shard makes Corridor::update false on every rank
        bad = ShardedCorridor(w, m, p, dist, "cpu", compute=lambda pl, b, e: dist.get_rank() != 1).update(False, init.clone_inputs())
        print(json.dumps({{"rank": dist.get_rank(), "same": bool(same), "bad": bool(bad)}}))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    # (both ranks write to the same pipe: two lines can arrive glued together, so the objects are cut out by their braces)
    import re
    res = [json.loads(t) for t in re.findall(r"\{[^{}]*\}", out.stdout)]
    assert len(res) == 2 and all(r["same"] and not r["bad"] for r in res), res


def test_bench_gpus_flag_spawns_the_ranks():
    """`python bench.py --gpus 2` outside a torchrun environment re-executes itself under torch.distributed.run with two ranks
    (here: gloo, --dry-run = initialise, aggregate, plan nothing); n_gpus is what the process group actually saw"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run",
                          "--missions-per-gpu", "3"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["agents_all_ranks"] == 2 * 3 * 64 and res["maps_rank0"] == [1, 2, 3]
    # a single process asked for 2 GPUs but launched by hand with WORLD_SIZE=1 reports what it has
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                         capture_output=True, text=True, env=env1, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1


def test_eight_rank_gloo_dry_run_of_the_scaling_bench():
    """what the driver's SCALE run launches at N = 8, on CPU (gloo, --dry-run): `python -m torch.distributed.run --nproc-per-node 8 bench.py
    --gpus 8` -- the missions of the sweep shard over eight ranks (disjoint slices, weak scaling), n_gpus is the real world size, the
    aggregate counts every rank's agents; and `--config c4`: 256 agents in eight contiguous slices, ONE fused all-gather gives every rank
    the whole corridor (bytes only: the sign of a zero normal survives); with --joint the ranks form the pairs {2k, 2k+1}"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1"]

    def run(extra):
        out = subprocess.run(base + ["--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--dry-run"] + extra,
                             capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    res = run(["--missions-per-gpu", "7"])
    assert res["n_gpus"] == 8 and res["agents_all_ranks"] == 8 * 7 * 64 and res["maps_rank0"] == [1, 2, 3, 4]
    maps = [bench.shard_missions(7, r, 8) for r in range(8)]
    assert sorted(m for s in maps for m in s) == [1 + i % 50 for i in range(56)] or len({m for s in maps for m in s}) == 50
    res = run(["--config", "c4", "--joint"])
    assert res["n_gpus"] == 8 and res["c4_gather_ok_on_every_rank"] is True and res["c4_joint_rank_pairs"] is True
    sl = res["c4_agent_slices"]
    assert len(sl) == 8 and sl[0][0] == 0 and sl[-1][1] == 256 and all(e - b == 32 for b, e in sl)


def test_four_rank_gloo_pair_groups(tmp_path):
    """the rank pairs that share a joint factorisation (sharded.pair_group; rbp_session_shard_joint): with four ranks {0, 1} and {2, 3},
    every rank taking part in the creation of every pair's group; an odd world has no pairs (the solve is then replicated)."""
    script = tmp_path / "p.py"
    script.write_text(textwrap.dedent(f"""
        import sys, json
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from swarm_simulator_amd.sharded import pair_group
        dist.init_process_group("gloo")
        r = dist.get_rank()
        g = pair_group(dist)
        assert pair_group(dist) is g                      # created once
        t = torch.tensor([float(r)])
        outs = [torch.zeros(1) for _ in range(2)]
        dist.all_gather(outs, t, group=g)                 # the exchange pattern of planner.Session.shard_joint under gloo
        print(json.dumps({{"rank": r, "group_rank": dist.get_rank(g), "size": dist.get_world_size(g), "members": [int(o.item()) for o in outs]}}))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                          "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json, re
    res = sorted((json.loads(t) for t in re.findall(r"\{[^{}]*\}", out.stdout)), key=lambda d: d["rank"])
    assert [d["members"] for d in res] == [[0, 1], [0, 1], [2, 3], [2, 3]]
    assert [d["group_rank"] for d in res] == [0, 1, 0, 1] and all(d["size"] == 2 for d in res)


def test_pair_group_of_odd_and_single_worlds():
    from swarm_simulator_amd.sharded import pair_group

    class _D:
        def __init__(self, ws, r):
            self.ws, self.r = ws, r
        def get_world_size(self):
            return self.ws
        def get_rank(self):
            return self.r

    assert pair_group(_D(1, 0)) is None and pair_group(_D(3, 2)) is None
