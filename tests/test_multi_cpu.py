"""N > 1 path of bench.py on CPU: world_size-2 gloo processes shard the map sweep and aggregate the metric."""
import os
import subprocess
import sys
import textwrap

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n"] == 2 * 5 * 64 and res["t"] == 2.0 and res["maps"] == [1, 2, 3, 4, 5]
