"""include/rbp_rccl.h: the exchange hook of rbp_session_shard_joint over RCCL (lib/librbp_rccl.so).  A one-GPU box cannot host two RCCL ranks,
so what runs here is the plumbing with a pair of ONE rank (send to self): unique id, communicator, grouped ncclSend / ncclRecv of a vector, a
64-agent inverse and a 256-agent inverse on the pair's stream, bitwise comparison on the device (csrc/rccl/selftest.hip, a binary of its
own: the python process holds torch's bundled HIP runtime, RCCL brings /opt/rocm's)."""
import os
import subprocess

import pytest

from swarm_simulator_amd import _abi as A

pytestmark = pytest.mark.gpu


def test_rccl_exchange_hook_self_pair():
    exe = os.path.join(A.LIB_DIR, "rbp_rccl_selftest")
    if not os.path.exists(exe):
        pytest.skip("lib/rbp_rccl_selftest not built (no RCCL headers where build() ran)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and "rbp_rccl selftest ok" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-1500:])
    assert out.stdout.count("0 words differ") == 4 and "stream-ordered exchange" in out.stdout


def test_native_rank_runner_gives_the_python_paths_answer():
    """lib/rbp_c4_joint_rank (csrc/rccl/c4_joint_rank.cpp): one rank of a joint solve as a plain C++ program on the three C ABIs -- mission JSON,
    octomap, distance grid and ECBS from librbp_host.so, the session from librbp_hip.so, the stream-ordered RCCL exchange from librbp_rccl.so
    when it has a peer (`bench.py --config c4 --joint --native-pair`; two ranks need two GPUs).  Here: the whole solve on one rank, on the
    64-agent mission, against the Python binding's answer for the same inputs -- same library, same bits."""
    import json
    import numpy as np
    from swarm_simulator_amd import host, planner
    from swarm_simulator_amd.types import Param
    exe = os.path.join(A.LIB_DIR, "rbp_c4_joint_rank")
    if not os.path.exists(exe):
        pytest.skip("lib/rbp_c4_joint_rank not built (no RCCL headers where build() ran)")
    out = subprocess.run([exe, "0", "1", "0", "/tmp/unused.id", os.path.join(A.REPO_ROOT, "data"), "1", "mission_64agents_15.json", "map1.bt"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    p = Param.test_sweep(sequential=False)
    m = host.load_mission("mission_64agents_15.json")
    w = host.load_world("map1.bt", p)
    g = host.ecbs_plan(w, m, p)
    assert planner.Corridor(w, m, p).update(False, g) and planner.RBPPlanner(m, p).update(False, g)
    assert res["agents"] == 64 and res["segments"] == g.M and res["qp_unpolished"] == 0 == g.qp_unpolished
    assert res["qp_iterations"] == g.qp_iterations
    assert abs(res["total_cost"] - g.total_cost) <= 1e-9 * abs(g.total_cost)
    assert abs(res["ctrl_sumsq"] - float((g.ctrl ** 2).sum())) <= 1e-9 * res["ctrl_sumsq"]
