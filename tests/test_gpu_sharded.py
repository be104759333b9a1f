"""GPU: agent-sharded Corridor::update (rbp_corridor_update_range) and the 256-agent config C4 on one GPU."""
import os

import numpy as np
import pytest

from swarm_simulator_amd import host, planner
from swarm_simulator_amd.sharded import agent_slices, pair_offset, plan_sharded
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

C4_PARAM = dict(world_x_min=-5, world_y_min=-5, world_x_max=15, world_y_max=5)


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_range_shards_tile_the_full_corridor(n_shards):
    """every shard call writes exactly its agents / pair rows, bit-identical to the unsharded call"""
    p = Param.test_sweep()
    m = host.load_mission("mission_16agents_15.json")
    w = host.load_world("map12.bt", p)
    init = host.ecbs_plan(w, m, p)
    full = init.clone_inputs()
    cor = planner.Corridor(w, m, p)
    assert cor.update(False, full), cor.last_error
    for b, e in agent_slices(m.qn, n_shards):
        part = init.clone_inputs()
        assert cor.update_range(part, b, e), cor.last_error
        assert np.array_equal(part.sfc_count[b:e], full.sfc_count[b:e])
        assert np.array_equal(part.sfc_box[b:e], full.sfc_box[b:e]) and np.array_equal(part.sfc_time[b:e], full.sfc_time[b:e])
        o0, o1 = pair_offset(m.qn, b), pair_offset(m.qn, e)
        assert np.array_equal(part.rsfc_normal[o0:o1].view(np.uint32), full.rsfc_normal[o0:o1].view(np.uint32))
        assert np.array_equal(part.rsfc_time, full.rsfc_time)
    assert not cor.update_range(init.clone_inputs(), 3, 99)  # outside [0, N]: refused


def test_c4_256_agents_single_rank():
    """config C4 (256 agents, tools/make_mission_256.py) through plan_sharded with one rank: corridor bit-exact against the
    CPU checker, the QP sweep (64 batches, 252 frozen neighbours each) feasible, all equalities met, every batch QP polished,
    and three of the batch QPs certified optimal by the independent numpy restatement."""
    p = Param.test_sweep(**C4_PARAM)
    m = host.load_mission("mission_256agents_c4.json")
    assert m.qn == 256
    w = host.load_world("map1.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref, gpu = init.clone_inputs(), init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    ok, err = plan_sharded(w, m, p, gpu, device="cuda")   # the mission stays resident in a session between the two stages
    assert ok, err
    assert np.array_equal(ref.sfc_count, gpu.sfc_count) and np.array_equal(ref.sfc_box, gpu.sfc_box)
    assert np.array_equal(ref.rsfc_normal.view(np.uint32), gpu.rsfc_normal.view(np.uint32))
    obj, veq, vbox, vrs = O.evaluate_ctrl(m, gpu)
    # equality rows 5e-8 (rows of Aeq_base carry factors up to 20 and a control point snapped onto an active face moves by <= 5e-9 m:
    # DESIGN.md 4, the EQ_TOL of the other parity tests), inequality rows 1e-8
    assert veq < 5e-8 and vbox < 1e-8 and vrs < 1e-8, (veq, vbox, vrs)
    assert abs(obj - gpu.total_cost) < 1e-6 * max(1.0, obj)
    assert gpu.qp_unpolished == 0 and gpu.qp_solves == 64
    # optimality, not only feasibility: the independent numpy certificate (tests/golden/make_kkt_reference.py) on the first, a middle
    # and the last batch QP of the sweep (4 agents against 252 frozen neighbours each)
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_kkt_reference as K
    for rep in K.certify_plan(init.T, init.init_traj, m.start, m.goal, m.radius, gpu.sfc_box, ref.sfc_time, gpu.sfc_count, gpu.rsfc_normal,
                              ref.rsfc_time, gpu.ctrl, p.sequential, p.batch_size, p.batch_iter, only_batches=[0, 31, 63]):
        tag = f"C4 batch {rep['batch']}: " + ", ".join(f"{k}={v:.3g}" for k, v in rep.items() if isinstance(v, float))
        assert rep["x_as_viol_ineq"] < 1e-7 and rep["x_as_viol_eq"] < 1e-8 and rep["stationarity"] < 1e-7, tag
        assert rep["forward_error"] < 2e-6, tag


@pytest.mark.parametrize("n_shards", [2, 5])
def test_device_side_exchange_of_shards(n_shards):
    """the exchange of the agent-sharded corridor on DEVICE buffers (sharded.pack_shard_device / unpack_shard_device over
    planner.Session.device_arrays): every "rank" is a session of its own on this GPU that runs the CORRIDOR stage on its slice; rank 0's
    session receives the other shards (what all_gather_into_tensor delivers), runs the PLANNER stage on the completed corridor and must
    give, bit for bit, what the unsharded calls give.  No host copy of the corridor exists between the two stages."""
    import torch
    from swarm_simulator_amd import _abi as A
    from swarm_simulator_amd import sharded
    p = Param.test_sweep()
    m = host.load_mission("mission_16agents_15.json")
    w = host.load_world("map12.bt", p)
    init = host.ecbs_plan(w, m, p)
    full = init.clone_inputs()
    assert planner.Corridor(w, m, p).update(False, full)
    assert planner.RBPPlanner(m, p).update(False, full)
    slices = agent_slices(m.qn, n_shards)
    plans = [init.clone_inputs() for _ in range(n_shards)]
    sessions = [planner.Session([w], [m], p, [pl]) for pl in plans]
    arrs = [s.device_arrays(0) for s in sessions]
    _, _, offs, lens = sharded._shard_layout(m.qn, plans[0].M, plans[0].sfc_box.shape[1], slices)
    packed = []
    for r, s in enumerate(sessions):
        s.set_agent_range(*slices[r])
        s.run(A.RBP_STAGE_CORRIDOR)
        buf = sharded.pack_shard_device(arrs[r], slices[r], offs[r])
        assert buf.is_cuda and buf.numel() == lens[r]
        packed.append(buf)
    for r in range(1, n_shards):
        sharded.unpack_shard_device(arrs[0], packed[r], slices[r], offs[r])
    torch.cuda.synchronize()
    sessions[0].set_agent_range(0, m.qn)
    sessions[0].run(A.RBP_STAGE_PLANNER)
    assert sessions[0].download() == [0]
    g = plans[0]
    assert np.array_equal(g.sfc_box, full.sfc_box) and np.array_equal(g.sfc_count, full.sfc_count)
    assert np.array_equal(g.rsfc_normal.view(np.uint32), full.rsfc_normal.view(np.uint32))
    assert np.array_equal(g.ctrl.view(np.uint64), full.ctrl.view(np.uint64)) and g.total_cost == full.total_cost
    for s in sessions:
        s.close()


def test_two_rank_plan_sharded_device(tmp_path):
    """plan_sharded_device with a real process group of two ranks (both on this one GPU, so the group is gloo: RCCL refuses two ranks on
    one device; on a multi-GPU node the same code takes all_gather_into_tensor over RCCL): flag all_reduce, padded shard exchange between
    the two sessions' HBM arrays, PLANNER stage on the completed corridor.  Every rank must return the unsharded plan bit for bit."""
    import json, re, socket, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "s.py"
    script.write_text(textwrap.dedent(f"""
        import sys, json
        sys.path.insert(0, {root!r})
        import numpy as np, torch
        import torch.distributed as dist
        from swarm_simulator_amd import host, planner
        from swarm_simulator_amd.types import Param
        from swarm_simulator_amd.sharded import plan_sharded_device
        dist.init_process_group("gloo")
        p = Param.test_sweep()
        m = host.load_mission("mission_16agents_15.json")
        w = host.load_world("map12.bt", p)
        init = host.ecbs_plan(w, m, p)
        full = init.clone_inputs()
        assert planner.Corridor(w, m, p).update(False, full) and planner.RBPPlanner(m, p).update(False, full)
        mine = init.clone_inputs()
        ok, err = plan_sharded_device(w, m, p, mine, dist, "cuda:0")
        same = bool(ok and np.array_equal(mine.sfc_box, full.sfc_box) and np.array_equal(mine.sfc_count, full.sfc_count)
                    and np.array_equal(mine.rsfc_normal.view(np.uint32), full.rsfc_normal.view(np.uint32))
                    and np.array_equal(mine.ctrl.view(np.uint64), full.ctrl.view(np.uint64)) and mine.total_cost == full.total_cost)
        print(json.dumps({{"rank": dist.get_rank(), "same": same, "err": err}}))
        dist.destroy_process_group()
    """))
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = str(sk.getsockname()[1]); sk.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1"),
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = [json.loads(t) for t in re.findall(r"\{[^{}]*\}", out.stdout)]
    assert len(res) == 2 and all(r["same"] for r in res), res
