"""The phase-split cases of tests/test_gpu_phase.py: run by it in a process of their own with RBP_HIP_LIB = lib/librbp_hip_dev.so (`make dev`, -DRBP_PHASE_SPLIT:
the release library does not carry kernels/qp_phase.inc).  `python -m pytest tests/phase_split_cases.py -m gpu` with that variable set runs them by hand."""
import numpy as np
import pytest

from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def _run(worlds, missions, p, inits, times=1, **opts):
    plans = [g.clone_inputs() for g in inits]
    sess = planner.Session(worlds, missions, p, plans, opts=planner.solver_opts(**opts))
    outs = []
    for _ in range(times):
        sess.reset()
        sess.run(A.RBP_STAGE_ALL)
        assert sess.download() == [0] * len(plans)
        outs.append([g.ctrl.copy() for g in plans])
    sess.close()
    return plans, outs


@pytest.mark.parametrize("agents,batch,iteration,maps", [(64, 4, 1, [1, 2, 46, 4, 5, 6]), (16, 8, 3, [3, 9]), (8, 3, 1, [5])])
def test_phase_split_equals_one_workgroup_per_mission(agents, batch, iteration, maps):
    """ragged session (M = 34..37), a last batch shorter than the others (8 agents in batches of 3), several Gauss-Seidel passes with the
    polish-first shortcut, the tiled path (batches of 8): the two schedules agree to 2e-7 m, solve and polish the same QPs; two groups of
    missions on two streams give the same bits as one"""
    p = Param.test_sweep(batch_size=batch, iteration=iteration)
    m = host.load_mission(f"mission_{agents}agents_15.json")
    worlds = [host.load_world(f"map{i}.bt", p) for i in maps]
    inits = [host.ecbs_plan(w, m, p) for w in worlds]
    mono, _ = _run(worlds, [m] * len(maps), p, inits, qp_schedule=1)
    phase, (r1, r2) = _run(worlds, [m] * len(maps), p, inits, times=2, qp_schedule=2, qp_groups=1)
    _, (g2,) = _run(worlds, [m] * len(maps), p, inits, qp_schedule=2, qp_groups=2)
    for a, b in zip(mono, phase):
        assert b.qp_solves == a.qp_solves and b.qp_unpolished == 0 and a.qp_unpolished == 0
        assert np.abs(a.ctrl - b.ctrl).max() < 2e-7
        assert abs(a.total_cost - b.total_cost) <= 1e-8 * max(1.0, abs(a.total_cost))
        obj, veq, vbox, vrs = O.evaluate_ctrl(m, b)
        assert veq < 5e-8 and vbox < 1e-8 and vrs < 1e-8
    for x, y, z in zip(r1, r2, g2):
        assert np.array_equal(x.view(np.uint64), y.view(np.uint64)) and np.array_equal(x.view(np.uint64), z.view(np.uint64))


def test_phase_split_round_budget_fails_loudly():
    """a mission the round budget does not finish is an error (RBP_ERR_QP_FAILED), never a half-solved plan"""
    p = Param.test_sweep()
    m = host.load_mission("mission_8agents_15.json")
    w = host.load_world("map5.bt", p)
    g = host.ecbs_plan(w, m, p)
    sess = planner.Session([w], [m], p, [g], opts=planner.solver_opts(qp_schedule=2, qp_rounds=5))
    sess.run(A.RBP_STAGE_ALL)
    assert sess.download() == [A.RBP_ERR_QP_FAILED]
    sess.close()
