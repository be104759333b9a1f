"""The PHASE-SPLIT schedule of the batch QPs (kernels/qp_phase.inc, rbp_solver_opts.qp_schedule = 2): chip-wide row sweeps as kernels of
their own, one workgroup per mission for chains and polish, a fixed budget of rounds enqueued without synchronisation.  Same device
functions as qp_batch_kernel (one workgroup per mission for everything, the default); only the block reductions of a sweep are summed in a
different order (and the monolith's interior-point phase works on the reduced row set of QP_FAR_SLACK in the first pass, the phase split on
every row): both schedules land on the same KKT-certified optimum.  Needs an MI355X."""
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


def test_reduced_row_set_is_the_same_optimum_for_any_radius():
    """rbp_solver_opts.qp_far_slack (kernels/qp.hip QP_FAR_SLACK): the interior-point phase of a first-pass batch QP leaves out the frozen-
    neighbour rows that are far from active at the starting point; the polish verifies EVERY row of the full QP, and a batch QP that does not
    end polished on the reduced set is solved again with every row.  Off, the default 0.7 m, and 0.15 m -- where so many needed rows are
    left out that second attempts actually happen (scalars slot 11 counts them) -- must give the same KKT-certified optimum, feasible for
    every row of the reference's constraint sets."""
    _reduced_row_set_case(0)


def test_reduced_row_set_second_attempts_in_the_256_thread_build():
    """the same with the throughput build of the QP kernel pinned (rbp_solver_opts.qp_variant = 4: what a session of more missions than CUs
    runs; small sessions default to the 512-thread build, so nothing else in the suite sends second attempts through it).  Round 6: its
    rest-of-schedule path read a vector register that a failed factorisation had left dirty -- a memory fault, seen only once the kernel's
    register allocation had shifted (the just-in-time block assembly) -- and now runs every batch QP of that path in a frame of its own."""
    _reduced_row_set_case(4)


def _reduced_row_set_case(variant):
    p = Param.test_sweep()
    m = host.load_mission("mission_64agents_15.json")
    maps = [1, 13, 27, 46] if variant == 0 else [1, 21]
    worlds = [host.load_world(f"map{i}.bt", p) for i in maps]
    inits = [host.ecbs_plan(w, m, p) for w in worlds]
    res, again = {}, {}
    for R in (0.0, 0.7, 0.15):
        plans = [g.clone_inputs() for g in inits]
        sess = planner.Session(worlds, [m] * len(maps), p, plans, opts=planner.solver_opts(qp_schedule=1, qp_variant=variant, qp_far_slack=R))
        sess.run(A.RBP_STAGE_ALL)
        assert sess.download() == [0] * len(maps)
        again[R] = sess.scalars(12)[:, 11].sum()
        sess.close()
        res[R] = plans
        for g in plans:
            assert g.qp_solves == 16 and g.qp_unpolished == 0 and g.kkt_max < 1e-8
            obj, veq, vbox, vrs = O.evaluate_ctrl(m, g)
            assert veq < 5e-8 and vbox < 1e-8 and vrs < 1e-8
    assert again[0.0] == 0
    assert again[0.15] > 0, "the 0.15 m radius is meant to exercise the second attempt"
    for R in (0.7, 0.15):
        for a, b in zip(res[0.0], res[R]):
            assert np.abs(a.ctrl - b.ctrl).max() < 5e-7
            assert abs(a.total_cost - b.total_cost) <= 1e-8 * max(1.0, abs(a.total_cost))
    assert sum(g.qp_iterations for g in res[0.7]) < sum(g.qp_iterations for g in res[0.0])
