"""The PHASE-SPLIT schedule of the batch QPs (kernels/qp_phase.inc, rbp_solver_opts.qp_schedule = 2; since round 6 in the developer build
lib/librbp_hip_dev.so only -- the release library refuses it -- so its cases, tests/phase_split_cases.py, run in a process of their own): chip-wide row sweeps as kernels of
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


def test_release_library_refuses_the_phase_split_schedule():
    """kernels/qp_phase.inc lost every comparison and is compiled by `make dev` only: the release library says so, before any launch"""
    p = Param.test_sweep()
    m = host.load_mission("mission_8agents_15.json")
    w = host.load_world("map5.bt", p)
    g = host.ecbs_plan(w, m, p)
    sess = planner.Session([w], [m], p, [g], opts=planner.solver_opts(qp_schedule=2))
    with pytest.raises(RuntimeError, match="phase-split schedule is not part of the release library"):
        sess.run(A.RBP_STAGE_ALL)
    sess.close()


def test_phase_split_cases_in_the_developer_build():
    """tests/phase_split_cases.py against lib/librbp_hip_dev.so: the two schedules agree to 2e-7 m, solve and polish the same QPs, two groups of
    missions give the same bits as one; a round budget that does not finish a mission is RBP_ERR_QP_FAILED"""
    import os
    import subprocess
    import sys
    dev = os.path.join(A.LIB_DIR, "librbp_hip_dev.so")
    assert os.path.exists(dev), "lib/librbp_hip_dev.so is missing: __graft_entry__.build() (make dev) builds it"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/phase_split_cases.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=root,
                       env={**os.environ, "RBP_HIP_LIB": dev}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


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
