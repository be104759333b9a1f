"""Round-3 parity additions (VERDICT r02 "next round" item 5 and the ADVICE items).  Needs an MI355X.

* C5 (batch 8 x 50 passes) on the 64-agent mission: every batch QP polished (or KKT residual < 1e-7) and three batch QPs of the
  LAST pass certified optimal by the independent numpy restatement (tests/golden/make_kkt_reference.py), which for plan/iteration
  > 1 needs the control points the pass started from (a second, 49-pass run: the kernel is bit-reproducible);
* a mission with MIXED radii (the reference's quad_size is per agent, mission.hpp:64): corridor bit-exact (the SFC kernel reads
  the float grid for agents whose radius differs from agent 0's, whose bit mask the workgroup shares), QP against the oracle;
* corridor-only use of a session whose joint batch the QP kernel cannot take; stale time_scale after a corridor-only run.
"""
import os
import sys

import numpy as np
import pytest

from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param, PlanResult
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

CTRL_TOL = 2e-6
FEAS_TOL = 1e-8
EQ_TOL = 5e-8


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _kkt():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_kkt_reference as K
    return K


def test_c5_64_agents_last_pass_certified():
    m = host.load_mission("mission_64agents_15.json")
    out = {}
    for it in (49, 50):
        p = Param.test_sweep(batch_size=8, iteration=it)
        w = host.load_world("map1.bt", p)
        g = host.ecbs_plan(w, m, p)
        T0 = g.T.copy()
        assert planner.Corridor(w, m, p).update(False, g)
        corr = g.clone()          # corridor times before timeScale
        pl = planner.RBPPlanner(m, p)
        assert pl.update(False, g), pl.last_error
        assert g.qp_solves == 8 * it
        assert g.qp_unpolished == 0 or g.kkt_max < 1e-7, (g.qp_unpolished, g.kkt_max)
        out[it] = (g, corr, T0)
    g, corr, T0 = out[50]
    K = _kkt()
    reps = K.certify_plan(T0, g.init_traj, m.start, m.goal, m.radius, g.sfc_box, corr.sfc_time, g.sfc_count, g.rsfc_normal,
                          corr.rsfc_time, g.ctrl, True, 8, -1, only_batches=[0, 3, 7], ctrl_before_pass=out[49][0].ctrl)
    assert len(reps) == 3
    for rep in reps:
        tag = f"batch {rep['batch']}: " + ", ".join(f"{k}={v:.3g}" for k, v in rep.items() if isinstance(v, float))
        assert rep["x_as_viol_ineq"] < 1e-7 and rep["x_as_viol_eq"] < 1e-8 and rep["stationarity"] < 1e-7, tag
        assert rep["forward_error"] < CTRL_TOL, tag


def _mixed_mission():
    m = host.load_mission("mission_16agents_15.json")
    m.radius = m.radius.copy()
    m.radius[[3, 4, 7]] = 0.25      # mission.hpp:64: quad_size is per agent (on map1.bt this radius changes the boxes of all three)
    return m


def test_mixed_radius_corridor_bit_exact_and_qp_vs_oracle():
    p = Param.test_sweep()
    m = _mixed_mission()
    w = host.load_world("map1.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref, gpu = init.clone_inputs(), init.clone_inputs()
    rc, ns = O.corridor_update(w, m, p, ref)
    assert rc == 0
    sess = planner.Session([w], [m], p, [gpu])
    sess.run(A.RBP_STAGE_CORRIDOR)
    assert sess.download() == [0]
    assert np.array_equal(ref.sfc_count, gpu.sfc_count) and np.array_equal(ref.sfc_box, gpu.sfc_box)
    assert np.array_equal(ref.sfc_time, gpu.sfc_time)
    assert np.array_equal(bits(ref.rsfc_normal), bits(gpu.rsfc_normal))
    assert int(sess.counters()["sfc_samples"]) == ns
    sess.close()
    # the wider agents really get other boxes than they would with the common radius (the test would be vacuous otherwise)
    m0 = host.load_mission("mission_16agents_15.json")
    same = init.clone_inputs()
    assert O.corridor_update(w, m0, p, same)[0] == 0
    assert all(not np.array_equal(same.sfc_box[a], ref.sfc_box[a]) for a in (3, 4, 7))
    rc, rep = O.planner_update(m, p, ref)
    assert rc == 0
    pl = planner.RBPPlanner(m, p)
    assert pl.update(False, gpu), pl.last_error
    assert gpu.qp_unpolished == 0
    assert np.abs(ref.ctrl - gpu.ctrl).max() < CTRL_TOL
    obj, veq, vbox, vrs = O.evaluate_ctrl(m, gpu)
    assert veq < EQ_TOL and vbox < FEAS_TOL and vrs < FEAS_TOL


def test_corridor_only_session_with_a_joint_batch_wider_than_the_qp_kernel():
    """plan/sequential=false is the reference's code default (param.hpp:67): setBatch makes one batch of all N agents.  For N above
    the ONE-WORKGROUP QP kernel's widest batch the PLANNER stage of that solver is refused -- but Corridor::update has nothing to do with
    the batch width.  (Since round 4 such a joint QP runs on the grid-wide solver, tests/test_gpu_joint.py; rbp_solver_opts.
    joint_wide_min_agents = 0 asks for the one-workgroup kernel.)  A corridor-only call reserves no QP workspace at all."""
    ctx = planner.Context(opts=planner.solver_opts(joint_wide_min_agents=0))
    p = Param.test_sweep(sequential=False, world_x_min=-5, world_y_min=-5, world_x_max=15, world_y_max=5)
    m = host.load_mission("mission_256agents_c4.json")
    w = host.load_world("map1.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref, gpu = init.clone_inputs(), init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    cor = planner.Corridor(w, m, p, ctx)
    assert cor.update(False, gpu), cor.last_error
    assert np.array_equal(ref.sfc_box, gpu.sfc_box) and np.array_equal(bits(ref.rsfc_normal), bits(gpu.rsfc_normal))
    pl = planner.RBPPlanner(m, p, ctx)
    assert pl.update(False, gpu) is False and pl.rc == A.RBP_ERR_BAD_ARGUMENT and "batch wider" in pl.last_error
    ctx.close()


def test_corridor_only_run_after_a_planner_run_is_not_time_scaled():
    """rbp_session_download multiplies T / SFC / RSFC times by time_scale on the host; after a CORRIDOR-only run that factor
    (left by an earlier planner run of the same session) does not apply"""
    p = Param.test_sweep()
    m = host.load_mission("mission_8agents_15.json")
    m.max_vel = m.max_vel * 0.25      # force time_scale > 1
    w = host.load_world("map5.bt", p)
    g = host.ecbs_plan(w, m, p)
    T0 = g.T.copy()
    sess = planner.Session([w], [m], p, [g])
    sess.run(A.RBP_STAGE_ALL)
    assert sess.download() == [0]
    assert g.time_scale > 1.0 and np.allclose(g.T, T0 * g.time_scale, rtol=0, atol=1e-12)
    scaled = g.sfc_time.copy()
    coef_planned, solves = g.coef.copy(), g.qp_solves
    assert solves > 0
    g.coef[:] = -7.0   # (host buffers the corridor-only download must leave alone)
    g.ctrl[:] = -7.0
    sess.run(A.RBP_STAGE_CORRIDOR)
    assert sess.download() == [0]
    assert g.time_scale == 1.0 and np.array_equal(g.T, T0)
    # ... and the earlier planner run's outputs (coef rescaled by ITS time_scale) are not handed out as if they went with this corridor
    assert np.all(g.coef == -7.0) and np.all(g.ctrl == -7.0) and g.qp_solves == 0 and g.total_cost == 0.0
    assert not np.array_equal(coef_planned, g.coef)
    assert np.allclose(g.sfc_time * 1.0, scaled / scaled.max() * g.sfc_time.max(), rtol=1e-12, atol=0)  # same boxes, unscaled
    assert g.sfc_time.max() == T0[-1]
    sess.close()


@pytest.mark.parametrize("nag,pkw", [(8, dict(batch_size=1)), (8, dict(batch_size=3)), (6, dict(batch_size=4)), (7, dict(batch_size=2, iteration=2))])
def test_odd_block_orders_of_the_wave_path_vs_oracle(nag, pkw):
    """the knot step of kernels/knot_lds.inc is instantiated for nk = 9, 18, 27, 36 (batches of 1..4 agents); the golden cases cover 18
    and 36 only.  batch_size 1 and 3 give the ODD orders 9 and 27 (other code paths in the pair-wise LDS loads), 6 agents with
    batch_size 4 a short last batch (36 then 18 inside one mission), 7 agents with batch_size 2 a last batch of one."""
    p = Param.test_sweep(**pkw)
    m = host.load_mission("mission_8agents_15.json")
    if nag < 8:
        m = m.subset(list(range(nag)))
    w = host.load_world("map5.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref, gpu = init.clone_inputs(), init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    rc, rep = O.planner_update(m, p, ref)
    assert rc == 0
    assert planner.Corridor(w, m, p).update(False, gpu)
    pl = planner.RBPPlanner(m, p)
    assert pl.update(False, gpu), pl.last_error
    assert gpu.qp_solves == rep["n_qp"] and gpu.qp_unpolished == 0
    assert np.abs(ref.ctrl - gpu.ctrl).max() < CTRL_TOL
    assert abs(ref.total_cost - gpu.total_cost) <= 1e-8 * max(1.0, abs(ref.total_cost))
    obj, veq, vbox, vrs = O.evaluate_ctrl(m, gpu)
    assert veq < EQ_TOL and vbox < FEAS_TOL and vrs < FEAS_TOL


def test_joint_qp_32_agents_block_order_288_vs_oracle():
    """plan/sequential=false (the reference's code default, param.hpp:67): ONE QP over all 32 agents, knot blocks of order nk = 288 on the
    MFMA-tiled path.  Round 2 compared the joint path with the oracle at 8 and 16 agents (nk 72, 144) only."""
    p = Param.test_sweep(sequential=False)
    m = host.load_mission("mission_32agents_15.json")
    w = host.load_world("map7.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref, gpu = init.clone_inputs(), init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    rc, rep = O.planner_update(m, p, ref)
    assert rc == 0 and rep["n_qp"] == 1 and rep["n_polished"] == 1
    assert planner.Corridor(w, m, p).update(False, gpu)
    pl = planner.RBPPlanner(m, p)
    assert pl.update(False, gpu), pl.last_error
    assert gpu.qp_solves == 1 and (gpu.qp_unpolished == 0 or gpu.kkt_max < 1e-7)
    assert np.abs(ref.ctrl - gpu.ctrl).max() < CTRL_TOL
    assert abs(ref.total_cost - gpu.total_cost) <= 1e-8 * max(1.0, abs(ref.total_cost))
    obj, veq, vbox, vrs = O.evaluate_ctrl(m, gpu)
    assert veq < EQ_TOL and vbox < FEAS_TOL and vrs < FEAS_TOL


def test_longest_first_block_order_does_not_change_results():
    """a session that is run again starts the missions that took longest in the previous run first (DevSession::qp_order); every mission
    is a workgroup of its own, so the plans must be bit-identical to the first run's and to a session with the order switched off"""
    p = Param.test_sweep()
    m = host.load_mission("mission_8agents_15.json")
    maps = ["map5.bt", "map1.bt", "map7.bt", "map3.bt", "map9.bt"]
    worlds = [host.load_world(f, p) for f in maps]
    inits = [host.ecbs_plan(w, m, p) for w in worlds]

    def run(times, **opts):
        plans = [g.clone_inputs() for g in inits]
        sess = planner.Session(worlds, [m] * len(maps), p, plans, opts=planner.solver_opts(**opts))
        outs = []
        for _ in range(times):
            sess.run(A.RBP_STAGE_ALL)
            assert sess.download() == [0] * len(maps)
            outs.append([g.ctrl.copy() for g in plans])
        sess.close()
        return outs

    first, second, third = run(3)
    (plain,) = run(1, qp_block_order=0)
    for a, b, c, d in zip(first, second, third, plain):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) and np.array_equal(a.view(np.uint64), c.view(np.uint64))
        assert np.array_equal(a.view(np.uint64), d.view(np.uint64))


def test_qp_workspace_is_reserved_by_the_first_planner_run():
    """a session that only runs the CORRIDOR stage reserves no QP workspace (round 4 cleared 0.35 GB per 64-agent joint mission
    at session create); the first PLANNER run -- or rbp_session_reserve_workspace -- reserves it for the solver options then in force, and a
    change of options that needs another layout (grid-wide joint solver <-> one workgroup) reserves anew."""
    p = Param.test_sweep(sequential=False)
    m = host.load_mission("mission_16agents_15.json")
    w = host.load_world("map3.bt", p)
    g = host.ecbs_plan(w, m, p)
    sess = planner.Session([w], [m], p, [g])
    sess.run(A.RBP_STAGE_CORRIDOR)
    assert sess.download() == [0] and sess.workspace_bytes_per_mission() == 0
    sess.reserve_workspace()
    wide = sess.workspace_bytes_per_mission()
    assert wide > 0
    sess.run(A.RBP_STAGE_PLANNER)
    assert sess.download() == [0] and g.qp_unpolished == 0
    ctrl_wide = g.ctrl.copy()
    sess.set_solver_opts(joint_wide_min_agents=0)      # the same joint QP on one workgroup: another workspace layout
    sess.reset()
    sess.run(A.RBP_STAGE_ALL)
    assert sess.download() == [0]
    one = sess.workspace_bytes_per_mission()
    assert 0 < one != wide
    assert np.abs(g.ctrl - ctrl_wide).max() < CTRL_TOL
    sess.close()


def test_corridor_lds_need_is_checked_at_session_create():
    """sfc_kernel keeps box_log [max_boxes][M + 1] per agent of a workgroup in LDS; a plan whose max_boxes does not fit is refused
    by rbp_session_create with the numbers in the message, not by the launch with a generic HIP error"""
    p = Param.test_sweep()
    m = host.load_mission("mission_8agents_15.json")
    w = host.load_world("map5.bt", p)
    g = host.ecbs_plan(w, m, p)
    from swarm_simulator_amd.types import PlanResult
    big = PlanResult(g.init_traj.copy(), g.T.copy(), 2000)   # box_log would be 8 agents x 2000 x (M + 1) x 2 bytes > 160 KB
    with pytest.raises(RuntimeError, match="LDS"):
        planner.Session([w], [m], p, [big])
