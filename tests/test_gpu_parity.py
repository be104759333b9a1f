"""Parity of the HIP path (through the C ABI) with the CPU oracle and the golden vectors.  Needs an MI355X.

Tolerances (DESIGN.md "Parity"): SFC boxes / times / counts and RSFC normals are integer / float32 procedures
and must be BIT-EXACT.  The QP answer is floating point: control points within CTRL_TOL metres (sup norm) of the
oracle's certified optimum, objective within OBJ_RTOL relative, constraints satisfied within FEAS_TOL.
"""
import numpy as np
import pytest

from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param, PlanResult
from tests import oracle_lib as O
from tests.common import BIG_CASES, MID_CASES, SMALL_CASES, Case, rsfc_hash

pytestmark = pytest.mark.gpu

CTRL_TOL = 2e-6   # metres, sup norm over all control points: GPU (IPM + active-set polish) vs the oracle's certified optimum
OBJ_RTOL = 1e-8
FEAS_TOL = 1e-8


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", SMALL_CASES + MID_CASES + BIG_CASES)
def test_corridor_bit_exact_vs_golden(name):
    c = Case(name)
    pr = c.inputs()
    cor = planner.Corridor(c.world, c.mission, c.param)
    assert cor.update(False, pr), cor.last_error
    g = c.g
    assert np.array_equal(pr.sfc_count, g["sfc_count"])
    assert np.array_equal(pr.sfc_box, g["sfc_box"])
    assert np.array_equal(pr.sfc_time, g["sfc_time0"])
    assert np.array_equal(pr.rsfc_time, g["rsfc_time0"])
    assert rsfc_hash(pr) == str(g["rsfc_sha256"])


@pytest.mark.parametrize("world,nag", [("map2.bt", 16), ("map17.bt", 32), ("ICRA2020_64agents_presentation.bt", 64),
                                       ("empty.bt", 8), ("map50.bt", 64)])
def test_corridor_bit_exact_vs_oracle_on_other_maps(world, nag):
    p = Param.test_sweep()
    m = host.load_mission(f"mission_{nag}agents_15.json")
    w = host.load_world(world, p)
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
    assert int(sess.counters()["sfc_samples"]) == ns   # same getDistance count, same early exits
    sess.close()


@pytest.mark.parametrize("name", SMALL_CASES + MID_CASES + BIG_CASES)
def test_planner_vs_golden(name):
    c = Case(name)
    pr = c.inputs()
    assert planner.Corridor(c.world, c.mission, c.param).update(False, pr)
    pl = planner.RBPPlanner(c.mission, c.param)
    assert pl.update(False, pr), pl.last_error
    g = c.g
    assert np.abs(pr.ctrl - g["ctrl"]).max() < CTRL_TOL
    assert abs(pr.total_cost - float(g["total_cost"])) <= OBJ_RTOL * max(1.0, abs(float(g["total_cost"])))
    assert pr.time_scale == float(g["time_scale"])
    assert np.allclose(pr.T, g["T"], rtol=0, atol=1e-12)
    assert np.abs(pr.coef - g["coef"]).max() < 10 * CTRL_TOL * 3 ** 5  # monomial coefficients amplify by basis * dt^-k
    assert (pr.x_size, pr.eq_size, pr.ineq_size) == tuple(int(v) for v in g["sizes"])
    # solver-independent check of the GPU answer against the reference's constraint sets
    obj, veq, vbox, vrs = O.evaluate_ctrl(c.mission, pr)
    assert veq < FEAS_TOL and vbox < FEAS_TOL and vrs < FEAS_TOL
    assert obj <= float(g["evaluate"][0]) * (1 + OBJ_RTOL) + 1e-9   # no worse than the certified optimum


@pytest.mark.parametrize("pkw", [dict(batch_size=8, iteration=2), dict(sequential=False)])
def test_wide_batches_vs_oracle(pkw):
    """batches of 8 agents (C5-style schedule; joint QP of 8 agents): block order 72 runs on the MFMA-tiled path; the
    active-set polish covers every batch width, so the tolerance is the same as for the reference batch size."""
    p = Param.test_sweep(**pkw)
    m = host.load_mission("mission_8agents_15.json")
    w = host.load_world("map5.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref, gpu = init.clone_inputs(), init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    assert O.planner_update(m, p, ref)[0] == 0
    assert planner.Corridor(w, m, p).update(False, gpu)
    pl = planner.RBPPlanner(m, p)
    assert pl.update(False, gpu), pl.last_error
    assert np.abs(ref.ctrl - gpu.ctrl).max() < CTRL_TOL
    assert abs(ref.total_cost - gpu.total_cost) < OBJ_RTOL * max(1.0, abs(ref.total_cost))
    obj, veq, vbox, vrs = O.evaluate_ctrl(m, gpu)
    assert veq < FEAS_TOL and vbox < FEAS_TOL and vrs < FEAS_TOL


@pytest.mark.parametrize("mission,pkw", [("mission_16agents_15.json", dict(sequential=False)),
                                         ("mission_16agents_15.json", dict(batch_size=16, iteration=1)),
                                         ("mission_64agents_15.json", dict(batch_size=12, batch_iter=2, iteration=1)),
                                         ("mission_64agents_15.json", dict(batch_size=8, iteration=2))])
def test_very_wide_batches_vs_oracle(mission, pkw):
    """batches of more than 8 agents (joint QP of a whole 16-agent mission, BASELINE.json C3-style; batches of 12):
    BASELINE.json C5-style batches of 8 on the 64-agent mission): block orders 72..144 on the MFMA-tiled path."""
    p = Param.test_sweep(**pkw)
    m = host.load_mission(mission)
    w = host.load_world("map3.bt", p)
    init = host.ecbs_plan(w, m, p)
    ref, gpu = init.clone_inputs(), init.clone_inputs()
    assert O.corridor_update(w, m, p, ref)[0] == 0
    assert O.planner_update(m, p, ref)[0] == 0
    assert planner.Corridor(w, m, p).update(False, gpu)
    pl = planner.RBPPlanner(m, p)
    assert pl.update(False, gpu), pl.last_error
    assert np.abs(ref.ctrl - gpu.ctrl).max() < CTRL_TOL
    assert abs(ref.total_cost - gpu.total_cost) < OBJ_RTOL * max(1.0, abs(ref.total_cost))
    obj, veq, vbox, vrs = O.evaluate_ctrl(m, gpu)
    assert veq < FEAS_TOL and vbox < FEAS_TOL and vrs < FEAS_TOL


def test_planner_only_call_with_host_corridor():
    """RBPPlanner::update as a drop-in on a PlanResult whose corridor came from elsewhere (here: the golden)."""
    c = Case("s8_map5_seq4")
    pr = c.with_corridor()
    pl = planner.RBPPlanner(c.mission, c.param)
    assert pl.update(False, pr), pl.last_error
    assert np.abs(pr.ctrl - c.g["ctrl"]).max() < CTRL_TOL


def test_session_batch_equals_single_missions():
    """K missions in one session give the same bits as K separate calls (no cross-talk between workgroups)."""
    p = Param.test_sweep()
    m = host.load_mission("mission_16agents_15.json")
    worlds = [host.load_world(f"map{i}.bt", p) for i in (4, 2, 11)]  # M = 34, 35, 34
    plans = [host.ecbs_plan(w, m, p) for w in worlds]  # every map with its own M = makespan + 2: a ragged session
    assert len({pl.M for pl in plans}) > 1, "pick maps with different makespans"
    singles = [pl.clone_inputs() for pl in plans]
    sess = planner.Session(worlds, [m] * 3, p, plans)
    sess.run()
    assert sess.download() == [0, 0, 0]
    for w, s in zip(worlds, singles):
        assert planner.Corridor(w, m, p).update(False, s)
        assert planner.RBPPlanner(m, p).update(False, s)
    for a, b in zip(plans, singles):
        assert np.array_equal(a.sfc_box, b.sfc_box) and np.array_equal(bits(a.rsfc_normal), bits(b.rsfc_normal))
        assert np.array_equal(a.ctrl, b.ctrl) and np.array_equal(a.coef, b.coef)
    # re-running after reset reproduces the same bits
    first = [a.ctrl.copy() for a in plans]
    sess.reset()
    sess.run()
    sess.download()
    for a, f in zip(plans, first):
        assert np.array_equal(a.ctrl, f)
    sess.close()


def test_full_size_properties():
    """C3 (64 agents): size-independent properties of the GPU answer (reference semantics)."""
    c = Case("c3_64agents_map1")
    pr = c.inputs()
    assert planner.Corridor(c.world, c.mission, c.param).update(False, pr)
    assert planner.RBPPlanner(c.mission, c.param).update(False, pr)
    M, oq = pr.M, 6 * pr.M
    ctrl = pr.ctrl
    # start / goal states (rbp_planner.hpp:408-432) and C2 continuity at every knot (:390-399)
    assert np.array_equal(ctrl[:, :, 0], c.mission.start[:, :3]) and np.array_equal(ctrl[:, :, oq - 1], c.mission.goal[:, :3])
    A_eq = O.Aeq_base(c.g["T0"])
    r = np.einsum("ej,akj->ake", A_eq, ctrl)
    assert np.abs(r[:, :, 6:]).max() < 1e-9
    # every control point inside its SFC box, every pair separated by its RSFC half-space
    obj, veq, vbox, vrs = O.evaluate_ctrl(c.mission, pr)
    assert vbox < FEAS_TOL and vrs < FEAS_TOL
    # the reference's own acceptance signal: safety-margin ratio >= 1 (rbp_publisher.hpp:769-798)
    ratio, dist = host.validate(c.mission, c.param, pr)
    assert ratio >= 1.0
    # sampled trajectory stays inside the world box
    assert ctrl[:, 2].min() >= c.param.world_z_min - FEAS_TOL and ctrl[:, 2].max() <= c.param.world_z_max + FEAS_TOL


def test_error_codes_match_the_reference_failure_sites():
    c = Case("s4_map1_joint")
    pr = c.inputs()
    occ = np.argwhere(c.world.dist == 0)[0]
    pr.init_traj[0, 3] = (np.array(c.world.key_min) + occ + 0.5) * c.world.res
    cor = planner.Corridor(c.world, c.mission, c.param)
    assert cor.update(False, pr) is False and cor.rc == A.RBP_ERR_OBSTACLE_IN_INIT_TRAJ   # rbp_corridor.hpp:181-187
    c2 = Case("c1_4agents_empty_joint")
    pr2 = c2.inputs()
    pr2.init_traj[1] = pr2.init_traj[0]
    cor2 = planner.Corridor(c2.world, c2.mission, c2.param)
    assert cor2.update(False, pr2) is False and cor2.rc == A.RBP_ERR_INIT_TRAJ_COLLIDE    # :385-388
    c3 = Case("c1_4agents_empty_joint")
    c3.param.n = 7
    pl = planner.RBPPlanner(c3.mission, c3.param)
    assert pl.update(False, c3.with_corridor()) is False and pl.rc == A.RBP_ERR_UNSUPPORTED_DEGREE  # rbp_planner.hpp:344-346
    # infeasible QP: shrink an SFC box so that the start point lies outside it  (rbp_planner.hpp:158-161)
    c4 = Case("s4_map1_joint")
    pr4 = c4.with_corridor()
    pr4.sfc_box[0, 0, 3] = pr4.sfc_box[0, 0, 0] + 0.05
    pl4 = planner.RBPPlanner(c4.mission, c4.param)
    assert pl4.update(False, pr4) is False and pl4.rc == A.RBP_ERR_QP_FAILED


def test_sweep_driver_serial_and_batched(capsys):
    """swarm_simulator_amd.test_all — the reference's map-sweep main loop (swarm_traj_planner_rbp_test_all.cpp:49-103) on
    three maps, once with the synchronous calls and once with all maps in one device session: same costs."""
    from swarm_simulator_amd import test_all
    assert test_all.main(["--mission", "mission_16agents_15.json", "--maps", "2-4", "--mode", "serial"]) == 0
    serial = [l for l in capsys.readouterr().out.splitlines() if l.startswith("map") and "QP total cost" in l]
    assert test_all.main(["--mission", "mission_16agents_15.json", "--maps", "2-4", "--mode", "batched"]) == 0
    batched = [l for l in capsys.readouterr().out.splitlines() if l.startswith("map") and "QP total cost" in l]
    assert len(serial) == len(batched) == 3
    cost = lambda l: float(l.split("QP total cost")[1].split()[0])
    ratio = lambda l: float(l.split("safety margin ratio")[1].split()[0])
    span = lambda l: float(l.split("makespan")[1].split()[0])
    for a, b in zip(serial, batched):  # the session keeps every map's own M: same QPs, same answers
        assert span(a) == span(b)
        assert abs(cost(a) - cost(b)) < 1e-9 * max(1.0, cost(a))
        assert ratio(a) >= 1.0 and ratio(b) >= 1.0


@pytest.mark.parametrize("name", ["c2_16agents_map3", "s8_map5_seq4_iter2", "c3_64agents_map1"])
def test_both_kernel_builds_agree(name):
    """the QP kernel is built twice (256 VGPRs, one workgroup per CU / 128 VGPRs, two per CU; csrc/Makefile) and picked per
    launch by the number of resident missions: both must land on the same certified optimum"""
    c = Case(name)
    out = {}
    for variant in ("w2", "w4"):
        ctx = planner.Context(opts=planner.solver_opts(qp_variant=int(variant[1])))
        if c.g["rsfc_normal"].size:
            pr = c.with_corridor()
        else:  # the big golden stores only a hash of the RSFC normals: build the corridor on the GPU
            pr = c.inputs()
            assert planner.Corridor(c.world, c.mission, c.param).update(False, pr)
        pl = planner.RBPPlanner(c.mission, c.param, ctx)
        assert pl.update(False, pr), pl.last_error
        ctx.close()
        assert np.abs(pr.ctrl - c.g["ctrl"]).max() < CTRL_TOL
        out[variant] = pr.ctrl.copy()
    assert np.abs(out["w2"] - out["w4"]).max() < 1e-7
