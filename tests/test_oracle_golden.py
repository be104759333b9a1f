"""The CPU oracle reproduces the committed golden vectors (regression pin) and certifies them.  CPU only."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests.common import BIG_CASES, MID_CASES, SMALL_CASES, Case, rsfc_hash


@pytest.mark.parametrize("name", SMALL_CASES + MID_CASES + BIG_CASES)
def test_corridor_matches_golden_bit_exact(name):
    c = Case(name)
    assert c.grid_matches(), "distance grid rebuilt from data/worlds differs from the one the golden was made with"
    pr = c.inputs()
    rc, ns = O.corridor_update(c.world, c.mission, c.param, pr)
    assert rc == 0
    g = c.g
    assert np.array_equal(pr.sfc_count, g["sfc_count"])
    assert np.array_equal(pr.sfc_box, g["sfc_box"])          # float64 boxes, bit-exact
    assert np.array_equal(pr.sfc_time, g["sfc_time0"])
    assert np.array_equal(pr.rsfc_time, g["rsfc_time0"])
    assert rsfc_hash(pr) == str(g["rsfc_sha256"])            # float32 normals, bit-exact
    assert ns == int(g["n_samples"])


@pytest.mark.parametrize("name", SMALL_CASES + MID_CASES + BIG_CASES)
def test_planner_matches_golden(name):
    c = Case(name)
    pr = c.inputs()
    rc, _ = O.corridor_update(c.world, c.mission, c.param, pr)
    assert rc == 0
    rc, rep = O.planner_update(c.mission, c.param, pr)
    assert rc == 0
    g = c.g
    assert rep["n_polished"] == rep["n_qp"] == int(g["n_qp"])
    assert np.abs(pr.ctrl - g["ctrl"]).max() < 1e-8
    assert abs(pr.total_cost - float(g["total_cost"])) <= 1e-9 * max(1.0, abs(float(g["total_cost"])))
    assert pr.time_scale == float(g["time_scale"])
    assert np.allclose(pr.T, g["T"], rtol=0, atol=1e-12)
    assert np.abs(pr.coef - g["coef"]).max() < 1e-7
    assert (pr.x_size, pr.eq_size, pr.ineq_size) == tuple(int(v) for v in g["sizes"])
    # KKT certificate of the answer (solver independent): stationarity, feasibility, complementarity, duality gap
    assert rep["kkt_stationarity"] < 1e-7 and rep["kkt_primal_eq"] < 1e-7 and rep["kkt_primal_ineq"] < 1e-9
    assert rep["kkt_dual_min"] >= 0 and rep["kkt_compl"] < 1e-7 and rep["duality_gap_rel"] < 1e-6
    obj, veq, vbox, vrs = O.evaluate_ctrl(c.mission, pr)
    assert veq < 1e-7 and vbox < 1e-9 and vrs < 1e-9


@pytest.mark.parametrize("name", ["c1_4agents_empty_joint", "c1_4agents_empty_seq2", "s4_map1_seq2"])
def test_dense_full_space_lu_agrees_with_null_space_solver(name):
    """two independent linear-algebra paths (full KKT matrix with multipliers vs continuity-eliminated blocks)."""
    c = Case(name)
    a = c.inputs()
    assert O.corridor_update(c.world, c.mission, c.param, a)[0] == 0
    b = a.clone()
    rc1, rep1 = O.planner_update(c.mission, c.param, a, linear_solver=1)
    rc2, rep2 = O.planner_update(c.mission, c.param, b, linear_solver=0)
    assert rc1 == 0 and rc2 == 0
    assert np.abs(a.ctrl - b.ctrl).max() < 1e-7
    assert abs(a.total_cost - b.total_cost) < 1e-8 * max(1, abs(a.total_cost))


def test_interior_point_answer_without_polish_is_close():
    c = Case("s8_map5_seq4")
    a = c.inputs()
    assert O.corridor_update(c.world, c.mission, c.param, a)[0] == 0
    rc, rep = O.planner_update(c.mission, c.param, a, polish=0)
    assert rc == 0 and rep["n_polished"] == 0
    assert np.abs(a.ctrl - c.g["ctrl"]).max() < 1e-3
    assert abs(a.total_cost - float(c.g["total_cost"])) < 1e-5


def test_schedule_is_part_of_the_function():
    """Gauss-Seidel batches (reference order) and the joint QP give different answers."""
    a, b = Case("s4_map1_joint"), Case("s4_map1_seq2")
    assert np.abs(a.g["ctrl"] - b.g["ctrl"]).max() > 1e-3
    assert float(b.g["total_cost"]) > float(a.g["total_cost"])


def test_obstacle_in_initial_trajectory_is_error_1():
    c = Case("s4_map1_joint")
    pr = c.inputs()
    occ = np.argwhere(c.world.dist == 0)[0]
    pos = (np.array(c.world.key_min) + occ + 0.5) * c.world.res
    pr.init_traj[0, 3] = pos
    rc, _ = O.corridor_update(c.world, c.mission, c.param, pr)
    assert rc == 1


def test_colliding_initial_trajectories_is_error_3():
    c = Case("c1_4agents_empty_joint")
    pr = c.inputs()
    pr.init_traj[1] = pr.init_traj[0]
    rc, _ = O.corridor_update(c.world, c.mission, c.param, pr)
    assert rc == 3


def test_unsupported_degree_is_error_11():
    c = Case("c1_4agents_empty_joint")
    pr = c.with_corridor()
    c.param.n = 7
    rc, _ = O.planner_update(c.mission, c.param, pr)
    assert rc == 11
