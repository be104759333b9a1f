"""How often does the one conscious deviation of timeScale change the answer?  (CPU)

roots_derivative (rbp_planner.hpp:727-754) inspects the first i = 2 of the three eigenvalues of the companion matrix of the velocity's
derivative, in the order Eigen's EigenSolver returns them (:747).  Eigen is absent and that order is not reproducible, so the product and
the oracle use ALL real roots.  Both rules are restated in tests/golden/make_kkt_reference.py (numpy; LAPACK's eigenvalue order stands in
for Eigen's) and run over the 50 maps of the sweep (64 agents, batch 4) with max_vel / max_acc scaled by 1, 0.5 and 0.25.
"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

from swarm_simulator_amd import host
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

SCALES = (1.0, 0.5, 0.25)


def _one(mid):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_kkt_reference as K
    p = Param.test_sweep()
    m = host.load_mission("mission_64agents_15.json")
    w = host.load_world(f"map{mid}.bt", p)
    pr = host.ecbs_plan(w, m, p)
    T0 = pr.T.copy()
    assert O.corridor_update(w, m, p, pr)[0] == 0
    big = host.load_mission("mission_64agents_15.json")
    big.max_vel = m.max_vel * 1e6   # no scaling inside the oracle: the coefficients stay those of the QP answer
    big.max_acc = m.max_acc * 1e6
    rc, _ = O.planner_update(big, p, pr)
    assert rc == 0 and pr.time_scale == 1.0
    out = []
    for sc in SCALES:
        ref = K.time_scale_of(pr.coef, T0, m.max_vel * sc, m.max_acc * sc, "reference")
        ours = K.time_scale_of(pr.coef, T0, m.max_vel * sc, m.max_acc * sc, "all_real")
        # and the oracle's own timeScale (the C restatement the GPU is compared with) agrees with the numpy "all real roots" rule
        lim = host.load_mission("mission_64agents_15.json")
        lim.max_vel, lim.max_acc = m.max_vel * sc, m.max_acc * sc
        chk = pr.clone()
        chk.T[:] = T0
        ts_oracle = O.time_scale(lim, chk)
        out.append((mid, sc, ref, ours, ts_oracle))
    return out


def test_all_real_roots_vs_first_two_eigenvalues_on_the_50_map_sweep():
    workers = max(1, min(50, (os.cpu_count() or 2) - 1))
    with ProcessPoolExecutor(max_workers=workers) as ex:
        rows = [r for out in ex.map(_one, range(1, 51)) for r in out]
    differ = [(mid, sc, ref, ours) for mid, sc, ref, ours, _ in rows if ref != ours]
    print(f"\ntimeScale: {len(rows)} (map, limit scale) cases; reference rule != all-real-roots rule in {len(differ)}: {differ[:8]}")
    for mid, sc, ref, ours, ts_oracle in rows:
        assert ours >= ref, (mid, sc)                 # more candidate times can only find a larger peak
        assert abs(ours - ts_oracle) < 1e-12, (mid, sc, ours, ts_oracle)
    # the recorded outcome: at the mission's own limits (scale 1) the two rules agree on all 50 maps; with the limits halved /
    # quartered they differ on 4 / 2 maps (6 of 150 cases), every time by ONE step of the 1.1 ladder (the skipped third eigenvalue held
    # the velocity peak)
    assert not [d for d in differ if d[1] == 1.0], differ
    assert len(differ) <= 10, differ
    for mid, sc, ref, ours in differ:
        assert abs(ours / ref - 1.1) < 1e-9, (mid, sc, ref, ours)
