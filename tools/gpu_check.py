"""Developer check on a GPU box: HIP path vs CPU oracle on one mission (not part of the test-suite)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

nag = int(os.environ.get("NAG", "64"))
world = os.environ.get("WORLD", "map1.bt")
p = Param.test_sweep(**({"batch_iter": int(os.environ["BI"])} if "BI" in os.environ else {}))
m = host.load_mission(f"mission_{nag}agents_15.json")
w = host.load_world(world, p)
pr = host.ecbs_plan(w, m, p)
print("N", m.qn, "M", pr.M, planner.lib().rbp_version(), "devices", planner.lib().rbp_device_count())
ref = pr.clone_inputs(); t = time.time(); rc, ns = O.corridor_update(w, m, p, ref); print("oracle corridor rc", rc, "samples", ns, "%.3fs" % (time.time() - t))
gpu = pr.clone_inputs(); t = time.time(); ok = planner.Corridor(w, m, p).update(False, gpu); print("gpu corridor ok", ok, "%.3fs" % (time.time() - t))
print(" sfc_count equal", np.array_equal(ref.sfc_count, gpu.sfc_count), "box equal", np.array_equal(ref.sfc_box, gpu.sfc_box),
      "time equal", np.array_equal(ref.sfc_time, gpu.sfc_time), "rsfc equal", np.array_equal(ref.rsfc_normal.view(np.uint32), gpu.rsfc_normal.view(np.uint32)),
      "rsfc_time", np.array_equal(ref.rsfc_time, gpu.rsfc_time))
if not np.array_equal(ref.sfc_box, gpu.sfc_box):
    bad = np.argwhere(ref.sfc_box != gpu.sfc_box); print(bad[:10]); 
t = time.time(); rc, rep = O.planner_update(m, p, ref, polish=int(os.environ.get("POLISH", "1"))); print("oracle planner rc", rc, "%.2fs" % (time.time() - t), rep)
g2 = ref.clone(); g2.ctrl[:] = 0; g2.coef[:] = 0
g2.T[:] = pr.T; g2.sfc_time[:] = gpu.sfc_time; g2.rsfc_time[:] = gpu.rsfc_time
t = time.time(); pl = planner.RBPPlanner(m, p); ok = pl.update(False, g2); print("gpu planner ok", ok, pl.last_error, "%.2fs" % (time.time() - t))
print(" cost oracle %.9f gpu %.9f  iters %d  time_scale %s %s" % (ref.total_cost, g2.total_cost, g2.qp_iterations, ref.time_scale, g2.time_scale))
print(" ctrl sup diff %.3e  coef sup diff %.3e" % (np.abs(ref.ctrl - g2.ctrl).max(), np.abs(ref.coef - g2.coef).max()))
print(" gpu evaluate (obj, eq, box, rsfc):", O.evaluate_ctrl(m, g2), " oracle:", O.evaluate_ctrl(m, ref))
print(" sizes", (g2.x_size, g2.eq_size, g2.ineq_size), (ref.x_size, ref.eq_size, ref.ineq_size))
