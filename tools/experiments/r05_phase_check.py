"""Developer tool (GPU box): the phase-split schedule of the batch QPs (kernels/qp_phase.inc, rbp_solver_opts.qp_schedule = 2) against one workgroup
per mission (qp_schedule = 1) on the same missions: status, polish counts, control points, cost, iterations; then wall time of both at
K resident missions.   usage: python tools/experiments/r05_phase_check.py [K] [agents] [batch] [iteration]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param

K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
agents = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
iteration = int(sys.argv[4]) if len(sys.argv) > 4 else 1
p = Param.test_sweep(batch_size=bs, iteration=iteration)
m = host.load_mission(f"mission_{agents}agents_15.json")
cache = {}
worlds, inits = [], []
for k in range(K):
    mid = k % 50 + 1
    if mid not in cache:
        w = host.load_world(f"map{mid}.bt", p)
        cache[mid] = (w, host.ecbs_plan(w, m, p))
    worlds.append(cache[mid][0]), inits.append(cache[mid][1])
stream = torch.cuda.current_stream().cuda_stream
res = {}
for path in ("mono", "phase"):
    plans = [g.clone_inputs() for g in inits]
    sess = planner.Session(worlds, [m] * K, p, plans, opts=planner.solver_opts(qp_schedule=2 if path == "phase" else 1, qp_groups=int(os.environ.get("QP_GROUPS", "0"))))
    sess.run(A.RBP_STAGE_ALL, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        sess.reset(stream)
        sess.run(A.RBP_STAGE_ALL, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = sess.download(stream)
    sess.close()
    res[path] = (plans, st, dt)
    print(f"{path}: K={K} {1e3 * dt:.1f} ms per step = {K * m.qn / dt:.0f} agent-traj/s; failed {sum(1 for x in st if x)} status {sorted(set(st))} "
          f"qp_solves {sum(q.qp_solves for q in plans)} unpolished {sum(q.qp_unpolished for q in plans)} iters {sum(q.qp_iterations for q in plans)} "
          f"kkt_max {max(q.kkt_max for q in plans):.2e}", flush=True)
a, b = res["mono"][0], res["phase"][0]
err = max(float(np.abs(x.ctrl - y.ctrl).max()) for x, y in zip(a, b))
rel = max(abs(x.total_cost - y.total_cost) / max(1.0, abs(x.total_cost)) for x, y in zip(a, b))
print(f"phase vs mono: ctrl sup-err {err:.3e} m, rel cost err {rel:.3e}")
ok = err < 2e-6 and rel < 1e-8 and not any(res["phase"][1])
print("PHASE_CHECK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
