"""Developer tool (GPU box): the grid-wide joint solver against committed oracle vectors, case by case.
usage: python tools/experiments/r05_joint_vs_golden.py [--schedule=N] <npz> [<npz> ...]     (joint64_sweep.npz | joint32_sweep.npz | joint_heldout.npz)"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from swarm_simulator_amd import _abi as A
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
p = Param.test_sweep(sequential=False)
SCHED = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--schedule=")), 0)
for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    gold = np.load(os.path.join(GOLDEN, name))
    n = len(gold["cost"])
    if "mission" in gold.files:
        cases = [(str(gold["mission"][i]), str(gold["world"][i])) for i in range(n)]
    else:
        cases = [("mission_64agents_15.json", f"map{i + 1}.bt") for i in range(n)]
    by_n = {}
    for i, (mf, wf) in enumerate(cases):
        if int(gold["rc"][i]) != 0:
            print(f"{name} {mf} {wf}: oracle rc {int(gold['rc'][i])}, skipped")
            continue
        by_n.setdefault(host.load_mission(mf).qn, []).append(i)
    worst_err = worst_rel = 0.0
    n_unpol = n_bad = 0
    for N, idx in sorted(by_n.items()):
        missions = [host.load_mission(cases[i][0]) for i in idx]
        worlds = [host.load_world(cases[i][1], p) for i in idx]
        inits = [host.ecbs_plan(w, m, p) for w, m in zip(worlds, missions)]
        plans = [g.clone_inputs() for g in inits]
        sess = planner.Session(worlds, missions, p, plans, opts=planner.solver_opts(joint_schedule=SCHED))
        t = time.time(); sess.run(A.RBP_STAGE_ALL); st = sess.download(); dt = time.time() - t
        sess.close()
        agents = [int(a) for a in gold["agents"]] if "agents" in gold.files else [0, N // 3, (2 * N) // 3, N - 1]
        for k, i in enumerate(idx):
            g, m = plans[k], missions[k]
            sha = hashlib.sha256(np.ascontiguousarray(inits[k].init_traj).tobytes()).hexdigest()
            same = sha == str(gold["init_traj_sha256"][i])
            if st[k] != 0 or not same or g.M != int(gold["M"][i]):
                print(f"{name} {cases[i]}: status {st[k]} initTraj same {same} M {g.M} vs {int(gold['M'][i])}")
                n_bad += 1
                continue
            err = float(np.abs(gold["ctrl"][i][:, :, :6 * g.M] - g.ctrl[agents]).max())
            rel = abs(float(gold["cost"][i]) - g.total_cost) / max(1.0, abs(g.total_cost))
            obj, veq, vbox, vrs = O.evaluate_ctrl(m, g)
            flag = "" if (err < 2e-6 and rel < 1e-8 and g.qp_unpolished == 0) else "   <<<<"
            if flag or os.environ.get("ALL"):
                print(f"{name} {cases[i][0]} {cases[i][1]}: unpolished {g.qp_unpolished} oracle polished {int(gold['polished'][i])} iters {g.qp_iterations} kkt {g.kkt_max:.2e} "
                      f"ctrl err {err:.3e} cost rel {rel:.3e} viol eq {veq:.1e} box {vbox:.1e} rsfc {vrs:.1e}{flag}")
            worst_err, worst_rel = max(worst_err, err), max(worst_rel, rel)
            n_unpol += g.qp_unpolished
            n_bad += 1 if flag else 0
        print(f"{name}: {len(idx)} missions of {N} agents in {dt:.2f}s")
    print(f"{name}: cases {n} outside tolerance {n_bad} unpolished {n_unpol} worst ctrl err {worst_err:.3e} worst cost rel {worst_rel:.3e}")
