"""Developer tool (GPU box): control points of JOINT missions (plan/sequential = false) through the library RBP_HIP_LIB names -> <out>.npy, to
compare two builds of kernels/jqp.hip bit for bit.  Cases: 64 agents x maps 1..6 in one session (look-ahead launches of many tiles), one
64-agent, one 32-agent and one 16-agent mission alone (fused-panel launches).  usage: python tools/experiments/r05_joint_dump_ctrl.py <out>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep(sequential=False)
chunks, info = [], []
for n, maps in ((64, [1, 2, 3, 4, 5, 6]), (64, [7]), (32, [7]), (16, [3])):
    m, worlds, plans = bench.build_inputs(maps, n, p)
    s = planner.Session(worlds, [m] * len(maps), p, plans)
    s.run(); st = s.download(); s.close()
    for g in plans:
        chunks.append(np.ascontiguousarray(g.ctrl).view(np.uint64).reshape(-1))
        info.append((n, int(g.qp_iterations), int(g.qp_unpolished), float(g.total_cost)))
    assert not any(st), st
np.save(sys.argv[1], np.concatenate(chunks))
print(info)
