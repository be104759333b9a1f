"""Developer tool: run one golden case on the GPU and print polish diagnostics."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from swarm_simulator_amd import planner, _abi as A
from tests.common import Case
name = sys.argv[1]
c = Case(name)
pr = c.inputs()
s = planner.Session([c.world], [c.mission], c.param, [pr])
s.run(); print("status", s.download())
sc = s.scalars()[0]
print("iters", sc[2], "qps", sc[3], "polished", sc[4], "fail codes", sc[7])
print("ctrl diff %.3e cost %.12f golden %.12f" % (np.abs(pr.ctrl - c.g["ctrl"]).max(), pr.total_cost, float(c.g["total_cost"])))
