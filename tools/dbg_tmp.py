import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
np.set_printoptions(linewidth=250, precision=4)
import bench
from swarm_simulator_amd import planner
from swarm_simulator_amd.types import Param
p = Param.test_sweep()
m, worlds, plans = bench.build_inputs([23], 64, p)
s = planner.Session(worlds, [m], p, plans)
s.run(); st = s.download()
print("status", st, s.scalars(28)[0][:12])
tr = plans[0].coef.reshape(-1)[:16 * 100].reshape(100, 16)
print("it gap pres dres |rbase| |cpacc| |Td| |Lf| |rhs| |sol| a_aff mu_aff |rhs2| |sol2| alpha")
for i in range(80):
    print(i, tr[i, :14])
