import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O
import os
p=Param.test_sweep(batch_size=8, iteration=1, batch_iter=int(os.environ.get("BI","7"))); m=host.load_mission("mission_64agents_15.json"); w=host.load_world("map3.bt",p)
init=host.ecbs_plan(w,m,p); ref=init.clone_inputs()
O.corridor_update(w,m,p,ref); rc,rep=O.planner_update(m,p,ref)
print("oracle", rc, {k:rep[k] for k in ('n_qp','n_polished','n_loose','iters_total')})
s=planner.Session([w],[m],p,[init.clone_inputs()]); s.run(); st=s.download(); sc=s.scalars()
print("gpu status",st,"qps",sc[0,3],"polished",sc[0,4],"diag",sc[0,7], "polstats", sc[0,20:24])
g=s.plans[0]
print("err", np.abs(ref.ctrl-g.ctrl).max(), "cost", ref.total_cost, g.total_cost)
d=np.abs(ref.ctrl-g.ctrl).max(axis=(1,2)); print("per agent err>1e-6:", np.nonzero(d>1e-6)[0], d[d>1e-6])
