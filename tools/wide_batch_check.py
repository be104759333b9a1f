import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
from tests import oracle_lib as O
cases=[("mission_8agents_15.json",5,dict(batch_size=8, iteration=2)),("mission_8agents_15.json",5,dict(sequential=False)),
("mission_16agents_15.json",3,dict(sequential=False)),("mission_16agents_15.json",3,dict(batch_size=16, iteration=1)),("mission_64agents_15.json",3,dict(batch_size=12, batch_iter=2, iteration=1)),
("mission_64agents_15.json",7,dict(batch_size=8, iteration=2)),("mission_32agents_15.json",11,dict(batch_size=8, iteration=3)),("mission_16agents_15.json",20,dict(batch_size=6, iteration=2))]
for mf,mid,pkw in cases:
    p=Param.test_sweep(**pkw); m=host.load_mission(mf); w=host.load_world(f"map{mid}.bt",p)
    init=host.ecbs_plan(w,m,p); ref,gpu=init.clone_inputs(),init.clone_inputs()
    O.corridor_update(w,m,p,ref); rc,rep=O.planner_update(m,p,ref)
    planner.Corridor(w,m,p).update(False,gpu); pl=planner.RBPPlanner(m,p); ok=pl.update(False,gpu)
    print(mf,mid,pkw,"ok",ok,rc,"err",np.abs(ref.ctrl-gpu.ctrl).max(),"relcost",abs(ref.total_cost-gpu.total_cost)/abs(ref.total_cost),"feas",O.evaluate_ctrl(m,gpu)[1:], "or.pol",rep['n_polished'],rep['n_qp'])
