import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from swarm_simulator_amd import host, planner
from swarm_simulator_amd.types import Param
for mf,mid,pkw in [("mission_64agents_15.json",7,dict(batch_size=8, iteration=2)),("mission_64agents_15.json",3,dict(batch_size=12, batch_iter=2, iteration=1)),("mission_64agents_15.json",7,dict(batch_size=8, iteration=50))]:
    p=Param.test_sweep(**pkw); m=host.load_mission(mf); w=host.load_world(f"map{mid}.bt",p)
    init=host.ecbs_plan(w,m,p)
    s=planner.Session([w],[m],p,[init.clone_inputs()])
    import time; t=time.time(); s.run(); st=s.download(); dt=time.time()-t; sc=s.scalars()
    print(pkw, "status",st,"qps",sc[0,3],"polished",sc[0,4],"iters",sc[0,2],"diag code sum",sc[0,7], "time %.2fs"%dt)
