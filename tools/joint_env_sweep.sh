#!/bin/bash
# GPU box (developer library: make -C swarm_simulator_amd/csrc dev; export RBP_HIP_LIB=$PWD/swarm_simulator_amd/lib/librbp_hip_dev.so): the 64-agent joint sweep (maps 1..K, default 50) under several RBP_JQ_* settings: "VAR=value VAR2=value2" per argument
K=${K:-50}
for envs in "$@"; do
  echo "== $envs"
  env $envs REPS=1 timeout 300 python tools/gpu_joint_sweep.py 64 1 $K 2>&1 < /dev/null > /tmp/sweep.txt
  grep -E "missions in|copies" /tmp/sweep.txt
  python - <<'PY'
import re
its=[]; kk=[]
for l in open('/tmp/sweep.txt'):
    m=re.match(r'map(\d+): status (\d+) M (\d+) iters (\d+) unpolished (\d+) kkt (\S+)',l)
    if m and m.group(2)=='0': its.append(int(m.group(4))); kk.append(float(m.group(6)))
if its: print(f"   solved {len(its)}: iterations sum {sum(its)} max {max(its)}; kkt max {max(kk):.2e}")
PY
done
