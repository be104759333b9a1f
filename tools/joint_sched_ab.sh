#!/bin/bash
# GPU box: the three schedules of the tile sweep (JOINT_SCHEDULE = 1 look-ahead | 2 bulk: rbp_solver_opts.joint_schedule) on one 64-agent mission, one 256-agent mission
# and a session of K 64-agent missions
K=${1:-50}
for sch in 1 2; do
  echo "== JOINT_SCHEDULE=$sch"
  JOINT_SCHEDULE=$sch timeout 300 python tools/gpu_joint_wide.py 64 3 --no-wg --reps 3 2>&1 < /dev/null | grep -E "wide=1|feas"
  JOINT_SCHEDULE=$sch timeout 300 python tools/gpu_joint_wide.py 256 1 --no-wg --reps 2 2>&1 < /dev/null | grep -E "wide=1|feas"
  JOINT_SCHEDULE=$sch REPS=1 timeout 400 python tools/gpu_joint_sweep.py 64 1 $K 2>&1 < /dev/null | grep -E "missions in|copies"
done
