#!/bin/bash
# GPU box: run the 64-agent joint sweep (maps 1..12) under several environment settings: "VAR=value VAR2=value2" per argument
for envs in "$@"; do
  echo "== $envs"
  env $envs REPS=1 timeout 300 python tools/gpu_joint_sweep.py 64 1 12 2>&1 < /dev/null | grep -E "missions in|copies"
done
