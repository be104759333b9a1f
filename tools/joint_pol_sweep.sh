#!/bin/bash
# GPU box (developer library: make -C swarm_simulator_amd/csrc dev; export RBP_HIP_LIB=.../librbp_hip_dev.so): polish parameter experiments on the 64-agent joint sweep (maps 1..12): polished count per setting
for pol in "$@"; do
  echo "== RBP_JQ_POL=$pol"
  RBP_JQ_POL=$pol timeout 300 python tools/gpu_joint_sweep.py 64 1 12 2>&1 < /dev/null | awk '/missions in/ {print} /^map/ {n++; if ($0 ~ /unpolished 0/) p++; if ($0 !~ /status 0/) f++} END {print "polished", p+0, "of", n, "failed", f+0}'
done
