#!/bin/bash
# A/B of jq_update variants (profiles/r05_joint_lookfirst_ab.txt): every variant is a librbp_hip.so under swarm_simulator_amd/lib/ab/
# (jqp.o of that variant linked with the other objects; the library is chosen through RBP_HIP_LIB, which swarm_simulator_amd/planner.py
# honours for developer builds), all on ONE box.  usage: VARIANTS="v4 v5 v4 v5" AGENTS="256 64" bash tools/experiments/r05_joint_lookfirst_ab.sh
for n in ${AGENTS:-256}; do
for v in ${VARIANTS:-v0 v1 v2 v3 v0 v1 v2 v3}; do
  echo "== $v ($n agents)"
  RBP_HIP_LIB=$PWD/swarm_simulator_amd/lib/ab/librbp_hip_$v.so timeout 300 python tools/joint_shard_replay.py --agents $n 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:round(v,4) for k,v in d.items() if k.endswith('_s')}, 'same bits:', d['rank0_same_bits_as_unsharded'], d['rank1_same_bits_as_unsharded'])"
done
done
