#!/bin/bash
# A/B of the PAIRED fused update (jq_update<true>: two tiles of a row per workgroup share the fused panel rows; no jq_panel launch in launches of
# many tiles) against the kept variant, one box: lone 256-agent mission, then 64 agents x 50 / 200 resident.  Variant libraries under
# swarm_simulator_amd/lib/ab/ as in tools/experiments/r05_joint_lookfirst_ab.sh.   usage: V="v6 p3 v6 p3" bash tools/experiments/r05_joint_pair_ab.sh
# (the paired kernel itself is not in the tree any more: git show of this commit's parent has no copy either -- the variant was an experiment)
VARIANTS="${V:-v6 p3 v6 p3}" AGENTS="${AGENTS:-256}" bash tools/experiments/r05_joint_lookfirst_ab.sh
for v in ${V:-v6 p3 v6 p3}; do echo "#### $v"; RBP_HIP_LIB=$PWD/swarm_simulator_amd/lib/ab/librbp_hip_$v.so CASES="${CASES:-64:50 64:200}" SCHEDS="0" bash tools/experiments/r05_joint_sched_ab.sh; done
