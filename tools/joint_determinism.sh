#!/bin/bash
# GPU box: run the same joint mission several times with the per-round trace and report where the runs diverge
N=${1:-64}; MAP=${2:-3}; REPS=${3:-6}
for i in $(seq 1 $REPS); do
  RBP_HIP_LIB=$PWD/swarm_simulator_amd/lib/librbp_hip_dev.so RBP_JOINT_TRACE=1 python tools/gpu_joint_wide.py $N $MAP --no-wg --reps 1 2>&1 | grep "^\[jqp\] round\|^wide=1" > /tmp/trace_$i.txt
  tail -1 /tmp/trace_$i.txt
done
for i in $(seq 2 $REPS); do
  if ! cmp -s /tmp/trace_1.txt /tmp/trace_$i.txt; then echo "== run $i differs from run 1 at:"; diff /tmp/trace_1.txt /tmp/trace_$i.txt | head -6 | cut -c1-220; fi
done
