# A/B of the joint solver: lib/librbp_hip_base.so against the current build on ONE box: bitwise control points of nine joint missions, then
# the joint bench at 200 and 50 resident missions and one 256-agent mission (C4 --joint)
cd /root/repo
L=swarm_simulator_amd/lib
RBP_HIP_LIB=$PWD/$L/librbp_hip_base.so python tools/experiments/r05_joint_dump_ctrl.py /tmp/j_base.npy 2>&1 | tail -1
python tools/experiments/r05_joint_dump_ctrl.py /tmp/j_new.npy 2>&1 | tail -1
python -c "
import numpy as np
a=np.load('/tmp/j_base.npy'); b=np.load('/tmp/j_new.npy'); print('joint control points bit-identical:', np.array_equal(a,b))"
for rep in 1 2; do for lib in librbp_hip_base.so librbp_hip.so; do
  for K in 200 50; do
    RBP_HIP_LIB=$PWD/$L/$lib python bench.py --joint --missions-per-gpu $K --steps 2 --no-cpu-baseline --no-latency 2>&1 | tail -1 | grep -o "\"value\": [0-9.]*\|\"planner\": [0-9.]*" | tr "\n" " "; echo " <- $lib K=$K"
  done
  RBP_HIP_LIB=$PWD/$L/$lib python bench.py --config c4 --joint --steps 1 2>&1 | tail -1 | grep -o "\"value\": [0-9.]*\|ms_per_step\": [0-9.]*" | tr "\n" " "; echo " <- $lib c4 joint"
done; done
