# A/B of the current build against lib/librbp_hip_base.so on ONE box: phase profile (prof twin), bench, bitwise control points (50 maps)
cd /root/repo
L=swarm_simulator_amd/lib
RBP_HIP_LIB=$PWD/$L/librbp_hip_prof.so K=2000 python tools/qp_phase_profile.py 2>&1 | tail -19
bash tools/ab_bench.sh $L/librbp_hip_base.so $L/librbp_hip.so -- --no-latency 2>&1 | tail -4
python tools/experiments/r05_dump_ctrl.py /tmp/ctrl_new.npy 50 2>&1 | tail -1
RBP_HIP_LIB=$PWD/$L/librbp_hip_base.so python tools/experiments/r05_dump_ctrl.py /tmp/ctrl_base.npy 50 2>&1 | tail -1
python -c "
import numpy as np
a=np.load('/tmp/ctrl_new.npy'); b=np.load('/tmp/ctrl_base.npy'); print('bit-identical control points:', np.array_equal(a.view(np.uint64), b.view(np.uint64)), np.abs(a-b).max())"
