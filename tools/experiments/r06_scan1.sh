cd /root/repo
L=swarm_simulator_amd/lib
RBP_HIP_LIB=$PWD/$L/pst/librbp_hip.so python tools/experiments/r05_polstats.py 2>&1 | tail -2
bash tools/ab_bench.sh $L/librbp_hip.so $L/e5/librbp_hip.so $L/e3/librbp_hip.so $L/g4/librbp_hip.so $L/b6/librbp_hip.so $L/s4/librbp_hip.so $L/f9/librbp_hip.so -- --no-latency 2>&1 | tail -16
