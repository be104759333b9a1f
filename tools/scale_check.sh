#!/bin/bash
# The scaling run as the driver makes it (1 / 2 / 4 / 8 GPUs of ONE node, one rank per GPU over RCCL), runnable as-is on an 8-GPU MI355X node:
#   tools/scale_check.sh [steps] [warmup]          -> gpurun_out/scale_check.jsonl (one bench line per N) + the c4 lines
# Missions shard over the ranks with no data-path collective ("scaling": "weak"); efficiency is for the reader to compute from the values.
# On a box with fewer GPUs the larger N are skipped.  CPU-only rehearsal of the launch + sharding logic: tests/test_multi_cpu.py
# (test_eight_rank_gloo_dry_run_of_the_scaling_bench).
set -u
STEPS=${1:-5}; WARM=${2:-1}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
mkdir -p gpurun_out; OUT=gpurun_out/scale_check.jsonl; : > $OUT
port=29511
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "skip N=$N (only $NGPU GPU(s))"; continue; }
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARM | tail -1 >> $OUT
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((port++)) bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline | tail -1 >> $OUT
  fi
  tail -1 $OUT | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=%d value=%.0f agent-traj/s ms_per_step=%.1f' % (d['n_gpus'], d['value'], d['ms_per_step']))"
done
# BASELINE config C4 (one 256-agent mission, corridor sharded by agent + fused all-gather; --joint: rank pairs share the joint solve)
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && continue
  for J in "" "--joint"; do
    if [ "$N" -eq 1 ]; then python bench.py --config c4 $J --steps 2 | tail -1 >> $OUT
    else python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((port++)) bench.py --config c4 $J --gpus $N --steps 2 | tail -1 >> $OUT; fi
  done
done
# the same joint solve with NO torch in the solve's process: lib/rbp_c4_joint_rank per rank, rank pairs over the stream-ordered RCCL exchange
for N in 2 4 8; do
  [ "$N" -gt "$NGPU" ] && continue
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((port++)) bench.py --config c4 --joint --native-pair --gpus $N --steps 2 | tail -1 >> $OUT
done
echo "wrote $OUT"
