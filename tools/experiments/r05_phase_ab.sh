#!/bin/bash
# Developer tool (GPU box): A/B of the two schedules of the batch QPs on the same box -> gpurun_out/r05_phase_ab.log
mkdir -p gpurun_out; OUT=gpurun_out/r05_phase_ab.log; : > $OUT
python tools/experiments/r05_phase_check.py 12 >> $OUT 2>&1 || { echo "check failed" >> $OUT; tail -30 $OUT; exit 1; }
python tools/experiments/r05_phase_check.py 1 >> $OUT 2>&1
python tools/experiments/r05_phase_check.py 50 >> $OUT 2>&1
for cfg in "mono 1" "phase 1" "phase 2" "phase 4" "phase 8"; do set -- $cfg
  echo "== --qp-schedule $1 --qp-groups $2" >> $OUT
  timeout 600 python bench.py --qp-schedule $1 --qp-groups $2 --no-cpu-baseline --no-latency --steps 3 2>&1 | tail -1 | \
    grep -o "\"value\": [0-9.]*\|\"value_first_run\": [0-9.]*\|ipm_iterations_per_step\": [0-9.]*\|unpolished_per_step\": [0-9]*\|\"corridor\": [0-9.]*\|\"planner\": [0-9.]*" | tr "\n" " " >> $OUT; echo >> $OUT
done
tail -40 $OUT
