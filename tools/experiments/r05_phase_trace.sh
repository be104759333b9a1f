#!/bin/bash
# Developer tool (GPU box): kernel trace of the phase-split schedule -> gpurun_out/r05_phase_trace/
set -u
OUT=$PWD/gpurun_out/r05_phase_trace; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for cfg in "2000 1" "50 1"; do set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$1 -- python $GRAFT_REPO_ROOT/bench.py --qp-schedule phase --qp-groups $2 --missions-per-gpu $1 --steps 1 --warmup 1 --no-cpu-baseline --no-latency > $OUT/kt_$1.log 2>&1
  find $OUT/kt_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$1.csv
  rm -rf $OUT/kt_$1
done
cd $GRAFT_REPO_ROOT; cat $OUT/kernel_stats_2000.csv | cut -c1-200; cat $OUT/kernel_stats_50.csv | cut -c1-200
