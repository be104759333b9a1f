#!/bin/bash
# Developer tool (GPU box): HBM-side bytes per launch of the QP kernel, quick (two PMC passes of a short bench run).
# usage: tools/pmc_quick.sh [missions-per-gpu]   (QP_VARIANT=w2|w4 -> bench.py --qp-variant)
K=${1:-2000}
OUT=$PWD/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python $GRAFT_REPO_ROOT/bench.py --missions-per-gpu $K --steps 1 --warmup 1 --no-cpu-baseline --no-latency > $OUT/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $(dirname $(find $OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1)) $(dirname $(find $OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1)) quicktmp $K | grep -A8 qp_batch
rm -f profiles/quicktmp_pmc.json
