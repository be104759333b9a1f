#!/bin/bash
# Developer tool (GPU box): instruction-cache counters of the QP kernel (one PMC pass; no trace domains).  usage: tools/pmc_icache.sh [K]
K=${1:-2000}
OUT=$PWD/gpurun_out/pmcic; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/g1 -- python $GRAFT_REPO_ROOT/bench.py --missions-per-gpu $K --steps 1 --warmup 1 --no-cpu-baseline --no-latency > $OUT/g1.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qp_batch" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg): print(f"{k:32s} {agg[k] / n[k]:.4g} per launch ({n[k]} launches)")
PY
rm -rf $OUT
