#!/bin/bash
# Runs on the GPU box (gpurun): bench line, rocprofv3 kernel trace and the separate PMC passes for profiles/<tag>_*.
# usage: tools/collect_profiles.sh <tag> [missions-per-gpu]
set -u
TAG=${1:-r01}; K=${2:-1000}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --missions-per-gpu $K > $OUT/bench.log 2>&1
python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/bench.py --missions-per-gpu $K --no-cpu-baseline --no-latency > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --missions-per-gpu $K --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- python $GRAFT_REPO_ROOT/bench.py --missions-per-gpu $K --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $OUT/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python tools/pmc_summary.py $(dirname $(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)) $(dirname $(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)) ${TAG}tmp $K > $OUT/pmc_summary.txt 2>&1
mv profiles/${TAG}tmp_pmc.json $OUT/pmc.json 2>/dev/null
python - <<PY > $OUT/pmc_mfma.txt 2>&1
import csv, glob, collections
f = glob.glob("$OUT/pmc_mfma/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c, v in agg[k].items(): print(f"   {c}: total {v:.4g} over {n[(k,c)]} dispatches -> {v / n[(k,c)]:.4g} per dispatch")
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_mfma $OUT/kt
ls -la $OUT
