#!/bin/bash
# GPU box: evidence for the grid-wide joint QP (kernels/jqp.hip) -> gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   bench lines (64 agents x 200 / 50 resident joint missions), rocprofv3 kernel trace of the 200-mission step, PMC pass with the FP64 MFMA
#   counters, single-mission latency at 64 and 256 agents.
# usage: tools/collect_joint_profiles.sh <tag>
set -u
TAG=${1:-r04_joint}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --joint --agents 64 --no-cpu-baseline --no-latency"
timeout 600 $B --missions-per-gpu 200 --steps 2 --warmup 1 > $OUT/bench_200.log 2>&1 < /dev/null
timeout 600 $B --missions-per-gpu 50 --steps 2 --warmup 1 > $OUT/bench_50.log 2>&1 < /dev/null
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --missions-per-gpu 200 --steps 1 --warmup 1 > $OUT/kt.log 2>&1 < /dev/null
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc -- $B --missions-per-gpu 50 --steps 1 --warmup 1 > $OUT/pmc.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
python - <<PY > $OUT/pmc_mfma.txt 2>&1
import csv, glob, collections
fs = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0))[:12]:
    print(k)
    for c, v in agg[k].items(): print(f"   {c}: total {v:.4g} over {n[(k,c)]} dispatches")
PY
rm -rf $OUT/kt $OUT/pmc
for cfg in "64 3" "256 1"; do set -- $cfg; timeout 600 python tools/gpu_joint_wide.py $1 $2 --no-wg --reps 3 > $OUT/single_$1.log 2>&1 < /dev/null; done
ls -la $OUT
