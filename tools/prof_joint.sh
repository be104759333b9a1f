#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel trace of the grid-wide joint QP.  usage: tools/prof_joint.sh <tag> <agents> <map> [reps]
set -u
TAG=${1:-j64}; N=${2:-64}; MAP=${3:-3}; REPS=${4:-3}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/tools/gpu_joint_wide.py $N $MAP --no-wg --reps $REPS > $OUT/kt.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; fi
rm -rf $OUT/kt
grep "wide=1" $OUT/kt.log
if [ -f $OUT/kernel_stats.csv ]; then python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:24]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):7d} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f}%')
PY
fi
