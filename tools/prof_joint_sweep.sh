#!/bin/bash
# GPU box: rocprofv3 kernel trace of the joint sweep tool.  usage: tools/prof_joint_sweep.sh <tag> <agents> <first> <count>
set -u
TAG=${1:-js}; N=${2:-64}; FIRST=${3:-1}; COUNT=${4:-16}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
REPS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $GRAFT_REPO_ROOT/tools/gpu_joint_sweep.py $N $FIRST $COUNT > $OUT/kt.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; fi
rm -rf $OUT/kt
grep "missions in" $OUT/kt.log
if [ -f $OUT/kernel_stats.csv ]; then python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):7d} total {float(r["TotalDurationNs"])/1e6:9.3f} ms avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f}%')
PY
fi
