#!/bin/bash
# look-ahead (1) against bulk (2) schedule of the joint tile sweep at several resident-set sizes, ONE box (profiles/r05_joint_lookfirst_ab.txt)
# usage: CASES="64:50 64:200" SCHEDS="2 1 2 1" bash tools/experiments/r05_joint_sched_ab.sh
for c in ${CASES:-64:50 64:200}; do N=${c%%:*}; K=${c##*:}; for S in ${SCHEDS:-2 1 2 1}; do
  echo "== agents=$N K=$K schedule=$S"
  timeout 400 python bench.py --joint --agents $N --no-cpu-baseline --no-latency --missions-per-gpu $K --steps 2 --warmup 1 --joint-schedule $S 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'agent-traj/s', round(d['ms_per_step'],1), 'ms/step ok', d['config'].get('all_missions_ok'))"
done; done
