#!/bin/bash
# Developer tool: clean builds of librbp_hip.so variants for tools/ab_bench.sh.  usage: tools/build_variants.sh name1 "EXTRA1" name2 "EXTRA2" ...
# -> swarm_simulator_amd/lib/<name>/librbp_hip.so (git-ignored).  Builds run in parallel; returns when all are done.
cd "$(dirname "$0")/../swarm_simulator_amd/csrc" || exit 1
while [ $# -ge 2 ]; do
  n=$1; e=$2; shift 2
  rm -rf ../lib/$n
  ( make hip LIBDIR=../lib/$n OBJDIR=../lib/$n/obj EXTRA="$e" > /tmp/build_$n.log 2>&1; grep -q " error" /tmp/build_$n.log && echo "$n: BUILD ERROR" || echo "$n: ok ($e)" ) &
done
wait
