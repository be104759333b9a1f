#!/bin/bash
# Developer tool (no GPU needed): compile-time resource usage of every kernel -> profiles/<tag>_resource_usage.txt
#   * hipcc -Rpass-analysis=kernel-resource-usage (registers, scratch bytes per lane, spills, LDS, occupancy) per kernel;
#   * scratch instructions (scratch_load/scratch_store) per FUNCTION from the ISA: the QP kernel is many __noinline__ functions, and a
#     spill inside a loop costs more than a prologue's.
# usage: tools/resource_usage.sh r03
TAG=${1:-r05}; cd "$(dirname "$0")/../swarm_simulator_amd/csrc" || exit 1
OUT=../../profiles/${TAG}_resource_usage.txt
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -Ikernels -Iabi --cuda-device-only"
one() {  # name, source, extra flags
  echo; echo "## $2 $1"
  hipcc $FL $3 -Rpass-analysis=kernel-resource-usage -c -o /dev/null $2 2>&1 | grep -E "Function Name|SGPRs:|VGPRs:|ScratchSize|Occupancy|Spill|LDS Size" | sed 's/^.*remark: [^ ]* //; s/^ *//'
  hipcc $FL $3 -S -o /tmp/ru_$$.s $2 2>/dev/null
  echo "# scratch instructions per function (ISA)"
  awk '/^_Z[A-Za-z0-9_]*:/ {name=$1} /scratch_(load|store)/ {n[name]++} /; NumVgprs:/ {v[name]=$3} END {for (k in v) printf "  %-100s scratch instr %4d  vgprs %s\n", substr(k,1,100), n[k]+0, v[k]}' /tmp/ru_$$.s | sort
  rm -f /tmp/ru_$$.s
}
{
  echo "# compile-time resource usage, $TAG (tools/resource_usage.sh: hipcc -Rpass-analysis=kernel-resource-usage; per-function scratch instruction counts from the ISA)"
  one "build _w2 (-DQP_WAVES_PER_EU=2 -DQP_ROW_PF=4 -DQP_SUFFIX=_w2)" kernels/qp.hip "-DQP_WAVES_PER_EU=2 -DQP_ROW_PF=4 -DQP_SUFFIX=_w2"
  one "build _w4 (-DQP_THREADS=256 -DQP_WAVES_PER_EU=2 -DQP_SUFFIX=_w4)" kernels/qp.hip "-DQP_THREADS=256 -DQP_WAVES_PER_EU=2 -DQP_SUFFIX=_w4"
  one "(every jq_* / jp_* kernel of the grid-wide joint solver)" kernels/jqp.hip ""
  one "" kernels/corridor.hip ""
  one "" kernels/edt.hip ""
} > $OUT
wc -l $OUT
