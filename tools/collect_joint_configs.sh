#!/bin/bash
# Runs on the GPU box: the JOINT lines of profiles/<tag>_other_configs.log again (after a change to kernels/jqp.hip) -> gpurun_out/<tag>/other_configs_joint.log
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
L=$OUT/other_configs_joint.log; : > $L
run() { echo "# $*" >> $L; "$@" 2>&1 | tail -1 >> $L; }
run python bench.py --agents 16 --joint --missions-per-gpu 250 --steps 2 --no-cpu-baseline
run python bench.py --agents 16 --joint --missions-per-gpu 1000 --steps 2 --no-cpu-baseline
run python bench.py --config c4 --joint --no-cpu-baseline
cat $L | cut -c1-260
