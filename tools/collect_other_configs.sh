#!/bin/bash
# Runs on the GPU box (gpurun): bench lines of the other BASELINE configurations and resident-set sizes for profiles/<tag>_other_configs.log
# usage: tools/collect_other_configs.sh <tag>
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
L=$OUT/other_configs.log; : > $L
run() { echo "# $*" >> $L; "$@" 2>&1 | tail -1 >> $L; }
run python bench.py --config c2 --no-cpu-baseline
run python bench.py --config c5 --no-cpu-baseline
run python bench.py --config c4 --no-cpu-baseline
run python bench.py --agents 16 --joint --missions-per-gpu 250 --steps 2 --no-cpu-baseline
run python bench.py --missions-per-gpu 50 --no-cpu-baseline
run python bench.py --missions-per-gpu 250 --no-cpu-baseline
run python bench.py --missions-per-gpu 500 --no-cpu-baseline
run python bench.py --config c4 --joint --no-cpu-baseline
run python bench.py --agents 64 --joint --missions-per-gpu 50 --steps 2 --no-cpu-baseline
cat $L
