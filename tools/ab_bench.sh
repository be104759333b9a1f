#!/bin/bash
# Developer tool (GPU box): A/B of several builds of librbp_hip.so on the SAME box (boxes of the pool differ by up to +-15 %).
# usage: tools/ab_bench.sh <lib> <lib> ... [-- bench args]   -- runs the list twice, prints value / stage times / latency of each
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2; do for lib in "${LIBS[@]}"; do
  RBP_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 2 "$@" 2>&1 | tail -1 | grep -o "\"value\": [0-9.]*\|ipm_iterations_per_step\": [0-9.]*\|unpolished_per_step\": [0-9]*\|corridor\": [0-9.]*\|planner\": [0-9.]*\|two_calls_ms[^,]*" | tr "\n" " "; echo " <- $lib"
done; done
