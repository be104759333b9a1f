#!/bin/bash
# Runs on the GPU box (gpurun): everything profiles/<tag>_* is made of, on the sources as they are -> gpurun_out/<tag>/
#   part "bench":  default bench line, kernel trace, PMC passes (FETCH_SIZE, WRITE_SIZE, FP64 MFMA), SQ counters of the headline kernel
#   part "joint":  joint QP bench lines (200 / 50 resident), kernel trace of the 200-mission step, MFMA + SQ counters, lone missions
#   part "other":  the other BASELINE configurations
#   part "tests":  pytest -m gpu
# usage: tools/collect_round.sh [bench] [joint] [other] [tests]
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
agg() {  # <dir> <kernel substring list, | separated> -> per-kernel counter sums
python - "$1" "$2" <<'PY'
import csv, glob, collections, sys
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
keys = sys.argv[2].split("|")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = next((x for x in keys if x in r["Kernel_Name"]), None)
        if k is None: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in keys:
    if k not in agg: continue
    print(k)
    for c in sorted(agg[k]): print(f"   {c:34s} total {agg[k][c]:.4g} over {n[(k, c)]} dispatches -> {agg[k][c] / n[(k, c)]:.4g} per dispatch")
PY
}
for part in "$@"; do case $part in
bench)
  python bench.py > $OUT/bench.log 2>&1
  B="python $R/bench.py --no-cpu-baseline --no-latency"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $B --steps 2 --warmup 1 > $OUT/pmc_$c.log 2>&1; done
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- $B --steps 2 --warmup 1 > $OUT/pmc_mfma.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq1 -- $B --steps 1 --warmup 1 > $OUT/pmc_sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $B --steps 1 --warmup 1 > $OUT/pmc_sq2.log 2>&1
  cd $R
  f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv
  python tools/pmc_summary.py $(dirname $(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)) $(dirname $(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)) ${TAG}tmp 2000 > $OUT/pmc_summary.txt 2>&1
  mv profiles/${TAG}tmp_pmc.json $OUT/pmc.json 2>/dev/null
  agg $OUT/pmc_mfma "qp_batch_kernel|sfc_kernel" > $OUT/pmc_mfma.txt 2>&1
  { agg $OUT/pmc_sq1 "qp_batch_kernel|sfc_kernel"; agg $OUT/pmc_sq2 "qp_batch_kernel|sfc_kernel"; } > $OUT/sq_counters.txt 2>&1
  rm -rf $OUT/kt $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_mfma $OUT/pmc_sq1 $OUT/pmc_sq2
  ;;
joint)
  J="python $R/bench.py --joint --agents 64 --no-cpu-baseline --no-latency"
  timeout 600 $J --missions-per-gpu 200 --steps 2 --warmup 1 > $OUT/joint_bench_200.log 2>&1 < /dev/null
  timeout 600 $J --missions-per-gpu 50 --steps 2 --warmup 1 > $OUT/joint_bench_50.log 2>&1 < /dev/null
  timeout 900 $J --missions-per-gpu 400 --steps 1 --warmup 1 > $OUT/joint_bench_400.log 2>&1 < /dev/null
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/jkt -- $J --missions-per-gpu 200 --steps 1 --warmup 1 > $OUT/jkt.log 2>&1 < /dev/null
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/jpmc -- $J --missions-per-gpu 200 --steps 1 --warmup 1 > $OUT/jpmc.log 2>&1 < /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/jpmc_$c -- $J --missions-per-gpu 200 --steps 1 --warmup 1 > $OUT/jpmc_$c.log 2>&1 < /dev/null; done
  cd $R
  f=$(find $OUT/jkt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/joint_kernel_stats.csv
  KL="jq_update_bulk|jq_update|jq_panel|jq_pivot0|jq_mv|jq_prep|jq_sweep"
  { agg $OUT/jpmc "$KL"; echo "---- HBM side, KB per dispatch as reported (FETCH_SIZE counts half of the bytes of wide reads on gfx950: MI355X_MICROARCH.md)"; agg $OUT/jpmc_FETCH_SIZE "$KL"; agg $OUT/jpmc_WRITE_SIZE "$KL"; } > $OUT/joint_pmc.txt 2>&1
  python tools/joint_kernel_json.py $OUT/joint_kernel_stats.csv $OUT/joint_bench_200.log $OUT/joint_kernel.json $OUT/joint_pmc.txt > /dev/null 2>&1
  rm -rf $OUT/jkt $OUT/jpmc $OUT/jpmc_FETCH_SIZE $OUT/jpmc_WRITE_SIZE
  timeout 600 python tools/experiments/r05_joint_async_ab.py > $OUT/joint_async_ab.txt 2>&1 < /dev/null
  for cfg in "64 3" "256 1"; do set -- $cfg; timeout 600 python tools/gpu_joint_wide.py $1 $2 --no-wg --reps 3 > $OUT/joint_single_$1.log 2>&1 < /dev/null; done
  ;;
other) bash tools/collect_other_configs.sh $TAG > /dev/null 2>&1 ;;
phase)
  RBP_HIP_LIB=$R/swarm_simulator_amd/lib/librbp_hip_prof.so K=2000 python tools/qp_phase_profile.py > $OUT/phase_profile_2000.txt 2>&1
  RBP_HIP_LIB=$R/swarm_simulator_amd/lib/librbp_hip_prof.so K=1 python tools/qp_phase_profile.py > $OUT/phase_profile_1.txt 2>&1
  ;;
tests) python -m pytest tests -m gpu -q --durations=10 > $OUT/gpu_tests.log 2>&1 ;;
esac; done
ls -la $OUT
