#!/bin/bash
# Copies what tools/collect_profiles.sh & co. left in gpurun_out/<tag>/ into the tracked profiles/<tag>_* files and prints the numbers
# DESIGN.md / profiles/README.md quote.  usage: tools/store_profiles.sh r02
TAG=${1:-r02}; cd "$(dirname "$0")/.." || exit 1
for f in bench.log pmc_mfma.txt gpu_tests.log other_configs.log sq_counters.txt icache_counters.txt ubench.txt phase_profile.txt; do [ -f gpurun_out/$TAG/$f ] && cp gpurun_out/$TAG/$f profiles/${TAG}_$f; done
cp gpurun_out/$TAG/kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp gpurun_out/$TAG/pmc.json profiles/${TAG}_pmc.json
[ -f gpurun_out/$TAG/pmc_calibrate.txt ] && cp gpurun_out/$TAG/pmc_calibrate.txt profiles/${TAG}_pmc_calibrate.txt
tail -1 profiles/${TAG}_bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],1), d['stage_ms'])
print('algorithmic TB', r['algorithmic_bytes_per_launch']/1e12, 'achieved GB/s', round(r['achieved'],1), 'frac', round(r['frac'],4), 'traffic', r['traffic'])
print('mfma TF', round(r['mfma_achieved_tflops'],3), 'frac', round(r['mfma_frac'],4), 'flops', r['mfma']['flops_per_launch'], 'iterations', r['ipm_iterations_per_step'], 'kkt_max', r['kkt_max'], 'unpolished', r['batch_qps_unpolished_per_step'])
print('latency ms', d['latency_ms_single_mission']['two_calls_ms'], 'cpu 1 core', d['cpu_baseline']['value'], 'all cores', d['cpu_baseline']['all_cores']['value'], d['cpu_baseline']['all_cores']['cores'])"
python -c "
import json, sys; sys.path.insert(0,'.'); import bench; d=json.load(open('profiles/${TAG}_pmc.json'))
print('pmc sha', d['kernel_source_sha'], 'sources now', bench.kernel_source_sha())
[print(' ', k, 'avg', round(v['avg_launch_s'],4), 's  hbm', round(v['hbm_bytes_per_launch']/1e9,2), 'GB ', round(v['hbm_GBps']), 'GB/s') for k,v in d['kernels'].items()]"
head -3 profiles/${TAG}_bench_kernel_stats.csv | cut -c1-150
grep -E "value|^#" profiles/${TAG}_other_configs.log | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/   value \1  ms_per_step \2/'
