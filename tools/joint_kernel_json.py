"""profiles/<tag>_joint_kernel.json: the rate of the joint solver's dominant kernel (jq_update / jq_update_bulk) from a rocprofv3 kernel trace of
`bench.py --joint --agents 64 --missions-per-gpu 200 --steps 1 --warmup 1` (two steps in the trace) and the flops the solver logs per step
(the bench line of the same workload), tied to the joint solver's sources by hash (bench.py prints it only while the hash matches).

usage: python tools/joint_kernel_json.py <joint_kernel_stats.csv> <joint_bench_200.log> <out.json> [joint_pmc.txt] [nblk nj]
(joint_pmc.txt: the per-kernel FETCH_SIZE / WRITE_SIZE sums tools/collect_round.sh writes; gives hbm_bytes_per_launch of the update kernel with the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md: the counters are in KB, FETCH_SIZE counts half of the bytes of wide reads)"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats, benchlog, out = sys.argv[1:4]
rest = sys.argv[4:]
pmc_txt = rest.pop(0) if rest and not rest[0].isdigit() else None
nblk, nj = (int(rest[0]), int(rest[1])) if len(rest) > 1 else (9, 35)
rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
# the update kernel of the schedule in force: jq_update (look-ahead, the automatic choice since round 5's lean kernel) or jq_update_bulk
cands = [(float(r["TotalDurationNs"]), ("jq_update_bulk" if "jq_update_bulk" in n else "jq_update"), r) for n, r in rows.items()
         if "jq_update_bulk(" in n or "jq_update(" in n]
_, kname, k = max(cands, key=lambda c: c[0])
line = [l for l in open(benchlog).read().splitlines() if l.startswith("{")][-1]
b = json.loads(line)
flops = b["roofline"]["flops_per_step"]
# the logged flops by kernel (kernels/jqp.hip jq_count): per knot nblk 64-column steps of (nblk-1) nblk / 2 update tiles and (nblk-1) panel
# tiles, 2 * 64^3 flops each; per iteration 2 (2 nj - 1) products with an explicit inverse, 2 nkp^2 flops each
tile = 2.0 * 64 ** 3
upd, pan = nj * nblk * (nblk - 1) * nblk / 2 * tile, nj * nblk * (nblk - 1) * tile
sol = 2.0 * (2 * nj - 1) * 2.0 * (64 * nblk) ** 2
share = upd / (upd + pan + sol)
secs = float(k["TotalDurationNs"]) * 1e-9
base = os.path.join(ROOT, "swarm_simulator_amd", "csrc", "kernels")
h = hashlib.sha256()
for f in sorted(os.listdir(base)):
    if f.startswith("jqp"):
        h.update(f.encode()), h.update(open(os.path.join(base, f), "rb").read())
tflops = share * flops * 2 / secs / 1e12
traffic = None
if pmc_txt and os.path.exists(pmc_txt):
    import re
    txt = open(pmc_txt).read()
    # blocks "kernel\n   COUNTER total X over N dispatches -> Y per dispatch"; the HBM part follows the "---- HBM side" line
    hbm = txt.split("---- HBM side")[-1]
    def per_dispatch(counter):
        mm = re.search(r"^" + re.escape(kname) + r"\n(?:   .*\n)*?   " + counter + r"\s+total\s+\S+ over \d+ dispatches -> (\S+) per dispatch", hbm, re.M)
        return float(mm.group(1)) if mm else None
    fe, wr = per_dispatch("FETCH_SIZE"), per_dispatch("WRITE_SIZE")
    if fe is not None and wr is not None:
        traffic = {"fetch_bytes_per_launch": 2.0 * fe * 1024.0, "write_bytes_per_launch": wr * 1024.0,
                   "hbm_bytes_per_launch": 2.0 * fe * 1024.0 + wr * 1024.0,
                   "correction": "FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of the bytes of wide reads (x2)"}
json.dump({"traffic": traffic,"joint_source_sha": h.hexdigest()[:16], "missions_per_gpu": b["config"]["missions_per_gpu"], "agents": b["config"]["agents"],
           "kernel": kname, "launches": int(k["Calls"]), "kernel_seconds_two_steps": secs, "logged_flops_per_step": flops,
           "share_of_logged_flops_in_this_kernel": share,
           "share_derivation": f"tile counts of jq_count (kernels/jqp.hip): update tiles (nblk-1) nblk / 2 against panel tiles (nblk-1) per 64-column step, "
                               f"plus the substitutions' 2 (2 nj - 1) 2 nkp^2; nblk = {nblk}, nj = {nj}",
           "kernel_time_includes": "the polish's S_AA sweeps (kind 1 launches of the same kernel), whose flops are NOT among the logged ones: the rate is a lower bound" if kname == "jq_update" else "the knots' sweeps only",
           "tflops": tflops, "frac_of_fp64_mfma_peak": tflops / 78.6,
           "note": "rocprofv3 --kernel-trace --stats of `bench.py --joint --agents 64 --missions-per-gpu 200 --steps 1 --warmup 1` (two steps in the "
                   "trace): the *_joint_kernel_stats.csv beside this file"}, open(out, "w"), indent=1)
print(open(out).read())
