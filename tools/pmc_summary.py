"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/<tag>_pmc.json.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced reads ->
doubled here; WRITE_SIZE is used as reported (uncalibrated).  Values are KB per dispatch in the CSV.
usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <tag> <missions_per_gpu>
"""
import collections, csv, glob, json, os, sys

def load(d, counter):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        k = "qp_batch_kernel" if "qp_batch" in n else "rsfc_kernel" if "rsfc_kernel" in n else "sfc_kernel" if "sfc_kernel" in n else None
        if k is None:
            continue
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"]) * 1024.0
        agg[k][2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return agg

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out = {"missions_per_gpu": int(sys.argv[4]), "kernel_source_sha": bench.kernel_source_sha(),
       "note": "bytes per launch; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md); WRITE_SIZE as reported", "kernels": {}}
for k in fetch:
    nf, bf, tf = fetch[k]
    nw, bw, tw = write.get(k, [1, 0.0, 0.0])
    fetch_b, write_b = 2.0 * bf / nf, bw / max(nw, 1)
    out["kernels"][k] = {"launches_profiled": nf, "avg_launch_s": tf / nf, "fetch_bytes_per_launch": fetch_b,
                         "write_bytes_per_launch": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
                         "hbm_GBps": (fetch_b + write_b) / (tf / nf) / 1e9}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", sys.argv[3] + "_pmc.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
