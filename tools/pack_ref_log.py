"""Pack the reference's committed run artefact (swarm_planner/log/coef1..64.csv) into one .npz.

The CSVs are *data* the reference holds: 64 agents x 36 segments,
rows `duration, x^0..x^7, y^0..y^7, z^0..z^7, yaw^0..yaw^7` (rbp_planner.hpp:303-319).
Run in the build container only (needs /root/reference):  python tools/pack_ref_log.py
"""
import glob, os, re, sys
import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/swarm_planner/log"
files = sorted(glob.glob(os.path.join(src, "coef*.csv")), key=lambda p: int(re.findall(r"coef(\d+)", p)[0]))
dur, coef = [], []
for f in files:
    rows = [l.strip().rstrip(",").split(",") for l in open(f).read().strip().split("\n")[1:]]
    a = np.array(rows, dtype=np.float64)
    dur.append(a[:, 0])
    coef.append(a[:, 1:25].reshape(-1, 3, 8))           # [M][dim][ascending power 0..7]
out = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ref_log_coef.npz")
np.savez_compressed(out, duration=np.array(dur), coef=np.array(coef))
print("wrote", out, np.array(coef).shape)
