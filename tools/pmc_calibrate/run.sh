#!/bin/bash
# on the GPU box: FETCH_SIZE / WRITE_SIZE of the calibration kernels (KB per dispatch in the CSV)
OUT=$PWD/gpurun_out/pmc_calib; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
BIN=$PWD/tools/pmc_calibrate/calib
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 -o $BIN $PWD/tools/pmc_calibrate/calib.hip
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- $BIN > $OUT/$c.log 2>&1; done
cd - > /dev/null
python - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and ("read" in r["Kernel_Name"] or "write" in r["Kernel_Name"]):
            print(c, r["Kernel_Name"][:24], "counter KB", float(r["Counter_Value"]), "-> bytes", float(r["Counter_Value"]) * 1024)
PY
cat $OUT/FETCH_SIZE.log | tail -1
