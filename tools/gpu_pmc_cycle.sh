#!/bin/bash
# HBM traffic of the roofline kernel: two separate counter passes (no tracing domains alongside --pmc)
set -e
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -o r -- python3 $R/tools/roofline_kernel.py > $R/gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -o r -- python3 $R/tools/roofline_kernel.py > $R/gpurun_out/pmc/write.log 2>&1
cd $R
ls -R gpurun_out/pmc | head -30
python3 - <<PY
import csv, glob
for name in ("fetch", "write"):
    for f in glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % name, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_conv_igemm" in r.get("Kernel_Name", "")]
        print(name, f, len(rows), [(r.get("Counter_Name"), r.get("Counter_Value")) for r in rows[:6]])
PY
