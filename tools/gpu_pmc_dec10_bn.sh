#!/bin/bash
# HBM traffic of the fused decnn.10 / BatchNorm backward passes: two separate counter passes (no tracing domains alongside --pmc)
set -e
R=$PWD
mkdir -p $R/gpurun_out/pmc10
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc10/fetch -o r -- python3 $R/tools/dec10_bn_kernel.py > $R/gpurun_out/pmc10/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc10/write -o r -- python3 $R/tools/dec10_bn_kernel.py > $R/gpurun_out/pmc10/write.log 2>&1
cd $R
python3 - <<PY
import csv, glob
for name in ("fetch", "write"):
    for f in glob.glob("gpurun_out/pmc10/%s/**/*counter_collection.csv" % name, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_bwd_data_bn" in r.get("Kernel_Name", "")]
        for r in rows[-2:]:
            print(name, r.get("Kernel_Name")[:50], r.get("Counter_Name"), r.get("Counter_Value"))
PY
