#!/bin/bash
# SQ counters of the decnn.1 kernels (own pass, no tracing domains): matrix-core busy cycles per launch at 512 and 4096 images
set -e
R=$PWD
mkdir -p $R/gpurun_out/pmc1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc1/sq -o r -- python3 $R/tools/time_dec1.py > $R/gpurun_out/pmc1/sq.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc1/sq/**/*counter_collection.csv", recursive=True):
    for kern in ("dec1::k_fwd", "dec1::k_bwd_data", "dec1::k_wgrad"):
        rows = [r for r in csv.DictReader(open(f)) if kern in r.get("Kernel_Name", "")]
        agg = collections.defaultdict(list)
        for r in rows: agg[(r["Counter_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")))].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()): print(kern, k, "launches", len(v), "mean", sum(v) / len(v))
PY
