#!/bin/bash
# SQ counters of the roofline kernel (own pass, no tracing domains): matrix-core busy cycles against wave cycles
set -e
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc/sq -o r -- python3 $R/tools/roofline_kernel.py > $R/gpurun_out/pmc/sq.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc/sq/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_conv_igemm" in r.get("Kernel_Name", "")]
    agg = collections.defaultdict(list)
    for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, sum(v[1:]) / max(1, len(v[1:])))
PY
