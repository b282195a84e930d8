#!/bin/bash
# kernel-trace profile of one GP draw + rollout + full backward at BASELINE configs[4] shapes (tools/time_flow_bwd.py)
set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg5 -o r1 -- python $R/tools/time_flow_bwd.py --workload cfg5 --reps 2 > $R/gpurun_out/prof_cfg5.log 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_cfg5/r1_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per rep %.2f ms" % (tot/1e6/3))
for r in rows[:25]:
    print("%-80s calls %5s per-rep %9.1fus avg %9.2fus" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"])/1e3/3, float(r["AverageNs"])/1e3))
PY
