#!/bin/bash
# tests -> bench (GPU leg) -> kernel-trace profile of the full ELBO step; every step gated on the previous one
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_vae_layers.py tests/test_gpu_model.py -q -x 2>&1 | tail -3
timeout -k 10 300 python bench.py --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_elbo_cfg2.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_elbo_cfg2 -o r1 -- python $R/bench.py --workload cfg2 --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_elbo.log 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_elbo_cfg2/r1_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step %.2f ms" % (tot/1e6/22))
for r in rows[:22]:
    print("%-66s calls %5s per-step %7.1fus avg %8.2fus" % (r["Name"][:66], r["Calls"], float(r["TotalDurationNs"])/1e3/22, float(r["AverageNs"])/1e3))
PY
