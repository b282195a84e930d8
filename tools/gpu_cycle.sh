#!/bin/bash
# One GPU round trip: parity tests, bench (GPU leg) and a kernel-trace profile for cfg2/cfg4.
# usage (on the GPU box, from the repo root): bash tools/gpu_cycle.sh [pytest-args]
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -4
for w in cfg2 cfg4; do
  timeout -k 10 120 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_$w.json
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in cfg2 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -o r1 -- python $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_$w.log 2>&1
done
cd $R && python tools/prof_summary.py gpurun_out/prof_cfg2/r1_kernel_stats.csv gpurun_out/prof_cfg4/r1_kernel_stats.csv
