#!/bin/bash
# One round's measurement set on the MI355X box: default bench line, kernel-trace statistics + step timelines of configs[1] and
# configs[0], and the three --pmc passes of the roofline kernels.  usage: tools/gpu_round_profile.sh <tag>   -> gpurun_out/<tag>/
set -e
R=$PWD
T=${1:-rXX}
O=$R/gpurun_out/$T
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
for W in cfg2 cfg1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o r1 -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/prof_$W.log 2>&1
  TR=$(find $O/prof_$W -name '*kernel_trace.csv' | head -1)
  ST=$(find $O/prof_$W -name '*kernel_stats.csv' | head -1)
  cp $ST $O/elbo_${W}_kernel_stats.csv
  python3 $R/tools/step_timeline.py $TR > $O/step_timeline_$W.txt
  head -1 $O/step_timeline_$W.txt
  rm -rf $O/prof_$W
done
cd $R
for K in "decnn.4 d/d input" "decnn.7 d/d weight" "decnn.7 forward"; do
  TAG=$(echo "$T $K" | tr ' ./' '___')
  tools/gpu_pmc_kernel.sh "$K" $TAG > $O/pmc_$(echo "$K" | tr ' ./' '___').json 2> /dev/null || true
  rm -rf gpurun_out/pmc_$TAG/fetch gpurun_out/pmc_$TAG/write gpurun_out/pmc_$TAG/sq
done
ls $O
