#!/bin/bash
# kernel-trace timeline of one replayed ELBO step.  usage: tools/gpu_timeline.sh <tag> <workload ...>  -> gpurun_out/<tag>/step_timeline_<w>.txt
set -e
R=$PWD
T=${1:-rXX}; shift
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$W -o r1 -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/prof_$W.log 2>&1
  TR=$(find $O/prof_$W -name '*kernel_trace.csv' | head -1)
  ST=$(find $O/prof_$W -name '*kernel_stats.csv' | head -1)
  cp $ST $O/elbo_${W}_kernel_stats.csv
  python3 $R/tools/step_timeline.py $TR > $O/step_timeline_$W.txt
  head -1 $O/step_timeline_$W.txt
  tail -1 $O/prof_$W.log | cut -c1-300
  rm -rf $O/prof_$W
done
