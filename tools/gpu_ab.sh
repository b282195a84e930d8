#!/bin/bash
# A/B of an environment switch on the replayed ELBO step.  usage: tools/gpu_ab.sh <tag> <VAR=VALUE> <workload ...>
R=$PWD; T=$1; V=$2; shift 2
O=$R/gpurun_out/$T; mkdir -p $O
for W in "$@"; do
  for rep in 1 2; do
    python3 bench.py --workload $W --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W base   ', d['ms_per_step'])" | tee -a $O/ab.txt
    env $V python3 bench.py --workload $W --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $V', d['ms_per_step'])" | tee -a $O/ab.txt
  done
done
