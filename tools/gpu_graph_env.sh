#!/bin/bash
# A/B of the HIP runtime's graph-execution switches on the replayed ELBO step (ms per step).  usage: tools/gpu_graph_env.sh <out file>
O=${1:-gpurun_out/graph_env.txt}
run() { # label, env assignments...
  local L=$1; shift
  for W in cfg1 cfg2; do
    local MS=$(env "$@" timeout -k 10 120 python bench.py --workload $W --no-extra --no-cpu-baseline --steps 200 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])" 2>/dev/null)
    echo "$L $W ${MS:-failed}" | tee -a $O
  done
}
run default A=1
run queues1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run queues2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run queues4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run queues8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run batch16 DEBUG_HIP_GRAPH_BATCH_SIZE=16
run batch256 DEBUG_HIP_GRAPH_BATCH_SIZE=256
