#!/bin/bash
# HBM traffic and matrix-pipe occupancy of one roofline kernel: three SEPARATE counter passes (no tracing domains alongside --pmc).
# usage: tools/gpu_pmc_kernel.sh "<decnn.7|decnn.4> <forward|d/d input|d/d weight>" <tag>   -> gpurun_out/pmc_<tag>/
set -e
R=$PWD
K="$1"; T="$2"
O=$R/gpurun_out/pmc_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- python3 $R/tools/roofline_kernel.py "$K" > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- python3 $R/tools/roofline_kernel.py "$K" > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -o r -- python3 $R/tools/roofline_kernel.py "$K" > $O/sq.log 2>&1
cd $R
python3 - "$K" "$T" <<'PY'
import csv, glob, json, sys, collections
kname, tag = sys.argv[1], sys.argv[2]
res = {}
for name in ('fetch', 'write', 'sq'):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pmc_%s/%s/**/*counter_collection.csv' % (tag, name), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r.get('Kernel_Name', '')
            if 'k_conv' in kn or 'k_convT' in kn:          # the kernel itself (not the reduction of the wgrad partials)
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        res[k] = sum(v[1:]) / max(1, len(v[1:]))           # per launch, the first (cold) launch dropped
print(json.dumps({kname: res}, indent=1))
json.dump({kname: res}, open('gpurun_out/pmc_%s/summary.json' % tag, 'w'), indent=1)
PY
