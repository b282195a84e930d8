run() { L=$1; shift; env "$@" GPODE_BENCH_FORCE_DIST=1 timeout -k 10 120 python bench.py --no-extra --no-cpu-baseline --steps 10 $EXTRA > /dev/null 2> gpurun_out/r3v/bis_$L.err; echo "$L rc=$?"; }
EXTRA="" run base A=1
EXTRA="" run eagerred GPODE_EAGER_REDUCTIONS=1
EXTRA="" run bn1 GPODE_BN_ONE_LAUNCH=0
EXTRA="--no-sync-bn" run nosyncbn A=1
EXTRA="--dp-graph fwdbwd" run fwdbwd A=1
EXTRA="" run nomarker GPODE_NO_MARKER=1
