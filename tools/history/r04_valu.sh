#!/bin/bash
# round 4: how much of the walk's time is VALU issue?  The same kernels with N extra dependent VALU instructions per iteration,
# before the next gather leaves (pre) or under its latency (post): libraries built with -DMOVI_PAD_PRE=N / -DMOVI_PAD_POST=N
# into .ref_<name>/ (git-ignored), selected through MOVI_HIP_LIB.
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_valu; mkdir -p $O
run() { n=$1; lib=$2; shift; shift
MOVI_HIP_LIB=$lib timeout 1200 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-16s %.2f Gb/s ms %.3f iter/base %s simt %s wpc %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for wl in c2 c3 c4; do
for v in base pre40 post40 post80; do
  lib=$PWD/.ref_$v/libmovi_hip.so; [ $v = base ] && lib=$PWD/movi_amd/lib/libmovi_hip.so
  run ${wl}_$v $lib --workload $wl --steps 10
done; done
} 2>&1 | tee $O/summary.txt
