#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_pair; mkdir -p $O
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f iter/base %s simt %s wpc %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for w in 5 9 12 16; do run c4_p1_w$w --workload c4 --steps 10 --opt pair_loads=1 --waves-per-cu $w; done
for w in 9 12; do run c4_p0_w$w --workload c4 --steps 10 --opt pair_loads=0 --waves-per-cu $w; done
run c4_p1_count --workload c4 --steps 5 --query count --opt pair_loads=1
} 2>&1 | tee $O/summary2.txt
