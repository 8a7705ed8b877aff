#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=/tmp/movi_bench_cache
O=gpurun_out/r04_pair; mkdir -p $O
run() { n=$1; shift
timeout 3000 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f rows %d iter/base %s simt %s wpc %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for a in 0 1; do for p in 0 1; do run big2_a${a}_p$p --workload c4real2 --steps 10 --ahead-rows $a --opt pair_loads=$p; done; done
for a in 0 1; do for p in 0 1; do run big_a${a}_p$p --workload c4real --steps 10 --ahead-rows $a --opt pair_loads=$p; done; done
} 2>&1 | tee $O/summary3.txt
