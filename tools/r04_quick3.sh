#!/bin/bash
# round 4: the three PML workloads, timed region only, on the final tree
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_quick3; mkdir -p $O
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-8s %.2f Gb/s ms %.3f kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{ run c2 --workload c2 --steps 20; run c3 --workload c3 --steps 5; run c4 --workload c4 --steps 10; } 2>&1 | tee $O/summary.txt
