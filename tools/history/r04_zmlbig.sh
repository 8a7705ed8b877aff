#!/bin/bash
# round 4: the ZML parse on tables beyond the TLBs' reach: base-synchronous (variant 0, the policy there) against the lane state machine (1)
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_zmlbig; mkdir -p $O
run() { n=$1; shift
timeout 1200 python3 bench.py --quick --query zml "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-16s %.2f Gb/s ms %.3f rows %d kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
run c4_auto --workload c4 --steps 5
run c4_v1 --workload c4 --steps 5 --zml-variant 1
run c4real2_auto --workload c4real2 --steps 5
run c4real2_v0 --workload c4real2 --steps 5 --zml-variant 0
run c2_auto --workload c2 --steps 10
} 2>&1 | tee $O/summary.txt
