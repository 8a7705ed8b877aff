#!/bin/bash
# round 4: occupancy cap on tables beyond the caches, walked with pair-shared gathers (everything a miss: does the L2-retention cap still apply?)
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_cap; mkdir -p $O
run() { n=$1; shift
timeout 1200 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f rows %d iter/base %s simt %s wpc %s staged %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["launch"].get("staged"),d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for w in 9 11 13 16 20; do run c4_w$w --workload c4 --steps 10 --opt waves_per_cu=$w; done
for w in 9 12 16; do run c4real2_w$w --workload c4real2 --steps 10 --opt waves_per_cu=$w; done
for w in 7 10 13; do run c4_plain_w$w --workload c4 --steps 10 --ahead-rows 0 --opt waves_per_cu=$w; done
} 2>&1 | tee $O/summary.txt
