#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_pair; mkdir -p $O
run() { n=$1; shift
timeout 1200 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f rows %d iter/base %s simt %s wpc %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
run c4_a1_p1_b --workload c4 --steps 10 --ahead-rows 1 --opt pair_loads=1
run c4_500M_a1_p1_b --workload c4 --rows 500000000 --steps 10 --ahead-rows 1 --opt pair_loads=1
run c4_700M_a1_p1 --workload c4 --rows 700000000 --steps 10 --ahead-rows 1 --opt pair_loads=1
run c4_700M_a0_p1 --workload c4 --rows 700000000 --steps 10 --ahead-rows 0 --opt pair_loads=1
run c4_350M_a1_p1 --workload c4 --rows 350000000 --steps 10 --ahead-rows 1 --opt pair_loads=1
run c4_350M_a0_p1 --workload c4 --rows 350000000 --steps 10 --ahead-rows 0 --opt pair_loads=1
run c4_a1_p0 --workload c4 --steps 10 --ahead-rows 1 --opt pair_loads=0
} 2>&1 | tee $O/summary6.txt
