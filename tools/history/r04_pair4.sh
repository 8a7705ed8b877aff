#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_pair; mkdir -p $O
MOVI_PAIR_LOADS=1 timeout 1500 python3 -m pytest tests/test_ahead_rows_gpu.py -x -q -m gpu -k "pair_shared" > $O/pytest_pair2.txt 2>&1; tail -2 $O/pytest_pair2.txt
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
run c4_200M_a1_p0 --workload c4 --rows 200000000 --steps 10 --ahead-rows 1 --opt pair_loads=0
run c4_200M_a1_p1 --workload c4 --rows 200000000 --steps 10 --ahead-rows 1 --opt pair_loads=1
run c4_a1_p1 --workload c4 --steps 10 --ahead-rows 1 --opt pair_loads=1
run c4_auto --workload c4 --steps 10
run c2_auto --workload c2 --steps 20
} 2>&1 | tee $O/summary4.txt
