#!/bin/bash
# round 4: pair-shared gathers in the count query's interval shrink -- parity + A/B on the 1 B-row tables (config 5) and smaller ones
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_countpair; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_ahead_rows_gpu.py -q -m gpu -k "count" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { n=$1; shift
timeout 1500 python3 bench.py --quick --query count "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f rows %d kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for p in 0 1; do
run c5_p$p --workload c5 --steps 5 --opt pair_loads=$p
run c4_200M_p$p --workload c4 --rows 200000000 --steps 5 --opt pair_loads=$p
run c4real2_p$p --workload c4real2 --steps 5 --opt pair_loads=$p
run c2_p$p --workload c2 --steps 10 --opt pair_loads=$p
done
run c5_a1_p1 --workload c5 --steps 5 --opt pair_loads=1 --ahead-rows 1
} 2>&1 | tee $O/summary.txt
