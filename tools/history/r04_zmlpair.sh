#!/bin/bash
# round 4: the ZML parse with its two windows fetched by pairs of lanes -- parity, then A/B on the 1 B-row table and a real 226 M-row one
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_zmlpair; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_ahead_rows_gpu.py tests/test_gpu_parity.py -q -m gpu -k "zml" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
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
run c4real2_auto --workload c4real2 --steps 5
run c4real2_p0 --workload c4real2 --steps 5 --opt pair_loads=0
run c2_p1 --workload c2 --steps 10 --opt pair_loads=1
} 2>&1 | tee $O/summary.txt
