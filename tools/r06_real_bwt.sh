#!/bin/bash
# round 6 (final tree): the slow part of the GPU suite (real BWTs of 113 M / 226 M rows against the oracle at size), then bench.py on both
# WITHOUT --quick (cpu_baseline = parity sample of the same launch) -> profiles/r06_real_bwt.txt
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=/tmp/movi_bench_cache
mkdir -p $MOVI_BENCH_CACHE; cp -rn .bench_cache/* $MOVI_BENCH_CACHE/ 2>/dev/null
O=gpurun_out/r06_real_bwt; mkdir -p $O
WLS=${1:-c4real c4real2}
K=$(echo $WLS | sed 's/ / or /g')
( time MOVI_SLOW_TESTS=1 timeout 3000 python3 -m pytest tests/test_real_bwt_gpu.py -q -m gpu -x -k "$K" ) > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for w in $WLS; do
  ( time timeout 1500 python3 bench.py --workload $w --no-big-table --no-long-reads --no-sustained ) > $O/bench_$w.json 2> $O/bench_$w.err
  python3 - $O/bench_$w.json $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-8s %.2f Gb/s ms %.3f rows %d kernel %s parity_sample_ok %s cpu_baseline %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],d["roofline"]["kernel"],d.get("parity_sample_ok", d.get("cpu_baseline",{}).get("parity_sample_ok")),d.get("cpu_baseline",{}).get("value")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
done 2>&1 | tee $O/summary.txt
