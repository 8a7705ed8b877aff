#!/bin/bash
# round 5: reposition hints -- the new parity test, then A/B of the timed regions (hints on / off) on c2, c3 and c4real-sized launches
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r05_hints${1:+_$1}; mkdir -p $O
( time timeout 1500 python3 -m pytest tests/test_ahead_rows_gpu.py tests/test_top_of_walk_gpu.py -q -m gpu -x ) > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-14s %.2f Gb/s ms %.3f iter/base %s simt %s scans/base %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c.get("iterations_per_base"),c.get("simt_efficiency"),c.get("scans_per_base"),d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
run c3_hints --workload c3 --steps 5
run c3_nohints --workload c3 --steps 5 --opt repo_hints=0
run c2_hints --workload c2 --steps 20
run c2_nohints --workload c2 --steps 20 --opt repo_hints=0
run c3_hints2 --workload c3 --steps 5
run c3_nohints2 --workload c3 --steps 5 --opt repo_hints=0
} 2>&1 | tee $O/summary.txt
