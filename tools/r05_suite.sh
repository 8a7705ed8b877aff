#!/bin/bash
# round 5: the whole GPU suite + smoke on one box, then the quick timed regions of c2 / c3
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r05_suite${1:+_$1}; mkdir -p $O
( time timeout 3000 python3 -m pytest tests -q -m gpu -x ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
( time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-8s %.2f Gb/s ms %.3f iter/base %s simt %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c.get("iterations_per_base"),c.get("simt_efficiency"),d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{ run c2 --workload c2 --steps 20; run c3 --workload c3 --steps 5; } 2>&1 | tee $O/summary.txt
