#!/bin/bash
# (historical: "ahead_rows" 2 exists only up to the commit before "chain rows removed"; kept as the record of how profiles/r04_chain_rows.txt was produced)
# round 4: chain rows (look-ahead entries two rows deep, "ahead_rows" 2) against the look-ahead rows ("ahead_rows" 1) on c2
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_chain; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ahead_rows_gpu.py -x -q -m gpu > $O/pytest_ahead.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_ahead.txt
tail -5 $O/pytest_ahead.txt
for a in 1 2; do for w in -1 7 11 13; do
  timeout 600 python3 bench.py --quick --workload c2 --ahead-rows $a --waves-per-cu $w --steps 20 --warmup 3 > $O/c2_a${a}_w${w}.json 2> $O/c2_a${a}_w${w}.err
  python3 - $O/c2_a${a}_w${w}.json $a $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("ahead=%s wpc=%s: %.2f Gbases/s  iter/base %.4f simt %.3f kernel %s" % (sys.argv[2], sys.argv[3], d["value"], c.get("iterations_per_base",0), c.get("simt_efficiency",0), d["roofline"]["kernel"]))
except Exception as e: print("failed", sys.argv[1:], e)
PY
done; done 2>&1 | tee $O/summary.txt
