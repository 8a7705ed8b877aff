#!/bin/bash
# round 4: lane refill on the staged kernels (variant 13) x look-ahead rows / chain rows on c2
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_refill; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_ahead_rows_gpu.py tests/test_top_of_walk_gpu.py -x -q -m gpu -k "variants or refill or fused_class or ahead or fat or top_of" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
run() {  # name, args...
  n=$1; shift
  timeout 600 python3 bench.py --quick --workload c2 --steps 20 --warmup 3 "$@" > $O/$n.json 2> $O/$n.err
  python3 - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-34s %.2f Gbases/s  iter/base %.4f simt %.3f wpc %s kernel %s" % (sys.argv[2], d["value"], c.get("iterations_per_base",0), c.get("simt_efficiency",0), d["roofline"].get("launch",{}).get("waves_per_cu","?"), d["roofline"]["kernel"]))
except Exception as e: print("failed", sys.argv[1:], e)
PY
}
{
run a1_v14 --ahead-rows 1
run a2_v14 --ahead-rows 2    # (historical: chain rows, removed since)
for a in 1 2; do
  for b in 8 16 32; do run a${a}_v13_b$b --ahead-rows $a --variant 13 --opt refill_batch=$b; done
  for w in 7 11 13; do run a${a}_v13_b16_w$w --ahead-rows $a --variant 13 --opt refill_batch=16 --waves-per-cu $w; done
done
} 2>&1 | tee $O/summary.txt
