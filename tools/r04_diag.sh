#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_diag; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or refill or fused_class" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
for w in 8 9 12; do for b in 8 16; do
timeout 600 python3 bench.py --quick --workload c2 --steps 10 --warmup 2 --ahead-rows 1 --variant 13 --opt refill_batch=$b --waves-per-cu $w > $O/v13_w${w}_b$b.json 2>$O/err.txt
python3 - $O/v13_w${w}_b$b.json $w $b <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
nw=256*int(sys.argv[2])
print("wpc %s b %s: %.2f Gb/s ms %.3f simt %.3f | mean wave dur %.1f us, max %.1f us"%(sys.argv[2],sys.argv[3],d["value"],d["ms_per_step"],c["simt_efficiency"], c["segments"]/nw/100.0, c["rewalked_reads"]/100.0))
PY
done; done
timeout 600 python3 bench.py --quick --workload c2 --steps 10 --warmup 2 --ahead-rows 1 | grep -o '"value": [0-9.]*'
