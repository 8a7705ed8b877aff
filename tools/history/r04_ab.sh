#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_ab; mkdir -p $O
run() { n=$1; shift
timeout 900 python3 bench.py --quick --steps 20 --warmup 3 "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-28s %.2f Gb/s ms %.3f iter/base %.4f simt %.3f"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
for v in u1 u2 u4; do
MOVI_HIP_LIB=$PWD/.ref_$v/libmovi_hip.so run c2_$v --workload c2 --ahead-rows 1
MOVI_HIP_LIB=$PWD/.ref_$v/libmovi_hip.so run c2_v13_$v --workload c2 --ahead-rows 1 --variant 13 --opt refill_batch=16
MOVI_HIP_LIB=$PWD/.ref_$v/libmovi_hip.so run c3_$v --workload c3 --ahead-rows 1 --steps 5
done
run c2_v13_cur --workload c2 --ahead-rows 1 --variant 13 --opt refill_batch=16
