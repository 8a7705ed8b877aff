#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_kmer; mkdir -p $O
run() { n=$1; shift
timeout 600 python3 bench.py --quick --workload c2 --steps 20 --warmup 3 "$@" > $O/$n.json 2>$O/err.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
print("%-28s %.2f Gb/s ms %.3f iter/base %.4f simt %.3f"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"]))
PY
}
for k in 12 11 10 9 8; do run a1_v14_k$k --ahead-rows 1 --kmer-k $k; done
for k in 12 11 10; do run a1_v13_k$k --ahead-rows 1 --kmer-k $k --variant 13 --opt refill_batch=16; done
for k in 12 10; do run a2_v14_k$k --ahead-rows 2 --kmer-k $k; done
for k in 12 10; do run a0_v14_k$k --ahead-rows 0 --kmer-k $k; done
