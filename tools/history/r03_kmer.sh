#!/bin/bash
# usage: tools/r03_kmer.sh <outdir>  -- round 3: top-of-walk table, K sweep on c2 / c3 / c2synth / c4 (one bench.py line each)
OUT=$1; mkdir -p "$OUT"
run() { name=$1; shift; python3 bench.py --quick --steps 20 --warmup 3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; 
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-22s %8.3f Gbases/s  kernel %8.3f ms  iters/base %s  simt %s  %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms_avg"], d["config"]["iterations_per_base"], d["config"]["simt_efficiency"], d["roofline"]["kernel"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for K in 0 6 8 10 11 12; do run c2_k$K --workload c2 --kmer-k $K; done
for K in 0 10 12; do run c3_k$K --workload c3 --kmer-k $K; done
for K in 0 10 12; do run c2synth_k$K --workload c2synth --kmer-k $K; done
for K in 0 11 12; do run c4_k$K --workload c4 --kmer-k $K; done
