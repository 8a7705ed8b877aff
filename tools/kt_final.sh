#!/bin/bash
# usage: tools/kt_final.sh <outdir>  -- rocprofv3 kernel traces of the shipped kernels on the bench workloads
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() {  # name, bench args
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/$1" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline $2 > "$OUT/$1.log" 2>&1
}
run c2_pml ""
run c3_pml "--workload c3 --steps 5"
run c2_count "--query count"
run c2_zml "--query zml"
run c3_classify "--workload c3 --steps 5 --classify 1"
run c2s_pml "--workload c2s"
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
for f in "$OUT"/*.log; do echo "== $f"; tail -1 "$f"; done >> "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
