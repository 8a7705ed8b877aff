OUT=gpurun_out/r3m
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() {  # name, bench args
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/$1" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-long-reads $2 > "$OUT/$1.log" 2>&1
}
run seg_25k "--workload c3synth --reads 25000"
run seg_ragged "--workload c3synth --ragged 1"
run noseg_25k "--workload c3synth --reads 25000 --seg-len 0"
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
for f in "$OUT"/*.log; do echo "== $f"; tail -1 "$f" | cut -c1-400; done >> "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
cat "$OUT/summary.txt" | cut -c1-250
