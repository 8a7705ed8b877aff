#!/bin/bash
# usage: tools/r03_ab.sh <outdir>  -- round 3 A/B matrix: reads staged through LDS x top-of-walk table x window fetch shape
OUT=$1; mkdir -p "$OUT"
SPLIT=$GRAFT_REPO_ROOT/movi_amd/lib/libmovi_hip_split.so
run() { name=$1; shift; python3 bench.py --quick --steps 20 --warmup 3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err";
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-26s %8.3f Gbases/s  kernel %8.3f ms  iters/base %s  staged %s  %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms_avg"], d["config"]["iterations_per_base"], d["roofline"]["launch"].get("staged"), d["roofline"]["kernel"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for W in c2 c4 c2synth; do
  run ${W}_base --workload $W --kmer-k 0 --stage-reads 0
  run ${W}_stage --workload $W --kmer-k 0 --stage-reads 1
  run ${W}_k12 --workload $W --kmer-k 12 --stage-reads 0
  run ${W}_stage_k12 --workload $W --kmer-k 12 --stage-reads 1
  MOVI_HIP_LIB=$SPLIT run ${W}_split_base --workload $W --kmer-k 0 --stage-reads 0
  MOVI_HIP_LIB=$SPLIT run ${W}_split_stage_k12 --workload $W --kmer-k 12 --stage-reads 1
done
run c3_base --workload c3 --kmer-k 0 --stage-reads 0
run c3_k12 --workload c3 --kmer-k 12 --stage-reads 1
MOVI_HIP_LIB=$SPLIT run c3_split --workload c3 --kmer-k 12 --stage-reads 1
