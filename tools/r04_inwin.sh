#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_inwin; mkdir -p $O
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
for rep in 1 2; do
MOVI_HIP_LIB=$PWD/.ref_r03/libmovi_hip.so run c2_r03_$rep --workload c2 --ahead-rows 1
run c2_i1_$rep --workload c2 --ahead-rows 1 --opt inwin_repo=1
run c2_i0_$rep --workload c2 --ahead-rows 1 --opt inwin_repo=0
MOVI_HIP_LIB=$PWD/.ref_r03/libmovi_hip.so run c3_r03_$rep --workload c3 --ahead-rows 1 --steps 5
run c3_i1_$rep --workload c3 --ahead-rows 1 --steps 5 --opt inwin_repo=1
run c3_i0_$rep --workload c3 --ahead-rows 1 --steps 5 --opt inwin_repo=0
MOVI_HIP_LIB=$PWD/.ref_r03/libmovi_hip.so run c2synth_r03_$rep --workload c2synth --ahead-rows 1
run c2synth_i1_$rep --workload c2synth --ahead-rows 1 --opt inwin_repo=1
run c2synth_i0_$rep --workload c2synth --ahead-rows 1 --opt inwin_repo=0
done
