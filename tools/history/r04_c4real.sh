#!/bin/bash
# round 4: a real BWT beyond the Infinity Cache (113.5 M rows, 0.9 GB; built here by tools/build_index, ~10 min): which table layout the walk should use
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=/tmp/movi_bench_cache
O=gpurun_out/r04_c4real; mkdir -p $O
run() { n=$1; shift
timeout 2400 python3 bench.py --quick --workload c4real --steps 10 --warmup 2 "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print("%-22s %.2f Gb/s ms %.3f iter/base %s simt %s ff %.3f scan %.3f frac %.4f index_gen %.0fs kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],c["fast_forwards_per_base"],c["scans_per_base"],r["frac"],c["index_gen_s"],r["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json",".txt").replace(sys.argv[2],"err_"+sys.argv[2])).read()[-500:])
PY
}
{
run a0 --ahead-rows 0
run a1 --ahead-rows 1
run a2 --ahead-rows 2    # (historical: chain rows, removed since)
run auto
run a1_v13 --ahead-rows 1 --variant 13
run a0_w0 --ahead-rows 0 --waves-per-cu 0
run a1_w7 --ahead-rows 1 --waves-per-cu 7
run a1_w12 --ahead-rows 1 --waves-per-cu 12
run a0_noinwin --ahead-rows 0 --opt inwin_repo=0
run count_a0 --query count --ahead-rows 0
run count_a1 --query count --ahead-rows 1
run zml --query zml
run a1_10k --ahead-rows 1 --reads 20000 --read-len 10000 --sub-rate 0.08 --steps 3
} 2>&1 | tee $O/summary.txt
MOVI_BENCH_CACHE=/tmp/movi_bench_cache bash tools/r04_pmc.sh $O/pmc_a0 "--workload c4real --ahead-rows 0" > $O/pmc_a0.txt 2>&1
MOVI_BENCH_CACHE=/tmp/movi_bench_cache bash tools/r04_pmc.sh $O/pmc_a1 "--workload c4real --ahead-rows 1" > $O/pmc_a1.txt 2>&1
tail -3 $O/pmc_a0.txt $O/pmc_a1.txt
