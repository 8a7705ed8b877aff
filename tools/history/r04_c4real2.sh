#!/bin/bash
# round 4: real BWTs of 113.5 M and ~220 M rows (built here): what the launch policy picks by itself, and the layouts side by side
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=/tmp/movi_bench_cache
O=gpurun_out/r04_c4real2; mkdir -p $O
run() { wl=$1; n=$2; shift; shift
timeout 3000 python3 bench.py --quick --workload $wl --steps 10 --warmup 2 "$@" > $O/${wl}_$n.json 2>$O/err_${wl}_$n.txt
python3 - $O/${wl}_$n.json ${wl}_$n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print("%-22s %.2f Gb/s ms %.3f rows %d iter/base %s simt %s ff %.3f scan %.3f frac %.4f no_ff %s derived %.2f GB gen %.0fs kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],c["iterations_per_base"],c["simt_efficiency"],c["fast_forwards_per_base"],c["scans_per_base"],r["frac"],c.get("no_ff_share"),c.get("derived_table_bytes",0)/1e9,c["index_gen_s"],r["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
run c4real auto
run c4real2 auto
run c4real2 a0 --ahead-rows 0
run c4real2 a1 --ahead-rows 1
run c4real2 count_auto --query count
run c4real2 count_a0 --query count --ahead-rows 0
} 2>&1 | tee $O/summary.txt
MOVI_BENCH_CACHE=/tmp/movi_bench_cache bash tools/r04_pmc.sh $O/pmc_big2_a0 "--workload c4real2 --ahead-rows 0" > $O/pmc_big2_a0.txt 2>&1
MOVI_BENCH_CACHE=/tmp/movi_bench_cache bash tools/r04_pmc.sh $O/pmc_big2_a1 "--workload c4real2 --ahead-rows 1" > $O/pmc_big2_a1.txt 2>&1
free -g | head -2 >> $O/summary.txt
