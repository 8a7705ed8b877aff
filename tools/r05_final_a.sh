#!/bin/bash
# round 5, final tree: (1) the launch policy on the 0.68 B-row real BWT (c4big): look-ahead rows on / off x pair-shared gathers on / off, the cap, count kernels;
# (2) PMC passes for profiles/traffic.json: every leg of the default bench line that has a roofline object
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=/tmp/movi_bench_cache
mkdir -p $MOVI_BENCH_CACHE; cp -rn .bench_cache/* $MOVI_BENCH_CACHE/ 2>/dev/null
O=gpurun_out/r05_final_a; mkdir -p $O
run() { n=$1; shift
timeout 1800 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-18s %.2f Gb/s ms %.3f rows %d iter/base %s simt %s wpc %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],c.get("iterations_per_base"),c.get("simt_efficiency"),d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
run c4big_default --workload c4big --steps 10
run c4big_ahead0 --workload c4big --steps 10 --ahead-rows 0
run c4big_pair0 --workload c4big --steps 10 --opt pair_loads=0
run c4big_ahead0_pair0 --workload c4big --steps 10 --ahead-rows 0 --opt pair_loads=0
run c4big_hints0 --workload c4big --steps 10 --opt repo_hints=0
run c4big_w7 --workload c4big --steps 10 --waves-per-cu 7
run c4big_w12 --workload c4big --steps 10 --waves-per-cu 12
run c4big_count --workload c4big --steps 5 --query count
run c4big_count_v0 --workload c4big --steps 5 --query count --opt count_variant=0
run c4big_count_pair0 --workload c4big --steps 5 --query count --opt pair_loads=0
run c4big_zml --workload c4big --steps 5 --query zml
} 2>&1 | tee $O/summary.txt
export MOVI_BENCH_CACHE=$PWD/.bench_cache
bash tools/r05_pmc.sh $O/pmc all "c2:--workload c2"
bash tools/r05_pmc.sh $O/pmc2 mem "c4:--workload c4" "c3_classify1:--workload c3 --classify 1" "c3_classify2:--workload c3 --classify 2" "c2_count:--workload c2 --query count" "c2_zml:--workload c2 --query zml"
cat $O/pmc/kernels.txt $O/pmc2/kernels.txt > $O/pmc_kernels.txt
