#!/bin/bash
# round 5: the count query as a lane state machine (zml_kernel_flat<..., CNT = 1>) -- parity, A/B against count_kernel_v0 on the
# 1 B-row blocked-thresholds table (config 5) and smaller ones, then PMC passes of both kernels on config 5
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r05_count${1:+_$1}; mkdir -p $O
( time timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_ahead_rows_gpu.py -q -m gpu -x -k "count or hints" ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
run() { n=$1; shift
timeout 1500 python3 bench.py --quick --query count "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f rows %d matched %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],c.get("matched_bases_per_step"),d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for v in 0 1; do
run c5_v$v --workload c5 --steps 5 --opt count_variant=$v
run c4_200M_v$v --workload c4 --rows 200000000 --steps 5 --opt count_variant=$v
run c2_v$v --workload c2 --steps 10 --opt count_variant=$v
done
run c5_auto --workload c5 --steps 5
run c5_v1_nopair --workload c5 --steps 5 --opt count_variant=1 --opt pair_loads=0
} 2>&1 | tee $O/summary.txt
if [ "$2" = "pmc" ]; then
bash tools/r05_pmc.sh $O/pmc all "c5_v0:--workload c5 --query count --opt count_variant=0" "c5_v1:--workload c5 --query count --opt count_variant=1"
cp $O/pmc/kernels.txt $O/pmc_kernels.txt
fi
