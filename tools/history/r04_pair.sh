#!/bin/bash
# round 4: pair-shared gathers (pml_kernel_flatp<..., PSH = 1>, "pair_loads" 1) against the default walk
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_pair; mkdir -p $O
MOVI_PAIR_LOADS=1 timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_ahead_rows_gpu.py tests/test_top_of_walk_gpu.py tests/test_pangenome_gpu.py -x -q -m gpu > $O/pytest_pair.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_pair.txt
tail -4 $O/pytest_pair.txt
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f iter/base %s simt %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for p in 0 1; do
run c2_p$p --workload c2 --steps 20 --opt pair_loads=$p
run c2_a0_p$p --workload c2 --steps 20 --ahead-rows 0 --opt pair_loads=$p
run c3_p$p --workload c3 --steps 5 --opt pair_loads=$p
run c2synth_p$p --workload c2synth --steps 20 --opt pair_loads=$p
run c4_200M_p$p --workload c4 --rows 200000000 --steps 10 --opt pair_loads=$p
run c4_p$p --workload c4 --steps 10 --opt pair_loads=$p
done
} 2>&1 | tee $O/summary.txt
