#!/bin/bash
# round 4: PMLs out through the LDS ring ("out_ring"): parity + A/B on the final kernels
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_ring2; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_ahead_rows_gpu.py tests/test_top_of_walk_gpu.py tests/test_device_entry_gpu.py tests/test_pangenome_gpu.py tests/test_gpu_parity.py -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
run() { n=$1; shift
timeout 1200 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f iter/base %s simt %s wpc %s staged %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["launch"].get("staged")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for r in 0 -1; do
  run c3_ring$r --workload c3 --steps 5 --opt out_ring=$r
  run c3_cls1_ring$r --workload c3 --classify 1 --steps 5 --opt out_ring=$r
  run c3_cls2_ring$r --workload c3 --classify 2 --steps 5 --opt out_ring=$r
done
for r in -1 1; do
  run c2_ring$r --workload c2 --steps 20 --opt out_ring=$r
done
run c3_25k_ring0 --workload c3 --reads 25000 --steps 5 --opt out_ring=0
run c3_25k_ring-1 --workload c3 --reads 25000 --steps 5
} 2>&1 | tee $O/summary.txt
