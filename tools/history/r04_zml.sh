#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_zml; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_ahead_rows_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "zml or ZML" > $O/pytest_zml.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_zml.txt
tail -6 $O/pytest_zml.txt
run() { n=$1; shift
timeout 900 python3 bench.py --quick --steps 10 --warmup 2 "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-28s %.2f Gb/s ms %.3f iter/base %s simt %s ff %.3f scan %.3f kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],c["fast_forwards_per_base"],c["scans_per_base"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
run c2_zml_a0 --workload c2 --query zml --ahead-rows 0
run c2_zml_a1 --workload c2 --query zml --ahead-rows 1
run c2_zml_auto --workload c2 --query zml
run c3_zml_a0 --workload c3 --query zml --ahead-rows 0 --steps 3
run c3_zml_a1 --workload c3 --query zml --ahead-rows 1 --steps 3
run c2synth_zml_a0 --workload c2synth --query zml --ahead-rows 0
run c2synth_zml_a1 --workload c2synth --query zml --ahead-rows 1
run c2synth_zml_auto --workload c2synth --query zml
for w in 12 16 24; do run c2_zml_a1_w$w --workload c2 --query zml --ahead-rows 1 --waves-per-cu $w; done
} 2>&1 | tee $O/summary.txt
