#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_classify; mkdir -p $O
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-28s %.2f Gb/s ms %.3f fused_classify %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["fused_classify"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for w in c2 c3; do
 st=20; [ $w = c3 ] && st=5
 run ${w}_plain --workload $w --steps $st
 run ${w}_cls1_fused --workload $w --steps $st --classify 1 --opt classify_fused=1
 run ${w}_cls1_twopass --workload $w --steps $st --classify 1 --opt classify_fused=0
 run ${w}_cls2 --workload $w --steps $st --classify 2
done
run c2synth_cls1_fused --workload c2synth --classify 1 --opt classify_fused=1
run c2synth_cls1_twopass --workload c2synth --classify 1 --opt classify_fused=0
} 2>&1 | tee $O/summary.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_pangenome_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "classif" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
