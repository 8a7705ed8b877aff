#!/bin/bash
# round 4: config 3 (100 k x 10 kbp: 6 wavefronts per CU, one lane per read) cut into segments whatever the policy says
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_c3seg; mkdir -p $O
run() { n=$1; shift
timeout 1200 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-16s %.2f Gb/s ms %.3f iter/base %s simt %s wpc %s segments %s rewalked %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"],c.get("segments"),c.get("rewalked_reads")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
run c3_base --workload c3 --steps 5
for s in 2048 3328 4992; do run c3_seg$s --workload c3 --steps 5 --opt seg_probe=0 --opt seg_len=$s; done
for w in 8 10 12; do run c3_w$w --workload c3 --steps 5 --opt waves_per_cu=$w; done
} 2>&1 | tee $O/summary.txt
