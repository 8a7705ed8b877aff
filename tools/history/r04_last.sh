#!/bin/bash
# round 4, last checks: the look-ahead rows' build time on 1 B rows (kernel trace), cap re-sweep on c2 with the final kernel
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_last; mkdir -p $O
R=$PWD; cd /tmp && export TMPDIR=/tmp && cd "$R"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_c4 -- python3 bench.py --quick --workload c4 --steps 5 --warmup 1 > $O/kt_c4.log 2>&1
python3 tools/prof_summary.py $O/kt_c4 > $O/kt_c4_summary.txt 2>&1; find $O -name "*.db" -delete
grep "KERNEL" $O/kt_c4_summary.txt | head -6 | cut -c1-170
run() { n=$1; shift
timeout 1200 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-16s %.2f Gb/s ms %.3f wpc %s staged %s"%(sys.argv[2],d["value"],d["ms_per_step"],d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["launch"].get("staged")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{ for w in 8 9 10 11 12; do run c2_w$w --workload c2 --steps 20 --opt waves_per_cu=$w; done; } 2>&1 | tee $O/summary.txt
