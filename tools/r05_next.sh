#!/bin/bash
# round 5: the command line's stage times, c2's occupancy cap with the hints, then the ~0.55 B-row real BWT (c4big): slow test + bench line
cd "$(dirname "$0")/.." || exit 1
bash tools/r05_cli.sh a
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r05_next; mkdir -p $O
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-14s %.2f Gb/s ms %.3f iter/base %s simt %s wpc %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c.get("iterations_per_base"),c.get("simt_efficiency"),d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{ for w in 8 9 10 11 12 14; do run c2_w$w --workload c2 --steps 20 --waves-per-cu $w; done; } 2>&1 | tee $O/summary.txt
bash tools/r05_real_bwt.sh c4big
