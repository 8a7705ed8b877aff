#!/bin/bash
# round 6: segment plan -- how many lanes should the segments of a batch make?  (25 k / 50 k / 12.5 k reads of 10 kbp, c2's index)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_q; mkdir -p $O
for cfg in "25000 2048" "25000 2500" "25000 3400" "25000 2048" "50000 2048" "50000 3400" "50000 5000" "50000 1024" "12500 2048" "12500 1250" "12500 1024" "12500 640"; do
  set -- $cfg
  timeout 400 python3 bench.py --quick --workload c3 --reads $1 --seg-len $2 > $O/r$1_seg$2.json 2> /dev/null
  python3 - $O/r$1_seg$2.json <<'PY'
import json,sys,os
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-24s value %.2f ms %.4f segs %s rewalked %s"%(os.path.basename(f), d["value"] or -1, d["ms_per_step"], d["config"]["segments"], d["config"]["rewalked_reads"]))
except Exception as e:
    print(os.path.basename(f),"unreadable",e)
PY
done 2>&1 | tee $O/sweep.txt
