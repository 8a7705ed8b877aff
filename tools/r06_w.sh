#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_w; mkdir -p $O
for nr in 16000 62500; do echo "== $nr reads per batch"; python3 tools/r06_w.py $nr 2>&1 | grep batches; done | tee $O/streams.txt
echo "== GPU_MAX_HW_QUEUES=8, 62500"; GPU_MAX_HW_QUEUES=8 python3 tools/r06_w.py 62500 2>&1 | grep "own streams" | tee -a $O/streams.txt
echo "== DEBUG_CLR_LIMIT_BLIT_WG / HIP_FORCE_DEV_KERNARG=1, 62500"; HIP_FORCE_DEV_KERNARG=1 python3 tools/r06_w.py 62500 2>&1 | grep "own streams" | tee -a $O/streams.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 tools/r06_w.py 62500 > $O/trace.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
ev=[]
for f in glob.glob(sys.argv[1]+"/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][12:50], r.get("Queue_Id","?"), r.get("Stream_Id","?")))
ev.sort()
t0=ev[-60][0]
for s,e,n,q,st in ev[-40:]:
    print("%9.3f %9.3f %7.3f q%s s%s %s" % ((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,q,st,n))
PY
find $O/trace -type f ! -name "*.csv" -delete
