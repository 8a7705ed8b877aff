#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $O/trace -- python3 tools/r06_t.py > $O/run.log 2>&1
grep -v amdgpu.ids $O/run.log | tail -10
find $O/trace -name "*.csv" | head; 
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "?")), "copy", ""))
for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", "kernel", r["Kernel_Name"][:40]))
ev.sort()
# the last call of each kind: find big gaps (> 20 ms) to split calls
calls, cur = [], []
for e in ev:
    if cur and e[0] - cur[-1][1] > 3_000_000: calls.append(cur); cur = []
    cur.append(e)
if cur: calls.append(cur)
print(len(calls), "bursts")
with open(O + "/timeline.txt", "w") as out:
    for ci in (3, 7) if len(calls) >= 8 else range(len(calls))[-2:]:
        c = calls[ci]; t0 = c[0][0]
        out.write("# burst %d: %d events, %.3f ms\n" % (ci, len(c), (c[-1][1] - t0) / 1e6))
        for s, e, d, k, nm in c:
            out.write("%8.3f %8.3f  %7.3f ms  %-6s %s %s\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k, d, nm))
print(open(O + "/timeline.txt").read()[:6000])
PY
find $O/trace -type f ! -name "*.csv" -delete
