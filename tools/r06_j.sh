#!/bin/bash
# round 6: ZML / count state machine with its reads staged through LDS: A/B ("stage_reads" 0 / 1), occupancy
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_j; mkdir -p $O
for q in count zml; do for st in 0 1; do
  timeout 400 python3 bench.py --quick --workload c2 --query $q --opt stage_reads=$st > $O/c2_${q}_st$st.json 2> /dev/null
done; done
for q in count zml; do for cap in 12 20 24; do
  timeout 400 python3 bench.py --quick --workload c2 --query $q --waves-per-cu $cap > $O/c2_${q}_st1_cap$cap.json 2> /dev/null
done; done
for q in count zml; do for st in 0 1; do
  timeout 900 python3 bench.py --quick --workload c4 --query $q --opt stage_reads=$st > $O/c4_${q}_st$st.json 2> /dev/null
done; done
for q in count zml; do for cap in 12 20; do
  timeout 900 python3 bench.py --quick --workload c4 --query $q --waves-per-cu $cap > $O/c4_${q}_st1_cap$cap.json 2> /dev/null
done; done
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    print("%-28s value %.2f kernel_ms %.4f cap %s | %s"%(os.path.basename(f), d["value"] or -1, d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"]["waves_per_cu"], d["roofline"]["kernel"]))
PY
