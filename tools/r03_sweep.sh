#!/bin/bash
# usage: tools/r03_sweep.sh <outdir>  -- occupancy cap with reads staged through LDS (capacity per lane follows the cap's LDS padding)
OUT=$1; mkdir -p "$OUT"
run() { name=$1; shift; python3 bench.py --quick --steps 20 --warmup 3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err";
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-22s %8.3f Gbases/s  kernel %8.3f ms  staged %s wpc %s  %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"].get("staged"), d["roofline"]["launch"].get("waves_per_cu"), d["roofline"]["kernel"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for W in c2 c4; do
  for wpc in 6 7 8 9 10 11 12 14 16; do run ${W}_wpc$wpc --workload $W --waves-per-cu $wpc; done
done
