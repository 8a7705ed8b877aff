#!/bin/bash
# usage: tools/r03_sweep.sh <outdir>  -- occupancy cap with reads staged through LDS (capacity per lane follows the cap's LDS padding)
OUT=$1; mkdir -p "$OUT"
run() { name=$1; shift; python3 bench.py --quick --steps 20 --warmup 3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err";
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-26s %8.3f Gbases/s  kernel %8.3f ms  staged %s wpc %s  %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"].get("staged"), d["roofline"]["launch"].get("waves_per_cu"), d["roofline"]["kernel"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for wpc in 3 4 5 6 7 8; do run c2_wpc$wpc --workload c2 --waves-per-cu $wpc; done
for wpc in 4 5 6 7 9; do run c2synth_wpc$wpc --workload c2synth --waves-per-cu $wpc; done
for wpc in 4 5 6 7 9; do run r200M_wpc$wpc --workload c4 --rows 200000000 --waves-per-cu $wpc; done
for wpc in 4 5 6 7 9; do run c2_400k_wpc$wpc --workload c2 --reads 400000 --waves-per-cu $wpc; done
for wpc in 4 5; do run c4_wpc$wpc --workload c4 --waves-per-cu $wpc; done
