#!/bin/bash
# usage: tools/r03_count_zml_sweep.sh <outdir> -- occupancy cap / block size of the count and ZML kernels (round 3: do they want the cap the PML kernel has?)
OUT=$1; mkdir -p "$OUT"
run() { name=$1; shift; python3 bench.py --quick --steps 20 --warmup 3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err";
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s %8.3f Gbases/s  kernel %8.3f ms  %s %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["kernel"], d["roofline"]["launch"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for bt in 256 64; do for wpc in 0 8 12 16 24; do run c2_count_bt${bt}_wpc$wpc --workload c2 --query count --block-threads $bt --waves-per-cu $wpc; done; done
for wpc in 0 6 8 10 12 16; do run c2_zml_wpc$wpc --workload c2 --query zml --waves-per-cu $wpc; done
for wpc in 0 12; do run c5_count_wpc$wpc --workload c5 --query count --waves-per-cu $wpc; done
