#!/bin/bash
# usage: tools/r03_count_zml_sweep.sh <outdir> -- occupancy cap / block size of the count kernel, second pass
OUT=$1; mkdir -p "$OUT"
run() { name=$1; shift; python3 bench.py --quick --steps 20 --warmup 3 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err";
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s %8.3f Gbases/s  kernel %8.3f ms  bt %s wpc %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"]["block_threads"], d["roofline"]["launch"]["waves_per_cu"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for wpc in 14 15 16 17 18 20; do run c2_count_bt64_wpc$wpc --workload c2 --query count --block-threads 64 --waves-per-cu $wpc; done
for wpc in 0 16 20 24 28; do run c5_count_bt64_wpc$wpc --workload c5 --query count --block-threads 64 --waves-per-cu $wpc; done
for wpc in 0 14 16 18 20 24; do run r200M_count_bt64_wpc$wpc --workload c4 --rows 200000000 --query count --block-threads 64 --waves-per-cu $wpc; done
for wpc in 0 14 16 18; do run c2synth_count_bt64_wpc$wpc --workload c2synth --query count --block-threads 64 --waves-per-cu $wpc; done
for wpc in 0 16; do run c3_count_bt64_wpc$wpc --workload c3 --query count --block-threads 64 --waves-per-cu $wpc; done
