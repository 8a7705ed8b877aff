#!/bin/bash
# round 5: which regime is c3 in?  throughput against the number of reads in flight (N), segments forced, and PMC with / without the hints
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r05_c3_regime; mkdir -p $O
run() { n=$1; shift
timeout 900 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-14s %.2f Gb/s ms %.3f iter/base %s simt %s wpc %s segs %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c.get("iterations_per_base"),c.get("simt_efficiency"),d["roofline"]["launch"]["waves_per_cu"],c.get("segments"),d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for n in 25000 50000 75000 100000 150000 200000 300000; do run c3_n$n --workload c3 --reads $n --steps 5 --seg-len 0; done
run c3_seg5k --workload c3 --steps 5 --opt seg_probe=0 --seg-len 4992
run c3_seg2k --workload c3 --steps 5 --opt seg_probe=0 --seg-len 2048
} 2>&1 | tee $O/summary.txt
bash tools/r05_pmc.sh $O/pmc all "c3_hints:--workload c3" "c3_nohints:--workload c3 --opt repo_hints=0" "c3_seg5k:--workload c3 --opt seg_probe=0 --seg-len 4992"
cp $O/pmc/kernels.txt $O/pmc_kernels.txt
