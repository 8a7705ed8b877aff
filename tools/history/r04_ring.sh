#!/bin/bash
# round 4: PMLs out through a ring in LDS instead of the register packer -- A/B against the library before (.ref_prering/)
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_ring; mkdir -p $O
run() { n=$1; lib=$2; shift; shift
MOVI_HIP_LIB=$lib timeout 1200 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f iter/base %s simt %s wpc %s staged %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["iterations_per_base"],c["simt_efficiency"],d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["launch"].get("staged")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for v in base ring; do
  lib=$PWD/.ref_prering/libmovi_hip.so; [ $v = ring ] && lib=$PWD/movi_amd/lib/libmovi_hip.so
  run c2_$v $lib --workload c2 --steps 20
  run c3_$v $lib --workload c3 --steps 5
  run c4_$v $lib --workload c4 --steps 10
  run c4real_$v $lib --workload c4real --steps 10
  run c2_150k_$v $lib --workload c2 --reads 150000 --steps 20
  run c2_280k_$v $lib --workload c2 --reads 280000 --steps 20
  run c2_cls1_$v $lib --workload c2 --classify 1 --steps 20
done
} 2>&1 | tee $O/summary.txt
timeout 2400 python3 -m pytest tests/test_ahead_rows_gpu.py tests/test_top_of_walk_gpu.py tests/test_device_entry_gpu.py tests/test_pangenome_gpu.py tests/test_gpu_parity.py -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
