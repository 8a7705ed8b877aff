#!/bin/bash
# round 5: the whole GPU suite + smoke, then the count query's occupancy-cap sweep and the parser alone on the box
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r05_suite2${1:+_$1}; mkdir -p $O
( time timeout 3000 python3 -m pytest tests -q -m gpu -x ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
( time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
run() { n=$1; shift
timeout 1500 python3 bench.py --quick "$@" > $O/$n.json 2>$O/err_$n.txt
python3 - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-22s %.2f Gb/s ms %.3f rows %d prepare_s %s wpc %s kernel %s"%(sys.argv[2],d["value"],d["ms_per_step"],c["rows"],c.get("prepare_s"),d["roofline"]["launch"]["waves_per_cu"],d["roofline"]["kernel"]))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
{
for w in 0 8 12 16 24; do run c2_count_w$w --workload c2 --query count --steps 10 --waves-per-cu $w; done
run c2_count_pair --workload c2 --query count --steps 10 --opt pair_loads=1
run c4_200M_count_pair --workload c4 --rows 200000000 --query count --steps 5 --opt pair_loads=1
run c4_200M_count_w16 --workload c4 --rows 200000000 --query count --steps 5 --waves-per-cu 16
run c2_pml --workload c2 --steps 20
run c3_pml --workload c3 --steps 5
} 2>&1 | tee $O/summary.txt
g++ -O2 -std=c++17 -mavx2 -o /tmp/parse_bench tools/parse_bench.cpp movi_amd/host/reads.cpp -lpthread 2> $O/parse_build.txt
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/reads_1M.fa', a)
PY
/tmp/parse_bench /tmp/reads_1M.fa 4 8 12 16 > $O/parse_bench.txt 2>&1; tail -6 $O/parse_bench.txt
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
