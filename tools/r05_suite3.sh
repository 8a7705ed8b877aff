#!/bin/bash
# round 5: the whole GPU suite + smoke, then the slow real-BWT tests and their bench lines
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r05_suite3${1:+_$1}; mkdir -p $O
( time timeout 3000 python3 -m pytest tests -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
( time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
[ "$2" = "real" ] && bash tools/r05_real_bwt.sh
