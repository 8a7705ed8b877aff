#!/bin/bash
# round 6 (final tree): the slow real-BWT parity tests alone (tools/r06_real_bwt.sh ran the bench lines)
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=/tmp/movi_bench_cache
mkdir -p $MOVI_BENCH_CACHE; cp -rn .bench_cache/* $MOVI_BENCH_CACHE/ 2>/dev/null
O=gpurun_out/r06_real_bwt; mkdir -p $O
( time MOVI_SLOW_TESTS=1 timeout 3000 python3 -m pytest tests/test_real_bwt_gpu.py -q -m gpu -x -k "c4real or c4real2" ) > $O/pytest2.txt 2>&1; tail -6 $O/pytest2.txt
