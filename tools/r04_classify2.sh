#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=$PWD/.bench_cache
O=gpurun_out/r04_classify; mkdir -p $O
for a in "--opt classify_fused=1" "--opt classify_fused=0" ""; do timeout 900 python3 bench.py --quick --workload c3 --steps 5 --classify 1 $a | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 classify 1 [$a]: %.2f Gb/s ms %.3f'%(d['value'],d['ms_per_step']))"; done 2>&1 | tee $O/summary2.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_pangenome_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "classif" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
