#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_suite; mkdir -p $O
( time timeout 5000 python3 -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
