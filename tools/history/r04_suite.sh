#!/bin/bash
# round 4: the whole GPU suite + smoke on one box
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_suite; mkdir -p $O
( time timeout 5000 python3 -m pytest tests -q -m gpu -x ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
( time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
