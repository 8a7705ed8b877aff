#!/bin/bash
# round 5, final tree: the whole GPU suite + smoke, then tools/r05_final_b.sh (default bench line, two ranks on one GPU, kernel trace)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_final_c; mkdir -p $O
( time timeout 3000 python3 -m pytest tests -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
( time python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
bash tools/r05_final_b.sh
