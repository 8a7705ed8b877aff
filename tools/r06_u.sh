#!/bin/bash
# round 6: worker threads of the host-side expansion, one process per setting (tools/r06_u.py)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_u; mkdir -p $O
for rep in 1 2 3; do for t in 10 12 14 15 16; do python3 tools/r06_u.py $t 2>&1 | grep host_threads; done; done | tee $O/threads.txt
