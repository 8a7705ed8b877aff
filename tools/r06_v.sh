#!/bin/bash
# round 6: all the reads of an overlapped host call go up ahead of the loop -- tests, rates, timeline
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_v; mkdir -p $O

for rep in 1 2; do python3 tools/r06_u.py 12 2>&1 | grep host_threads; done | tee $O/rates.txt
bash tools/r06_t.sh > $O/trace.log 2>&1; grep "mask \|vector_via" $O/trace.log | tail -8
python3 tools/r06_t_timeline.py | tail -6
