#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_x; mkdir -p $O
for rep in 1 2 3 4 5 6; do python3 tools/r06_u.py 12 2>&1 | grep host_threads; done | tee $O/rates_upfront_priority.txt
python3 tools/r06_s.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/rates_upfront_priority.txt
