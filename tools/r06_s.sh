#!/bin/bash
# round 6: movi_pml_host, chunk size of the overlapped path for calls that bring masks down (tools/r06_s.py)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_s; mkdir -p $O

timeout 900 python3 tools/r06_s.py 2>&1 | grep -v amdgpu.ids | tee $O/host_threads.txt
