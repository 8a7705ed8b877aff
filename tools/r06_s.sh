#!/bin/bash
# round 6: movi_pml_host, chunk size of the overlapped path for calls that bring masks down (tools/r06_s.py)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_s; mkdir -p $O
timeout 900 python3 -m pytest tests/test_mask_gpu.py tests/test_device_entry_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
timeout 900 python3 tools/r06_s.py 2>&1 | grep -v amdgpu.ids | tee $O/host_func_equal_chunks.txt
