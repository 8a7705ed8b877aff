#!/bin/bash
# round 6: movi_pml_host, both ways down, the call's tail cut in halves
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_o; mkdir -p $O
timeout 600 python3 tools/r06_n.py 2>&1 | grep -v amdgpu.ids | tee $O/share_sweep_tapered.txt
timeout 900 python3 -m pytest tests/test_mask_gpu.py tests/test_device_entry_gpu.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
