#!/bin/bash
# round 6: final host path -- host-side tests, then rates (masks only | vector through masks | vector by DMA)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_x; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_mask_gpu.py tests/test_device_entry_gpu.py tests/test_gpu_parity.py tests/test_cli_gpu.py -q -m gpu -p no:cacheprovider -x -k "host or overlap or pipel or chunk or mask or pinned or large or cli or zml" 2>&1 | tail -3
python3 tools/r06_s.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/rates_final.txt
