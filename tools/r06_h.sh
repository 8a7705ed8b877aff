#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_h; mkdir -p $O
( time timeout 2400 python3 -m pytest tests/test_ahead_rows_gpu.py tests/test_big_table_gpu.py tests/test_deep_rows_gpu.py tests/test_device_entry_gpu.py tests/test_mask_gpu.py tests/test_pangenome_gpu.py tests/test_bench_gpu.py -q -m gpu -p no:cacheprovider ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -12 $O/pytest.log | cut -c1-300
