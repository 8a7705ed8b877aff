#!/bin/bash
# round 4: the parity suites with pair-shared gathers forced on every handle (MOVI_PAIR_LOADS=1): the PSH instantiations of the
# segment kernels and of the ring kernels run on tables that would not select them by size.  Kernel-name / launch-shape assertions
# are expected to fail; answers must not.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_suite_pairs; mkdir -p $O
MOVI_PAIR_LOADS=1 timeout 3000 python3 -m pytest tests/test_gpu_parity.py tests/test_ahead_rows_gpu.py tests/test_top_of_walk_gpu.py tests/test_device_entry_gpu.py tests/test_cli_gpu.py tests/test_pangenome_gpu.py -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
grep -n "^E  " $O/pytest.txt | cut -c1-200 | head -40 > $O/errors.txt
