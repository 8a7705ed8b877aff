#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
bash tools/r05_cli.sh b
timeout 1200 python3 -m pytest tests/test_bench_gpu.py -q -m gpu -x -k "default_line_legs or single_gpu" > gpurun_out/r05_cli_b/pytest_bench.txt 2>&1; tail -5 gpurun_out/r05_cli_b/pytest_bench.txt
