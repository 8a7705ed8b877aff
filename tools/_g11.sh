mkdir -p gpurun_out/r3l
python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3l/cli_tests.txt
bash tools/cli_ragged.sh > gpurun_out/r3l/cli_ragged.txt 2>&1
cat gpurun_out/r3l/cli_tests.txt gpurun_out/r3l/cli_ragged.txt
