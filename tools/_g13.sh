mkdir -p gpurun_out/r3n
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz_small or separators or segment" 2>&1 | tail -25 > gpurun_out/r3n/tests3.txt
cat gpurun_out/r3n/tests3.txt
