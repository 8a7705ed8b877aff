mkdir -p gpurun_out/r3i
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "segment_parallel or fused_classification or classif" 2>&1 | tail -25 > gpurun_out/r3i/tests.txt
cat gpurun_out/r3i/tests.txt
