mkdir -p gpurun_out/r3n
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "segment_parallel_zml or zml" 2>&1 | tail -25 > gpurun_out/r3n/tests4.txt
cat gpurun_out/r3n/tests4.txt
