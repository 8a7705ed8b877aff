mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlapped or invariant or ragged" 2>&1 | tail -8 > gpurun_out/r3b/tests.txt
timeout 900 python tools/host_rate.py > gpurun_out/r3b/host_rate.txt 2>&1
cat gpurun_out/r3b/tests.txt gpurun_out/r3b/host_rate.txt
