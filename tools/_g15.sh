mkdir -p gpurun_out/r3p
timeout 900 python tools/host_rate.py > gpurun_out/r3p/host_rate.txt 2>&1
grep -E "==|movi_pml_host|movi_zml_host" gpurun_out/r3p/host_rate.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlapped or default_policy" 2>&1 | tail -2
