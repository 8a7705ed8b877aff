mkdir -p gpurun_out/r3f
timeout 600 python tools/cpu_threads_sweep.py > gpurun_out/r3f/cpu_sweep.txt 2>&1
cat gpurun_out/r3f/cpu_sweep.txt
