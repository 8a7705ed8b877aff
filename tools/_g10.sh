mkdir -p gpurun_out/r3k
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3k/tests.txt
timeout 900 python tools/host_rate.py > gpurun_out/r3k/host_rate.txt 2>&1
python bench.py > gpurun_out/r3k/bench.json 2> gpurun_out/r3k/bench.err
cat gpurun_out/r3k/tests.txt; grep -E "==|movi_pml_host" gpurun_out/r3k/host_rate.txt; tail -2 gpurun_out/r3k/bench.err; cat gpurun_out/r3k/bench.json
