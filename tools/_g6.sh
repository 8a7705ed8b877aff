mkdir -p gpurun_out/r3g
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3g/tests.txt
python bench.py > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err
cat gpurun_out/r3g/tests.txt; tail -3 gpurun_out/r3g/bench.err; cat gpurun_out/r3g/bench.json
