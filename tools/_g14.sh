mkdir -p gpurun_out/r3o
python bench.py > gpurun_out/r3o/bench.json 2> gpurun_out/r3o/bench.err
tail -2 gpurun_out/r3o/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r3o/bench.json')); print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['long_reads'], d.get('few_long_reads'), d.get('host_path'))"
python tests/test_bench_gpu.py 2>/dev/null | tail -1
python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -2
