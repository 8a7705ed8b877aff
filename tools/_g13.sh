mkdir -p gpurun_out/r3n
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "segment or overlapped" 2>&1 | tail -4 > gpurun_out/r3n/tests2.txt
run() { echo "== $1"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-long-reads $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%.2f Gbases/s | %.3f ms | f=%.3f s=%.3f | segs=%s rewalked=%s' % (d['value'], r['kernel_ms_avg'], c['fast_forwards_per_base'], c['scans_per_base'], c.get('segments'), c.get('rewalked_reads')))"; }
{
run "c3synth 0.1% 25k seg 2048" "--workload c3synth --reads 25000 --sub-rate 0.001"
run "c3synth 0% 25k seg 2048" "--workload c3synth --reads 25000 --sub-rate 0.0"
run "c3synth 8% 25k seg 2048" "--workload c3synth --reads 25000"
run "c3synth 8% 25k classify 2" "--workload c3synth --reads 25000 --classify 2"
run "c3synth 0.1% 25k classify 2" "--workload c3synth --reads 25000 --sub-rate 0.001 --classify 2"
} > gpurun_out/r3n/perf2.txt 2>&1
cat gpurun_out/r3n/tests2.txt gpurun_out/r3n/perf2.txt
