mkdir -p gpurun_out/r3j
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "segment_parallel" 2>&1 | tail -5
run() { echo "== $1"; timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-long-reads $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%.2f Gbases/s | %.3f ms | f=%.3f s=%.3f | segs=%s rewalked=%s | simt=%s it/base=%s' % (d['value'], r['kernel_ms_avg'], c['fast_forwards_per_base'], c['scans_per_base'], c.get('segments'), c.get('rewalked_reads'), c.get('simt_efficiency'), c.get('iterations_per_base')))"; }
{
for sl in 2048; do
run "c3synth 100k x 10kbp seg $sl" "--workload c3synth --seg-len $sl"
run "c3synth 60k x 10kbp seg $sl" "--workload c3synth --reads 60000 --seg-len $sl"
run "c3synth 60k x 10kbp seg 0" "--workload c3synth --reads 60000 --seg-len 0"
run "c3synth 25k x 10kbp seg $sl" "--workload c3synth --reads 25000 --seg-len $sl"
run "c3synth 5k x 10kbp seg $sl" "--workload c3synth --reads 5000 --seg-len $sl"
run "c3synth 200 x 1Mbp seg $sl" "--workload c3synth --reads 200 --read-len 1000000 --seg-len $sl"
run "c3synth ragged 100k seg $sl" "--workload c3synth --ragged 1 --seg-len $sl"
run "c3synth 1% subst 25k x 10kbp seg $sl" "--workload c3synth --reads 25000 --sub-rate 0.01 --seg-len $sl"
run "c3synth 0.1% subst 25k x 10kbp seg $sl" "--workload c3synth --reads 25000 --sub-rate 0.001 --seg-len $sl"
run "c3synth 0.1% subst 100k x 10kbp seg $sl" "--workload c3synth --sub-rate 0.001 --seg-len $sl"
done
} > gpurun_out/r3j/seg5.txt 2>&1
cat gpurun_out/r3j/seg5.txt
