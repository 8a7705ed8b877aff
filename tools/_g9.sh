mkdir -p gpurun_out/r3j
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
run() { echo "== $1"; timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-long-reads $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%.2f Gbases/s | %.3f ms | f=%.3f s=%.3f | segs=%s rewalked=%s' % (d['value'], r['kernel_ms_avg'], c['fast_forwards_per_base'], c['scans_per_base'], c.get('segments'), c.get('rewalked_reads')))"; }
{
run "pml c3synth 25k x 10kbp" "--workload c3synth --reads 25000"
run "pml c3synth 5k x 10kbp" "--workload c3synth --reads 5000"
run "pml c3synth 200 x 1Mbp" "--workload c3synth --reads 200 --read-len 1000000"
run "pml c3synth 1% 25k" "--workload c3synth --sub-rate 0.01 --reads 25000"
run "pml c3synth 60k x 10kbp" "--workload c3synth --reads 60000"
run "zml c2synth (1 M x 150 bp, unaffected)" "--query zml --workload c2synth"
run "zml c3synth 0.1% 100k (declined)" "--query zml --workload c3synth --sub-rate 0.001"
} > gpurun_out/r3j/final_seg.txt 2>&1
cat gpurun_out/r3j/final_seg.txt
