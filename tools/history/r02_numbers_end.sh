#!/bin/bash
# End-of-round-2 result table: one bench.py line per workload (the 1 B-row workloads but c4 PML left out: minutes each).
OUT=${1:-gpurun_out/r02_numbers_end.txt}
mkdir -p "$(dirname "$OUT")"
run() { echo "== $1"; timeout 900 python bench.py --steps 10 --warmup 2 $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%s | %s | rows=%d reads=%d len=%d | f=%.2f s=%.2f B=%.1f | %.2f Gbases/s | %.3f ms | frac=%.4f | segs=%s | cpu=%s | long=%s | few=%s | host=%s' % (d['metric'], r['kernel'], c['rows'], c['reads_per_gpu'], c['read_len'], c['fast_forwards_per_base'], c['scans_per_base'], c['algorithmic_bytes_per_base'], d['value'], r['kernel_ms_avg'], r['frac'], c.get('segments'), (d.get('cpu_baseline') or {}).get('value'), (d.get('long_reads') or {}).get('value'), (d.get('few_long_reads') or {}).get('segment_parallel'), (d.get('host_path') or {}).get('page_locked')))"; }
{
run "c2 default" ""
run "c2 classify 1" "--classify 1 --no-cpu-baseline --no-long-reads"
run "c2 classify 2" "--classify 2 --no-cpu-baseline --no-long-reads"
run "c2 count" "--query count --no-cpu-baseline"
run "c2 zml" "--query zml --no-cpu-baseline"
run "c2b" "--workload c2b --no-cpu-baseline"
run "c2s" "--workload c2s --no-cpu-baseline"
run "c3" "--workload c3 --no-cpu-baseline --steps 5"
run "c3 classify 1" "--workload c3 --classify 1 --no-cpu-baseline --steps 5"
run "c3 classify 2" "--workload c3 --classify 2 --no-cpu-baseline --steps 5"
run "c3 zml" "--workload c3 --query zml --no-cpu-baseline --steps 3"
run "c3 zml uncut" "--workload c3 --query zml --no-cpu-baseline --steps 3 --seg-len 0"
run "c2synth" "--workload c2synth --no-cpu-baseline"
run "c3synth" "--workload c3synth --no-cpu-baseline --steps 5"
run "c3synth 25k reads" "--workload c3synth --reads 25000 --no-cpu-baseline --steps 5"
run "c3synth 25k reads uncut" "--workload c3synth --reads 25000 --no-cpu-baseline --steps 5 --seg-len 0"
run "c3synth ragged" "--workload c3synth --ragged 1 --no-cpu-baseline --steps 5"
run "200M rows" "--workload c2synth --rows 200000000 --no-cpu-baseline"
} > "$OUT" 2>&1
cat "$OUT"
