#!/bin/bash
# look-ahead rows A/B: c2 (pangenome), c2synth (random 10 M rows), 1.6 GB random table, 300 bp reads
mkdir -p gpurun_out/r03r
run() { python3 bench.py "$@" --quick --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'it/base', c.get('iterations_per_base'), 'simt', c.get('simt_efficiency'), d['roofline']['kernel'])"; }
for ah in 0 1; do
  run --workload c2 --ahead-rows $ah
  run --workload c2 --ahead-rows $ah --kmer-k 0
  run --workload c2 --ahead-rows $ah --read-len 300 --reads 500000
  run --workload c2synth --ahead-rows $ah
  run --workload c2synth --rows 200000000 --ahead-rows $ah
done 2>&1 | tee gpurun_out/r03r/ahead_ab.txt
