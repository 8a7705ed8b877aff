#!/bin/bash
# look-ahead rows (four entries per window): long reads, HBM-sized tables, the cap
mkdir -p gpurun_out/r03u
run() { python3 bench.py "$@" --quick --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; l=d['roofline']['launch']
print('$*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'it/base', c.get('iterations_per_base'), 'simt', c.get('simt_efficiency'), 'cap', l['waves_per_cu'], 'staged', l['staged'], 'ahead', l['ahead'])"; }
for ah in 0 1; do
  run --workload c3 --ahead-rows $ah
  run --workload c2synth --rows 200000000 --ahead-rows $ah
  run --workload c4 --ahead-rows $ah
done
for w in 8 9 10 12; do run --workload c2 --waves-per-cu $w; run --workload c2synth --waves-per-cu $w; done
run --workload c2 --read-len 300 --reads 500000
