#!/bin/bash
# rolling staged stretch + look-ahead rows: short reads, long reads, mid-size batches, the 4-entry variant
mkdir -p gpurun_out/r03t
run() { python3 bench.py "$@" --quick --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; l=d['roofline']['launch']
print('$MOVI_AHD2 $*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'it/base', c.get('iterations_per_base'), 'simt', c.get('simt_efficiency'), 'cap', l['waves_per_cu'], 'staged', l['staged'], 'ahead', l['ahead'])"; }
run --workload c2
export MOVI_AHD2=1; run --workload c2; run --workload c2synth; unset MOVI_AHD2
run --workload c2synth
for ah in 0 1; do
  run --workload c3 --ahead-rows $ah
  run --workload c3 --ahead-rows $ah --stage-reads 0
done
for n in 100000 200000 280000; do for sr in 0 1; do run --workload c2 --reads $n --stage-reads $sr; done; done
run --workload c2 --ragged 1
run --workload c2 --ragged 1 --stage-reads 0
