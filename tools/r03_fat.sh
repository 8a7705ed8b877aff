#!/bin/bash
mkdir -p gpurun_out/r04h
run() { python3 bench.py "$@" --quick --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; l=d['roofline']['launch']
print('$*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'it/base', c.get('iterations_per_base'), 'cap', l['waves_per_cu'], 'staged', l['staged'], 'ahead', l['ahead'])"; }
for ah in 0 2; do
  run --workload c4 --ahead-rows $ah
  run --workload c2synth --rows 200000000 --ahead-rows $ah
done
run --workload c4 --ahead-rows 2 --waves-per-cu 9
run --workload c4 --ahead-rows 2 --waves-per-cu 6
run --workload c2synth --rows 100000000 --ahead-rows 2
run --workload c2synth --rows 100000000 --ahead-rows 1
run --workload c2 --ahead-rows 2
run --workload c2synth --ahead-rows 2
