#!/bin/bash
mkdir -p gpurun_out/r03v
run() { python3 bench.py "$@" --quick --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; l=d['roofline']['launch']
print('$*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'it/base', c.get('iterations_per_base'), 'simt', c.get('simt_efficiency'), 'cap', l['waves_per_cu'], 'staged', l['staged'], 'ahead', l['ahead'])"; }
for rows in 20000000 30000000 45000000 60000000 100000000; do for ah in 0 1; do run --workload c2synth --rows $rows --ahead-rows $ah; done; done 2>&1 | tee gpurun_out/r03v/threshold.txt
bash tools/r03_profiles.sh gpurun_out/r03v c2 "--workload c2"
bash tools/r03_profiles.sh gpurun_out/r03v c3 "--workload c3"
