#!/bin/bash
# table-footprint sweep of the default PML kernel on random tables (Infinity Cache = 256 MiB): where does the cliff start?
mkdir -p gpurun_out/r03q
for rows in 10000000 14000000 20000000 28000000 34000000 48000000 80000000; do
  python3 bench.py --workload c2synth --rows $rows --quick --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($rows, $rows*8//1000000, 'MB', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms')"
done | tee gpurun_out/r03q/footprint.txt
