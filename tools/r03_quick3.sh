#!/bin/bash
# c2 / c2synth / c3 / c4 in one go (A/B of a kernel change against the committed numbers)
run() { python3 bench.py "$@" --quick --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; l=d['roofline']['launch']
print('$*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'it/base', c.get('iterations_per_base'), 'cap', l['waves_per_cu'], 'staged', l['staged'], 'ahead', l['ahead'])"; }
run --workload c2
run --workload c2synth
run --workload c3 --steps 5
run --workload c2synth --rows 200000000
[ "$1" = "c4" ] && run --workload c4 --steps 10
