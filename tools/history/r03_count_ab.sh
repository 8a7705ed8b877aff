#!/bin/bash
run() { python3 bench.py "$@" --quick --query count --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['roofline']['launch']
print('$*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'cap', l['waves_per_cu'])"; }
run --workload c2
run --workload c2synth
run --workload c3 --steps 3
run --workload c2synth --rows 200000000
[ "$1" = "c5" ] && run --workload c5 --steps 5
