#!/bin/bash
# look-ahead rows on: occupancy cap and top-of-walk K re-swept (the balance between lines and iterations moved)
mkdir -p gpurun_out/r03s
run() { python3 bench.py "$@" --quick --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$*', '->', round(d['value'],2), 'Gbases/s', round(d['ms_per_step'],3),'ms', 'it/base', c.get('iterations_per_base'), 'simt', c.get('simt_efficiency'), d['roofline']['launch']['staged'])"; }
for w in 5 6 7 8 9 10 12; do run --workload c2 --waves-per-cu $w; done
for k in 8 10 11; do run --workload c2 --kmer-k $k; done
for w in 6 7 8 9; do run --workload c2synth --waves-per-cu $w; done
