#!/bin/bash
# usage: tools/sweep.sh "<bench args>" "<list of variants>" "<list of waves_per_cu>" "<list of block_threads>"
for v in $2; do for bt in $4; do for w in $3; do
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $1 --variant $v --waves-per-cu $w --block-threads $bt 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('rows=%d reads=%d len=%d bt=%s wpc=%s var=%s : %.2f Gbases/s  kern %.3f ms  ff=%.3f sc=%.3f simt=%s it/base=%s' % (c['rows'],c['reads_per_gpu'],c['read_len'],c['block_threads'],c['waves_per_cu'],c['pml_variant'],d['value'],d['roofline']['kernel_ms_avg'],c['fast_forwards_per_base'],c['scans_per_base'],c.get('simt_efficiency'),c.get('iterations_per_base')))"
done; done; done
