#!/bin/bash
# usage: tools/sweep_zml.sh "<bench args>" "<zml variants>" "<waves_per_cu list>"
for v in $2; do for w in $3; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --query zml $1 --zml-variant $v --waves-per-cu $w 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('rows=%d reads=%d len=%d wpc=%s zml_variant=%s : %.2f Gbases/s  kern %.3f ms  ff=%.3f sc=%.3f simt=%s it/base=%s' % (c['rows'],c['reads_per_gpu'],c['read_len'],c['waves_per_cu'],'$v',d['value'],d['roofline']['kernel_ms_avg'],c['fast_forwards_per_base'],c['scans_per_base'],c.get('simt_efficiency'),c.get('iterations_per_base')))"
done; done
