#!/bin/bash
# round 6: PMC of the final default launches (c2 / c3 / c4 PML; c2 / c4 count and ZML), kernel trace of the few-long-reads segment plan
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_i; mkdir -p $O
bash tools/r06_pmc.sh $O c2 "--workload c2"
bash tools/r06_pmc.sh $O c2_packer "--workload c2 --opt pml_via_mask=0"
bash tools/r06_pmc.sh $O c3 "--workload c3"
bash tools/r06_pmc.sh $O c2_count "--workload c2 --query count"
bash tools/r06_pmc.sh $O c2_zml "--workload c2 --query zml"
bash tools/r06_pmc.sh $O c4 "--workload c4"
bash tools/r06_pmc.sh $O c4_count "--workload c4 --query count"
bash tools/r06_pmc.sh $O c4_zml "--workload c4 --query zml"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_few -- python3 bench.py --quick --workload c3 --reads 25000 --steps 5 --warmup 1 > $O/kt_few.log 2>&1
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1; find $O -name "*.db" -delete
grep -h "KERNEL\|PMC.*movi::\(pml_kernel_flatp\|zml_kernel_flat\|seg_\)" $O/summary.txt | cut -c1-250 > $O/summary_short.txt
grep -h '"value"' $O/kt_*.log | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['config']['workload'], d['config']['query'], 'value %.2f kernel_ms %.4f'%(d['value'], d['roofline']['kernel_ms_avg']), d['roofline']['kernel'], 'it/base', d['config']['iterations_per_base'], 'simt', d['config']['simt_efficiency'], 'matched', d['config']['matched_bases_per_step'])
"
