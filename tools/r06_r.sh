#!/bin/bash
# round 6: zml_kernel_flat's two-ended advance as predicated straight-line code -- parity, then ZML / count on c2 and on the 1 B-row table
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_r; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_zml_coverage_gpu.py tests/test_gpu_parity.py tests/test_cli_gpu.py -q -m gpu -p no:cacheprovider -x -k "zml or count or coverage" 2>&1 | tail -3
for q in zml count; do
  for wl in c2 c4; do
    timeout 900 python3 bench.py --quick --workload $wl --query $q > $O/${wl}_$q.json 2> /dev/null
    python3 - $O/${wl}_$q.json <<'PY'
import json,sys,os
f=sys.argv[1]
try:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-16s value %.2f ms %.4f kernel %s iter/base %s simt %s"%(os.path.basename(f), d["value"] or -1, d["ms_per_step"], d["roofline"]["kernel"], d["config"].get("iterations_per_base"), d["config"].get("simt_efficiency")))
except Exception as e:
    print(os.path.basename(f),"unreadable",e)
PY
  done
done 2>&1 | tee $O/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $O/pmc_zml -- python3 bench.py --quick --steps 2 --warmup 1 --workload c2 --query zml > $O/pmc_zml.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $O/pmc_count -- python3 bench.py --quick --steps 2 --warmup 1 --workload c2 --query count > $O/pmc_count.log 2>&1
python3 tools/prof_summary.py $O 2>/dev/null | grep "PMC.*zml_kernel_flat" | cut -c1-70,100-200
