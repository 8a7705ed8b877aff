#!/bin/bash
# round 6: movi_pml_host -- the way down chosen chunk by chunk, tail cut in halves (A/B on one box); stitch kernel with 16 lanes per wavefront
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_p; mkdir -p $O
timeout 600 python3 tools/r06_n.py 2>&1 | grep -v amdgpu.ids | tee $O/ways_down.txt
for i in 1 2; do timeout 400 python3 bench.py --quick --workload c3 --reads 25000 > $O/few_$i.json 2> /dev/null; done
timeout 400 python3 bench.py --quick --workload c3 --reads 50000 > $O/r50k.json 2> /dev/null
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    print("%-24s value %.2f ms %.4f segs %s rewalked %s | %s"%(os.path.basename(f), d["value"] or -1, d["ms_per_step"], d["config"]["segments"], d["config"]["rewalked_reads"], d["roofline"]["kernel"]))
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_few -- python3 bench.py --quick --steps 5 --warmup 1 --workload c3 --reads 25000 > $O/kt_few.log 2>&1
python3 tools/prof_summary.py $O 2>/dev/null | grep "KERNEL" | head -8 | cut -c1-200
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_device_entry_gpu.py tests/test_mask_gpu.py tests/test_ahead_rows_gpu.py -q -m gpu -k "seg or mask or ways" -p no:cacheprovider -x 2>&1 | tail -3
