#!/bin/bash
# round 6, third GPU pass: deep rows -- parity tests; output path x occupancy cap on c2 (packer / LDS ring / reset masks)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python3 -m pytest tests/test_deep_rows_gpu.py tests/test_mask_gpu.py tests/test_kernel_coverage_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for ring in 0 1; do for cap in 9 10 11 12 13 14 16; do
  timeout 400 python3 bench.py --quick --workload c2 --opt deep_rows=1 --opt out_ring=$ring --waves-per-cu $cap > $O/c2_deep_ring${ring}_cap$cap.json 2> /dev/null
done; done
for cap in 9 11 13; do timeout 400 python3 bench.py --quick --workload c2 --opt deep_rows=0 --opt out_ring=1 --waves-per-cu $cap > $O/c2_ahead_ring1_cap$cap.json 2> /dev/null; done
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    m=d.get("mask_path",{})
    print("%-32s value %.2f kernel_ms %.4f staged %s | masks %.2f (%.4f ms) %s"%(os.path.basename(f), d["value"] or -1, d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"]["staged"], m.get("masks_gbases_s",-1), m.get("walk_ms",-1), d["roofline"]["kernel"]))
PY
