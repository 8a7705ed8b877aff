#!/bin/bash
# round 6: expand tile kernel
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python3 -m pytest tests/test_mask_gpu.py -x -q -k "not parity_files" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
timeout 400 python3 bench.py --quick --workload c2 > $O/c2_default.json 2> $O/c2_default.err
timeout 400 python3 bench.py --quick --workload c2 --opt pml_via_mask=1 > $O/c2_via1.json 2> $O/c2_via1.err
timeout 400 python3 bench.py --quick --workload c2 --opt pml_via_mask=1 --waves-per-cu 11 > $O/c2_via1_cap11.json 2> $O/c2_via1.err
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    m=d.get("mask_path",{})
    print("%-28s value %.2f kernel_ms %.4f cap %s | masks %.2f (%.4f ms) expand %.4f ms = %.0f GB/s -> %.2f | %s"%(os.path.basename(f), d["value"] or -1, d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"]["waves_per_cu"], m.get("masks_gbases_s",-1), m.get("walk_ms",-1), m.get("expand_ms",-1), m.get("expand_write_gb_s",-1), m.get("masks_plus_expand_gbases_s",-1), d["roofline"]["kernel"]))
PY
