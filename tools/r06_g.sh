#!/bin/bash
# round 6: the vector through fused masks (movi_pml_device default on the deep rows)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python3 -m pytest tests/test_mask_gpu.py tests/test_deep_rows_gpu.py tests/test_pangenome_gpu.py -x -q -k "not parity_files" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
timeout 400 python3 bench.py --quick --workload c2 > $O/c2_default.json 2> $O/c2_default.err
timeout 400 python3 bench.py --quick --workload c2 --opt fused_expand=0 > $O/c2_unfused.json 2> $O/c2_unfused.err
for cap in 9 11 12 14; do timeout 400 python3 bench.py --quick --workload c2 --waves-per-cu $cap > $O/c2_cap$cap.json 2> /dev/null; done
timeout 400 python3 bench.py --quick --workload c2 --opt pml_via_mask=0 > $O/c2_via0.json 2> /dev/null
for via in 0 1; do timeout 400 python3 bench.py --quick --workload c3 --opt pml_via_mask=$via > $O/c3_via$via.json 2> /dev/null; done
for via in 0 1; do timeout 900 python3 bench.py --quick --workload c4 --opt pml_via_mask=$via > $O/c4_via$via.json 2> /dev/null; done
for via in 0 1; do timeout 400 python3 bench.py --quick --workload c2synth --opt pml_via_mask=$via > $O/c2synth_via$via.json 2> /dev/null; done
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    m=d.get("mask_path",{})
    print("%-24s value %.2f kernel_ms %.4f cap %s staged %s | masks %.2f (%.4f ms) expand %.4f ms | %s"%(os.path.basename(f), d["value"] or -1, d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"]["waves_per_cu"], d["roofline"]["launch"]["staged"], m.get("masks_gbases_s",-1), m.get("walk_ms",-1), m.get("expand_ms",-1), d["roofline"]["kernel"]))
PY
