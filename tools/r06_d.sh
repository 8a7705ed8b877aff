#!/bin/bash
# round 6, fourth GPU pass: expand kernel v3; host path thread counts
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python3 -m pytest tests/test_mask_gpu.py tests/test_deep_rows_gpu.py -x -q -k "not parity_files" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
timeout 400 python3 bench.py --quick --workload c2 --waves-per-cu 13 > $O/c2_default_cap13.json 2> $O/c2_default_cap13.err
timeout 400 python3 bench.py --quick --workload c2 > $O/c2_default.json 2> $O/c2_default.err
for th in 6 8 12 16 24; do
  ( timeout 900 python3 bench.py --no-big-table --no-long-reads --opt host_threads=$th ) > $O/bench_c2_th$th.json 2> $O/bench_c2_th$th.err
done
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    m=d.get("mask_path",{})
    print("%-28s value %.2f kernel_ms %.4f | masks %.2f (%.4f ms) expand %.4f ms = %.0f GB/s -> %.2f | %s"%(os.path.basename(f), d["value"] or -1, d["roofline"]["kernel_ms_avg"], m.get("masks_gbases_s",-1), m.get("walk_ms",-1), m.get("expand_ms",-1), m.get("expand_write_gb_s",-1), m.get("masks_plus_expand_gbases_s",-1), d["roofline"]["kernel"]))
    if "host_path" in d: print("    host", json.dumps({a:b for a,b in d["host_path"].items() if a!="note" and not a.endswith("_ok")}))
PY
