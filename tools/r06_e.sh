#!/bin/bash
# round 6, fifth GPU pass: expand kernel v4, ZML / count enumeration test, deep policy (cap 13) on c2 / c3 defaults
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python3 -m pytest tests/test_zml_coverage_gpu.py tests/test_mask_gpu.py tests/test_deep_rows_gpu.py -x -q -k "not parity_files" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for wl in c2 c3; do timeout 400 python3 bench.py --quick --workload $wl > $O/${wl}_default.json 2> $O/${wl}_default.err; done
timeout 400 python3 bench.py --quick --workload c2 --opt pml_via_mask=1 > $O/c2_via1.json 2> $O/c2_via1.err
( timeout 900 python3 bench.py --no-big-table --no-long-reads ) > $O/bench_c2.json 2> $O/bench_c2.err
( timeout 900 python3 bench.py --no-big-table --no-long-reads ) > $O/bench_c2_again.json 2> $O/bench_c2_again.err
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
    if "host_path" in d: print("    host", json.dumps({a:b for a,b in d["host_path"].items() if a!="note" and not a.endswith("_ok")}))
    if "parity_sample_ok" in d: print("    parity", d["parity_sample_ok"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
