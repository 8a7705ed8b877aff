#!/bin/bash
# round 6: the command line after the clock change (the input's first scan counted) and with / without the mask paths; the count query on
# the look-ahead rows ("zml_ahead" 1); c2mid on its three layouts
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_k; mkdir -p $O
D=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
import bench
r = np.fromfile(".bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin", np.uint8).reshape(-1, 150)
bench.write_fasta("/tmp/reads150.fa", r)
PY
for via in default 0; do
  for flags in "--no-output" "-o /tmp/out_a"; do
    for rep in 1 2 3 4 5; do
      rm -f /tmp/out_a*.bpf
      if [ $via = default ]; then ./movi_amd/bin/movi query -i $D -r /tmp/reads150.fa --verbose $flags 2> $O/cli.err > /dev/null
      else MOVI_PML_VIA_MASK=0 ./movi_amd/bin/movi query -i $D -r /tmp/reads150.fa --verbose $flags 2> $O/cli.err > /dev/null; fi
      echo "via=$via $flags: $(grep -h 'processing the reads' $O/cli.err | sed 's/.*reads: //')"
    done
  done
done 2>&1 | tee $O/cli.txt
for za in 0 1; do
  timeout 400 python3 bench.py --quick --workload c2 --query count --opt ahead_rows=1 --opt zml_ahead=$za > $O/c2_count_za$za.json 2> /dev/null
  timeout 400 python3 bench.py --quick --workload c2 --query zml --opt ahead_rows=1 --opt zml_ahead=$za > $O/c2_zml_za$za.json 2> /dev/null
done
timeout 900 python3 bench.py --quick --workload c2mid > $O/c2mid_default.json 2> $O/c2mid_default.err
timeout 900 python3 bench.py --quick --workload c2mid --opt deep=0 > $O/c2mid_ahead.json 2> /dev/null
timeout 900 python3 bench.py --quick --workload c2mid --opt ahead_rows=0 > $O/c2mid_plain.json 2> /dev/null
timeout 900 python3 bench.py --quick --workload c2mid --opt pml_via_mask=0 > $O/c2mid_packer.json 2> /dev/null
timeout 900 python3 bench.py --quick --workload c2mid --query count > $O/c2mid_count.json 2> /dev/null
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    print("%-24s value %.2f kernel_ms %.4f cap %s it/base %s simt %s | %s"%(os.path.basename(f), d["value"] or -1, d["roofline"]["kernel_ms_avg"], d["roofline"]["launch"]["waves_per_cu"], d["config"]["iterations_per_base"], d["config"]["simt_efficiency"], d["roofline"]["kernel"]))
PY
