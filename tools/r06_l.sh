#!/bin/bash
# round 6: few long reads (25 k x 10 kbp) -- segment length; the command line with the input read inside the clock; zml coverage test
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_l; mkdir -p $O
timeout 900 python3 -m pytest tests/test_zml_coverage_gpu.py tests/test_cli_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
for sl in 2048 1536 1024 768 512; do
  timeout 400 python3 bench.py --quick --workload c3 --reads 25000 --seg-len $sl > $O/few_seg$sl.json 2> /dev/null
done
timeout 400 python3 bench.py --quick --workload c3 --reads 25000 --seg-len 1024 --opt deep=1 > $O/few_seg1024_deep.json 2> /dev/null
timeout 400 python3 bench.py --quick --workload c3 --reads 50000 --seg-len 1024 > $O/r50k_seg1024.json 2> /dev/null
timeout 400 python3 bench.py --quick --workload c3 --reads 50000 > $O/r50k_seg2048.json 2> /dev/null
D=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
import bench
r = np.fromfile(".bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin", np.uint8).reshape(-1, 150)
bench.write_fasta("/tmp/reads150.fa", r)
PY
for flags in "--no-output" "-o /tmp/out_a"; do
  for rep in 1 2 3 4 5; do
    rm -f /tmp/out_a*.bpf
    ./movi_amd/bin/movi query -i $D -r /tmp/reads150.fa --verbose $flags 2> $O/cli.err > /dev/null
    echo "$flags: $(grep -h 'processing the reads' $O/cli.err | sed 's/.*reads: //')"
  done
done 2>&1 | tee $O/cli.txt
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
