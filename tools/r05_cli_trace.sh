#!/bin/bash
# round 5: `movi query` itself under rocprofv3 --kernel-trace --stats (the binary right behind `--`): what the GPU does inside the command's
# 13 - 31 ms of read processing -- the walk kernel's durations per chunk against the command's own clock.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_cli_trace; mkdir -p $O
IDX=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/short.fa', a)
PY
rm -f /tmp/trace_out*
timeout 600 rocprofv3 --kernel-trace --stats -d $O/noout -- movi_amd/bin/movi query -i $IDX -r /tmp/short.fa --no-output --verbose > $O/noout.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/bpf -- movi_amd/bin/movi query -i $IDX -r /tmp/short.fa -o /tmp/trace_out --verbose > $O/bpf.log 2>&1
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
for f in $O/noout.log $O/bpf.log; do echo "== $f"; grep "processing the reads\|Stage times\|loading the index" $f; done >> $O/summary.txt
find $O -name "*.db" -delete
cat $O/summary.txt | cut -c1-220
