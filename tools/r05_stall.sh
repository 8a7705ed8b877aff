#!/bin/bash
# round 5: the one GPU call in ~15 of `movi query` that takes 20 - 30 ms instead of 3: which step of the synchronous host call it is
# (MOVI_TRACE_HOST_CALLS=1: the engine prints every host call's steps to stderr)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_stall${1:+_$1}; mkdir -p $O
IDX=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/short.fa', a)
PY
MODE=${2:-noout}
for rep in $(seq 1 25); do
  if [ $MODE = bpf ]; then rm -f /tmp/stall_out*; MOVI_TRACE_HOST_CALLS=1 movi_amd/bin/movi query -i $IDX --verbose -r /tmp/short.fa -o /tmp/stall_out 2> $O/run.$rep.err > /dev/null
  else MOVI_TRACE_HOST_CALLS=1 movi_amd/bin/movi query -i $IDX --verbose -r /tmp/short.fa --no-output 2> $O/run.$rep.err > /dev/null; fi
done
grep -h "processing the reads" $O/run.*.err | awk '{print $8}' | sort -n | tr '\n' ' ' > $O/summary.txt; echo >> $O/summary.txt
grep -h "host call: 2237" $O/run.*.err | awk '{ up += $15; down += $32; n++ } END { printf "per full chunk: bases up %.3f ms, results down %.3f ms (mean of %d calls)\n", up / n, down / n, n }' >> $O/summary.txt
for f in $O/run.*.err; do
  if awk '/host call/ { for (i = 1; i <= NF; i++) if ($i ~ /^[0-9.]+$/ && $i + 0 > 8 && $(i+1) != "reads" && $(i+1) != "bases") slow = 1 } END { exit !slow }' $f; then echo "== $f"; grep -h "host call\|processing" $f; fi
done >> $O/summary.txt
cat $O/summary.txt | cut -c1-330
