#!/bin/bash
# round 5: the bulk cut's batches in closed form (default) against the running sum over the records (MOVI_NO_CLOSED_CUT=1): 1 M x 150 bp --no-output
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_cut${1:+_$1}; mkdir -p $O
IDX=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
import bench
a = np.fromfile('.bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin', np.uint8).reshape(-1, 150)
bench.write_fasta('/tmp/short.fa', a)
PY
run() { local name=$1 n=$2 rep; shift 2
  for rep in $(seq 1 $n); do movi_amd/bin/movi query -i $IDX --verbose "$@" 2> $O/$name.$rep.err > /dev/null; done
  echo -n "$name: "; grep -h "processing the reads" $O/$name.*.err | awk '{print $8}' | sort -n | tr '\n' ' '; echo
  grep -h "Parser phases" $O/$name.*.err | sed 's/.*batch cut \([0-9.e-]*\) s, lengths \([0-9.e-]*\) s, copy \([0-9.e-]*\) s.*/\1 \2 \3/' | awk '{c += $1; l += $2; k += $3; n++} END {printf "   mean of %d runs: batch cut %.2f, lengths %.2f, copy %.2f ms\n", n, c / n * 1e3, l / n * 1e3, k / n * 1e3}'
}
{
for round in 1 2 3; do
run closed_r$round 12 -r /tmp/short.fa --no-output
MOVI_NO_CLOSED_CUT=1 run sum_r$round 12 -r /tmp/short.fa --no-output
done
} 2>&1 | tee $O/summary.txt
