#!/bin/bash
# round 6: `movi query` with the engine's host-side mask expansion off (the default of the command now) against on; option split checked
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_m; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_mask_gpu.py tests/test_deep_rows_gpu.py tests/test_device_entry_gpu.py tests/test_kernel_coverage_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
D=.bench_cache/pg_5000000_64_0.001_11_m6
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
import bench
r = np.fromfile(".bench_cache/pg_5000000_64_0.001_11_m6/reads_1000000x150_0.01.bin", np.uint8).reshape(-1, 150)
bench.write_fasta("/tmp/reads150.fa", r)
PY
for hm in 0 -1 0 -1; do
  for rep in 1 2 3 4 5; do
    rm -f /tmp/out_a*.bpf
    MOVI_HOST_MASKS=$hm ./movi_amd/bin/movi query -i $D -r /tmp/reads150.fa --verbose -o /tmp/out_a 2> $O/cli.err > /dev/null
    echo "host_masks $hm: $(grep -h 'processing the reads' $O/cli.err | sed 's/.*reads: //')"
  done
done 2>&1 | tee $O/cli.txt
grep -h "Stage times\|BPF writer" $O/cli.err
