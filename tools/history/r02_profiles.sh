#!/bin/bash
# Round-2 evidence run: kernel traces of the shipped kernels on the bench workloads, then PMC passes for c2 / c3 / c4.
OUT=${1:-gpurun_out/r02_final}
mkdir -p "$OUT"
tools/kt_final.sh "$OUT/kt"
tools/pmc_passes.sh "$OUT/pmc_c2" "--workload c2 --no-long-reads"
tools/pmc_quick.sh "$OUT/pmc_c3" "--workload c3"
tools/pmc_quick.sh "$OUT/pmc_c4" "--workload c4"
