#!/bin/bash
# round 4, final tree: kernel trace + PMC passes (own runs, never combined with tracing) of the default launches of c2, c3 and c4
# usage: tools/r04_final.sh <outdir> [workloads: "c2 c3 c2_count c4"]
OUT=${1:-gpurun_out/r04_final}
WLS=${2:-c2 c3 c2_count c4}
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=${MOVI_BENCH_CACHE:-$PWD/.bench_cache}
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p "$OUT"
one() {  # name, bench args
  n=$1; shift
  d="$OUT/$n"; mkdir -p "$d"
  timeout 900 rocprofv3 --kernel-trace --stats -d "$d/kt" -- python3 bench.py --quick --steps 10 --warmup 2 "$@" > "$d/kt.log" 2>&1
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp -d "$d/pmc$i" -- python3 bench.py --quick --steps 3 --warmup 1 "$@" > "$d/pmc$i.log" 2>&1
  done
  python3 tools/prof_summary.py "$d" > "$d/summary.txt" 2>&1
  find "$d" -name "*.db" -delete
  grep -o '"value": [0-9.]*\|"iterations_per_base": [0-9.]*\|"simt_efficiency": [0-9.]*\|"fast_forwards_per_base": [0-9.]*\|"scans_per_base": [0-9.]*' "$d/kt.log" | tr '\n' ' ' > "$d/bench_line.txt"
}
for w in $WLS; do
  case $w in
    c2_count) one c2_count --workload c2 --query count;;
    c4_plain) one c4_plain --workload c4 --ahead-rows 0;;
    *) one $w --workload $w;;
  esac
done
{ for n in $WLS; do echo "==== $n: bench.py --quick --workload ... under rocprofv3 ($(cat $OUT/$n/bench_line.txt))"; grep -h "KERNEL\|PMC" $OUT/$n/summary.txt | grep "pml_kernel_flatp\|count_kernel\|kmer_table\|ahead_rows\|ftab_kernel" | sed 's/void movi:://' | cut -c1-250; done; } > $OUT/r04_final_kernels.txt
tail -5 $OUT/r04_final_kernels.txt
