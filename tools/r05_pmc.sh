#!/bin/bash
# round 5: kernel trace + PMC passes (own runs, never combined with tracing) of one bench.py launch per entry
# usage: tools/r05_pmc.sh <outdir> <groups: all|mem|few> "<name>:<bench args>" ...
OUT=$1; GR=$2; shift 2
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=${MOVI_BENCH_CACHE:-$PWD/.bench_cache}
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p "$OUT"
G_ALL=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS")
G_MEM=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum")
G_FEW=("TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS")
case $GR in all) GS=("${G_ALL[@]}");; mem) GS=("${G_MEM[@]}");; *) GS=("${G_FEW[@]}");; esac
names=""
for ent in "$@"; do
  n=${ent%%:*}; a=${ent#*:}; names="$names $n"
  d="$OUT/$n"; mkdir -p "$d"
  timeout 1200 rocprofv3 --kernel-trace --stats -d "$d/kt" -- python3 bench.py --quick --steps 10 --warmup 2 $a > "$d/kt.log" 2>&1
  i=0
  for grp in "${GS[@]}"; do
    i=$((i+1))
    timeout 1200 rocprofv3 --pmc $grp -d "$d/pmc$i" -- python3 bench.py --quick --steps 3 --warmup 1 $a > "$d/pmc$i.log" 2>&1
  done
  python3 tools/prof_summary.py "$d" > "$d/summary.txt" 2>&1
  find "$d" -name "*.db" -delete
  grep -o '"value": [0-9.]*\|"iterations_per_base": [0-9.]*\|"simt_efficiency": [0-9.]*\|"fast_forwards_per_base": [0-9.]*\|"scans_per_base": [0-9.]*\|"matched_bases_per_step": [0-9.]*' "$d/kt.log" | tr '\n' ' ' > "$d/bench_line.txt"
done
{ for n in $names; do echo "==== $n: bench.py --quick ... under rocprofv3 ($(cat $OUT/$n/bench_line.txt))"; grep -h "KERNEL\|PMC" $OUT/$n/summary.txt | grep "pml_kernel\|count_kernel\|zml_kernel\|kmer_table\|ahead_rows\|ftab_kernel\|seg_\|expand_" | sed 's/void movi:://' | cut -c1-250; done; } > $OUT/kernels.txt
tail -4 $OUT/kernels.txt
