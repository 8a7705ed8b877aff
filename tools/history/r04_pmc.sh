#!/bin/bash
# usage: tools/r04_pmc.sh <outdir> "<bench args>"   -- kernel trace + 5 PMC passes of `bench.py --quick <args>`
OUT=$1; ARGS=$2
cd "$(dirname "$0")/../.." || exit 1
export MOVI_BENCH_CACHE=${MOVI_BENCH_CACHE:-$PWD/.bench_cache}
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 bench.py --quick --steps 10 --warmup 2 $ARGS > "$OUT/kt.log" 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc$i" -- python3 bench.py --quick --steps 3 --warmup 1 $ARGS > "$OUT/pmc$i.log" 2>&1
done <<'GROUPS'
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum
GROUPS
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -h "pml_kernel_flatp\|count_kernel\|zml_kernel" "$OUT/summary.txt" | grep -v "kmer\|ahead_rows\|chain_rows" | sed 's/void movi:://' | cut -c1-260
