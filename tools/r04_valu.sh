#!/bin/bash
# usage: tools/r04_valu.sh <outdir> "<bench args>"  -- ONE PMC pass (instruction mix) of `bench.py --quick <args>`
OUT=$1; ARGS=$2
cd "$(dirname "$0")/.." || exit 1
export MOVI_BENCH_CACHE=${MOVI_BENCH_CACHE:-$PWD/.bench_cache}
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p "$OUT"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d "$OUT/pmc" -- python3 bench.py --quick --steps 3 --warmup 1 $ARGS > "$OUT/pmc.log" 2>&1
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -h "pml_kernel_flatp\|count_kernel\|zml_kernel" "$OUT/summary.txt" | grep "VALU\|SALU\|INSTS_LDS\|VMEM_RD\|WAVE_CYCLES" | sed 's/void movi:://' | awk '{print $2,$3,$4,$5,$6,$7,$8,$9,$10, $14,$15,$16}' 
grep -o '"value": [0-9.]*\|"iterations_per_base": [0-9.]*\|"simt_efficiency": [0-9.]*' "$OUT/pmc.log" | tr '\n' ' '; echo
