#!/bin/bash
# usage: tools/r03_profiles.sh <outdir> <name> "<bench args>"
# rocprofv3 over `bench.py --quick <args>`: one --kernel-trace --stats pass, then one pass per PMC group (counters in their own
# runs, never combined with tracing domains).  Summaries: <outdir>/<name>/summary.txt
OUT=$1/$2; ARGS=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 bench.py --quick --steps 10 --warmup 2 $ARGS > "$OUT/kt.log" 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d "$OUT/pmc$i" -- python3 bench.py --quick --steps 3 --warmup 1 $ARGS > "$OUT/pmc$i.log" 2>&1
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_BRANCH
GROUPS
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
