#!/bin/bash
# usage: tools/pmc_quick.sh <outdir> "<bench args>"   -- kernel trace + 4 PMC passes (for slow-to-set-up workloads)
OUT=$1; ARGS=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline $ARGS > "$OUT/kt.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d "$OUT/pmc$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $ARGS > "$OUT/pmc$i.log" 2>&1
done
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
