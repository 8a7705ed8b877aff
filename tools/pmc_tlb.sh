#!/bin/bash
# usage: tools/pmc_tlb.sh <outdir> <log2 rows>  -- PMC passes over tools/tlb_bench (hipMalloc table only)
OUT=$1; LG=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
export TLB_ALLOC=0
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d "$OUT/pmc$i" -- ./tools/tlb_bench $LG > "$OUT/pmc$i.log" 2>&1
done
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
