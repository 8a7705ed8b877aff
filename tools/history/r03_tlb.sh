#!/bin/bash
# usage: tools/r03_tlb.sh <outdir>  -- round 3: which level of address translation bounds the random gather on an 8 GB table?
# (a) driver parameters that set the page-table fragment size; (b) tools/tlb_bench on 2^24 / 2^30 rows, every load shape;
# (c) PMC passes (separate runs, no tracing domains) over the 2^30-row table for the 8-byte, 16-byte and split-load gathers.
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
{
  for f in vm_fragment_size vm_block_size vm_size vm_update_mode noretry mes; do
    printf "%s = " $f; cat /sys/module/amdgpu/parameters/$f 2>&1
  done
  uname -r
  cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E "simd_count|cu_count|gfx_target|local_mem" | head -20
} > "$OUT/driver_params.txt" 2>&1
export TLB_ALLOC=0
timeout 300 ./tools/tlb_bench 24 30 > "$OUT/tlb_bench.txt" 2>&1
i=0
for var in 0 1 4 5; do
  export TLB_VAR=$var
  g=0
  while read -r grp; do
    [ -z "$grp" ] && continue
    g=$((g+1))
    timeout 300 rocprofv3 --pmc $grp -d "$OUT/var${var}_pmc$g" -- ./tools/tlb_bench 30 > "$OUT/var${var}_pmc$g.log" 2>&1
  done <<'GROUPS'
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_SERIALIZATION_STALL_sum
TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum
GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_EA_BUSY
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_GATE_EN1_sum
GROUPS
done
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*.db" -delete
