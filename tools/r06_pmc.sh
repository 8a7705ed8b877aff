#!/bin/bash
# round 6: PMC passes of the round's default launches (one bench process per counter group: counters are never combined with tracing)
# usage: tools/r06_pmc.sh <outdir> <tag> "<bench args>"
O=$1; TAG=$2; ARGS=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_$TAG -- python3 bench.py --quick --steps 5 --warmup 1 $ARGS > $O/kt_$TAG.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum"; do
  d=$O/pmc_${TAG}_$(echo $grp | cut -d' ' -f1)
  timeout 900 rocprofv3 --pmc $grp -d $d -- python3 bench.py --quick --steps 2 --warmup 1 $ARGS > $d.log 2>&1
done
