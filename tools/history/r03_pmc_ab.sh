#!/bin/bash
# usage: tools/r03_pmc_ab.sh <outdir>: fabric reads / TCP->L2 requests of c2 under a few launch variants (one PMC pass each)
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
python3 bench.py --quick --steps 2 --warmup 1 > /dev/null 2>&1     # builds the pangenome cache once
pass() { name=$1; shift
  timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT/$name" -- python3 bench.py --quick --steps 3 --warmup 1 "$@" > "$OUT/$name.log" 2>&1
  echo "== $name: $*" >> "$OUT/summary.txt"
  python3 tools/prof_summary.py "$OUT/$name" 2>/dev/null | grep "PMC" | grep "pml_kernel_flatp" >> "$OUT/summary.txt"
  tail -1 "$OUT/$name.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value', round(d['value'],2), 'it/base', d['config']['iterations_per_base'], 'simt', d['config']['simt_efficiency'])" >> "$OUT/summary.txt" 2>/dev/null
  find "$OUT/$name" -name "*.db" -delete
}
pass default
pass cap7 --waves-per-cu 7
pass cap12 --waves-per-cu 12
pass nokmer --kmer-k 0
pass k10 --kmer-k 10
pass plain --ahead-rows 0
