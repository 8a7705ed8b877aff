#!/bin/bash
# round 6, second GPU pass: deep rows (three bases per gather, 21.33 B per row) -- parity tests, A/B against the look-ahead rows on c2 / c3 / the random
# 10 M-row table, PMC of both; the reset-mask path again (coalesced expand kernel, host expansion with / without non-temporal stores)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python3 -m pytest tests/test_deep_rows_gpu.py tests/test_mask_gpu.py tests/test_kernel_coverage_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for wl in c2 c3 c2synth; do for deep in 0 1; do
  timeout 400 python3 bench.py --quick --workload $wl --opt deep_rows=$deep > $O/${wl}_deep$deep.json 2> $O/${wl}_deep$deep.err
done; done
for cap in 7 11 13; do timeout 400 python3 bench.py --quick --workload c2 --opt deep_rows=1 --waves-per-cu $cap > $O/c2_deep1_cap$cap.json 2> /dev/null; done
timeout 400 python3 bench.py --quick --workload c2 --opt deep_rows=1 --opt repo_hints=0 > $O/c2_deep1_nohints.json 2> /dev/null
for wl in c2 c3; do for deep in 0 1; do
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_DRAM_sum" "FETCH_SIZE"; do
    d=$O/pmc_${wl}_deep${deep}_$(echo $grp | cut -d' ' -f1)
    timeout 400 rocprofv3 --pmc $grp -d $d -- python3 bench.py --quick --steps 2 --warmup 1 --workload $wl --opt deep_rows=$deep > $d.log 2>&1
  done
done; done
python3 tools/prof_summary.py $O > $O/pmc_summary.txt 2>&1; find $O -name "*.db" -delete
grep -h "PMC.*flatp" $O/pmc_summary.txt | cut -c1-230 | sort | uniq > $O/pmc_walk.txt
for nt in 1 0; do
  ( MOVI_EXPAND_NT=$nt timeout 900 python3 bench.py --no-big-table --no-long-reads --opt deep_rows=0 ) > $O/bench_c2_nt$nt.json 2> $O/bench_c2_nt$nt.err
done
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    c=d["config"]
    print("%-28s value %.2f kernel_ms %.4f iters/base %s simt %s %s"%(os.path.basename(f), d["value"] or -1, d["roofline"]["kernel_ms_avg"], c.get("iterations_per_base"), c.get("simt_efficiency"), d["roofline"]["kernel"]))
    for k in ("mask_path","host_path"):
        if k in d: print("   ",k,json.dumps({a:b for a,b in d[k].items() if a!="note"}))
PY
