#!/bin/bash
# round 6, first GPU pass: reset-mask path -- parity tests, the bench line's mask legs, A/B of movi_pml_device direct vs mask walk + expand
# (c2, c3, c4), PMC (SQ_INSTS_VALU, WRITE_SIZE) of the vector walk against the mask walk
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python3 -m pytest tests/test_mask_gpu.py tests/test_kernel_coverage_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
( time timeout 900 python3 bench.py --no-big-table --no-long-reads ) > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?" >> $O/bench_c2.err
for wl in c2 c3; do for via in 0 1; do
  timeout 400 python3 bench.py --quick --workload $wl --opt pml_via_mask=$via > $O/${wl}_via$via.json 2> $O/${wl}_via$via.err
done; done
for wl in c2 c3; do for via in 0 1; do
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    d=$O/pmc_${wl}_via${via}_$(echo $grp | cut -d' ' -f1)
    timeout 400 rocprofv3 --pmc $grp -d $d -- python3 bench.py --quick --steps 2 --warmup 1 --workload $wl --opt pml_via_mask=$via > $d.log 2>&1
  done
done; done
python3 tools/prof_summary.py $O > $O/pmc_summary.txt 2>&1; find $O -name "*.db" -delete
grep -h "PMC.*flatp\|PMC.*expand" $O/pmc_summary.txt | cut -c1-230 | sort | uniq > $O/pmc_walk.txt
timeout 900 python3 bench.py --quick --workload c4 --opt pml_via_mask=0 > $O/c4_via0.json 2> $O/c4_via0.err
timeout 900 python3 bench.py --quick --workload c4 --opt pml_via_mask=1 > $O/c4_via1.json 2> $O/c4_via1.err
python3 - $O <<'PY'
import json,sys,glob,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f),"unreadable",e); continue
    print(os.path.basename(f), "value %.2f kernel_ms %.4f %s"%(d["value"] or -1, d["roofline"]["kernel_ms_avg"], d["roofline"]["kernel"]))
    for k in ("mask_path","host_path"):
        if k in d: print("   ",k,json.dumps(d[k]))
PY
