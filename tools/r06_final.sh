#!/bin/bash
# round 6, final tree: the -m gpu suite + smoke as the driver runs them, the default bench line, two ranks sharing the box's GPU, and the
# kernel trace of the default command
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_final; mkdir -p $O
( time timeout 2400 python3 -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -25 $O/pytest.log | cut -c1-200
( time python3 -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
( time python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
tail -3 $O/bench_default.err
( time MOVI_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --no-sustained ) > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "rc=$?" >> $O/bench_n2_shared.err
tail -3 $O/bench_n2_shared.err
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 rocprofv3 --kernel-trace --stats -d $O/kt -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1
python3 tools/prof_summary.py $O/kt > $O/kt_summary.txt 2>&1; find $O -name "*.db" -delete
head -30 $O/kt_summary.txt | cut -c1-200
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
bt=d.get("big_table",{}); lr=d.get("long_reads",{})
print("c2 %.2f ms %.4f frac %.4f traffic %s kernel %s"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["traffic"],d["roofline"]["kernel"]))
print("parity", d.get("parity_sample_ok"), "cpu", d.get("cpu_baseline",{}).get("value"), "cores", d.get("cpu_baseline",{}).get("cores"))
print("mask_path", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.get("mask_path",{}).items()})
print("zml", d.get("zml",{}).get("value"), d.get("zml",{}).get("roofline",{}).get("frac"), "count", d.get("count",{}).get("value"), d.get("count",{}).get("roofline",{}).get("frac"))
print("long_reads %.2f frac %.4f (classify %.2f / %.2f)"%(lr.get("value",-1), lr.get("roofline",{}).get("frac",-1), lr.get("classify_vector_and_bins",{}).get("value",-1), lr.get("classify_bins_only",{}).get("value",-1)))
print("few", d.get("few_long_reads"))
print("big_table %.2f frac %.4f parity %s count %.2f zml %s"%(bt.get("value",-1), bt.get("roofline",{}).get("frac",-1), bt.get("parity_sample_ok"), bt.get("count",{}).get("value",-1), bt.get("zml",{}).get("value")))
c8=bt.get("count_blocked_thresholds",{})
print("c5 mode8", {k:c8.get(k) for k in ("value","expand_s","expand_gb_s","parity_sample_ok","index_upload_s","error")}, c8.get("roofline",{}).get("frac"), c8.get("roofline",{}).get("algorithmic_bytes_per_base"), c8.get("roofline",{}).get("resident_layout"))
print("host", {k:v for k,v in d.get("host_path",{}).items() if k!="note" and not k.endswith("_ok")})
print("big host", bt.get("host_path"))
print("cli", {k:(v.get("value") if isinstance(v,dict) else v) for k,v in d.get("cli_path",{}).items() if k not in ("note","unit")})
print("wall_s", d.get("wall_s"), "rss", d.get("host_peak_rss_mb"))
PY
