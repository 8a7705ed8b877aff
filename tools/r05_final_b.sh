#!/bin/bash
# round 5, final tree: the default bench line (as the driver runs it), two ranks sharing the box's GPU, and the kernel trace of the default command
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r05_final_b; mkdir -p $O
( time python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
tail -3 $O/bench_default.err
( time MOVI_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --no-sustained ) > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "rc=$?" >> $O/bench_n2_shared.err
tail -3 $O/bench_n2_shared.err
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1200 rocprofv3 --kernel-trace --stats -d $O/kt -- python3 bench.py --no-cpu-baseline > $O/kt.log 2>&1
python3 tools/prof_summary.py $O/kt > $O/kt_summary.txt 2>&1; find $O -name "*.db" -delete
head -14 $O/kt_summary.txt | cut -c1-200
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2 %.2f frac %.4f traffic %s | long_reads %.2f (classify %.2f / %.2f) | big_table %.2f count %.2f parity %s | host %s | cli %s"%(d["value"],d["roofline"]["frac"],d["roofline"]["traffic"],d["long_reads"]["value"],d["long_reads"]["classify_vector_and_bins"]["value"],d["long_reads"]["classify_bins_only"]["value"],d["big_table"]["value"],d["big_table"]["count"]["value"],d["big_table"].get("parity_sample_ok"),d.get("host_path",{}).get("page_locked"),{k:v.get("value") for k,v in d.get("cli_path",{}).items() if isinstance(v,dict)}))
print("big_table host", d["big_table"].get("host_path"), "cli", d["big_table"].get("cli_path"))
PY
