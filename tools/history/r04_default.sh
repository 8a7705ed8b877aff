#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r04_default; mkdir -p $O
( time python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err
tail -3 $O/bench_default.err
( time MOVI_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --no-sustained ) > $O/bench_n2_shared.json 2> $O/bench_n2_shared.err; echo "rc=$?" >> $O/bench_n2_shared.err
tail -3 $O/bench_n2_shared.err
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -- python3 bench.py --no-big-table --no-cpu-baseline > $O/kt.log 2>&1
python3 tools/prof_summary.py $O/kt > $O/kt_summary.txt 2>&1; find $O -name "*.db" -delete
head -12 $O/kt_summary.txt | cut -c1-200
