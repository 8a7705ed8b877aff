#!/bin/bash
# round 6: movi_pml_host, both ways down side by side (share sweep); few long reads under the counters
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_n; mkdir -p $O
timeout 900 python3 -m pytest tests/test_mask_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
timeout 600 python3 tools/r06_n.py 2>&1 | grep -v amdgpu.ids | tee $O/share_sweep.txt
bash tools/r06_pmc.sh $O few "--workload c3 --reads 25000"
python3 tools/prof_summary.py $O > $O/summary.txt 2>&1
grep -h "seg_\|flatp" $O/summary.txt | cut -c1-260 | head -60
