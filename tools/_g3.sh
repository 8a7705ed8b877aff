mkdir -p gpurun_out/r3c
env | grep -i -E "^(hip|hsa|amd|gpu|roc)" > gpurun_out/r3c/env.txt
./tools/overlap_bench > gpurun_out/r3c/overlap.txt 2>&1
echo "--- HSA_ENABLE_SDMA=0" >> gpurun_out/r3c/overlap.txt
HSA_ENABLE_SDMA=0 ./tools/overlap_bench >> gpurun_out/r3c/overlap.txt 2>&1
echo "--- GPU_MAX_HW_QUEUES=8" >> gpurun_out/r3c/overlap.txt
GPU_MAX_HW_QUEUES=8 ./tools/overlap_bench >> gpurun_out/r3c/overlap.txt 2>&1
cat gpurun_out/r3c/env.txt gpurun_out/r3c/overlap.txt
