mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overlapped or invariant" 2>&1 | tail -4 > gpurun_out/r3e/tests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3e/trace -- python3 $GRAFT_REPO_ROOT/tools/_pipe_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r3e/log.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python tools/host_rate.py > gpurun_out/r3e/host_rate.txt 2>&1
cat gpurun_out/r3e/tests.txt gpurun_out/r3e/host_rate.txt; grep "^call" gpurun_out/r3e/log.txt
