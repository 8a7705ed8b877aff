mkdir -p gpurun_out/r3q
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3q/trace -- python3 $GRAFT_REPO_ROOT/tools/_pipe_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r3q/log.txt 2>&1
cd $GRAFT_REPO_ROOT
grep "^call" gpurun_out/r3q/log.txt
