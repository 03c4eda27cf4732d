cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MODES=pairs DEVICE=0
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/trace2 -o t -- python tools/host_api_rate.py r9_10k_8kb 2 > gpurun_out/trace2.log 2>&1
ls gpurun_out/trace2
