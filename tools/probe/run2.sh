cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MODES=pairs DEVICE=0
echo "--- slots=1"; ABEA_HOST_SLOTS=1 python tools/host_api_rate.py r9_10k_8kb 3 2>&1 | grep "rep [12]"
echo "--- slots=2"; ABEA_HOST_SLOTS=2 python tools/host_api_rate.py r9_10k_8kb 3 2>&1 | grep "rep [12]"
echo "--- slots=8"; ABEA_HOST_SLOTS=8 python tools/host_api_rate.py r9_10k_8kb 3 2>&1 | grep "rep [12]"
echo "--- one chunk"; ABEA_HOST_CHUNK_READS=100000 ABEA_HOST_CHUNK_READS_MAX=100000 python tools/host_api_rate.py r9_10k_8kb 3 2>&1 | grep "rep [12]"
echo "--- GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 python tools/host_api_rate.py r9_10k_8kb 3 2>&1 | grep "rep [12]"
rocprofv3 --kernel-trace -d gpurun_out/trace1 -o t -- python tools/host_api_rate.py r9_10k_8kb 2 > gpurun_out/trace1.log 2>&1
ls gpurun_out/trace1 | head
