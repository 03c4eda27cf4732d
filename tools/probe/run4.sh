export MODES=pairs DEVICE=0
for t in 16 14 12 10 8; do echo "--- threads $t (blocking sync)"; ABEA_HOST_THREADS=$t python tools/host_api_rate.py r9_10k_8kb 4 2>&1 | grep "rep [23]"; done
echo "--- threads 16 spin"; ABEA_HOST_SPIN=1 python tools/host_api_rate.py r9_10k_8kb 4 2>&1 | grep "rep [23]"
echo "--- threads 14 spin"; ABEA_HOST_SPIN=1 ABEA_HOST_THREADS=14 python tools/host_api_rate.py r9_10k_8kb 4 2>&1 | grep "rep [23]"
cat /sys/fs/cgroup/cpu.stat
