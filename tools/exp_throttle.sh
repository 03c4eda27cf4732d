cd $GRAFT_REPO_ROOT
for t in 14 12 10; do echo "--- threads $t"; grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; MODES=pairs DEVICE=0 ABEA_HOST_THREADS=$t python tools/host_api_rate.py r9_10k_8kb 4 2>&1 | grep "rep [23]"; grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; done
