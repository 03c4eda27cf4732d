export MODES=pairs DEVICE=0 ABEA_HOST_TRACE=1
for t in 16 12; do echo "--- threads $t"; ABEA_HOST_THREADS=$t python tools/host_api_rate.py r9_10k_8kb 3 2>&1 | grep -v "amdgpu.ids" | tail -42; done
