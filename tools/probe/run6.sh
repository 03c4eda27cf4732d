export MODES=pairs DEVICE=0 ARENA_GIB=150
ABEA_HOST_THREADS=12 python tools/host_api_rate.py r9_100k_mixed 3 2>&1 | grep -v "amdgpu.ids"
free -g | head -2
