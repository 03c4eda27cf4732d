#!/bin/bash
# Round-2 first GPU call: box facts + baseline numbers that decide the host-pipeline design.
O=gpurun_out/probe1; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; free -g; lscpu | head -25; } > $O/sys.txt 2>&1
./build/probe/hostbw 4 > $O/hostbw.txt 2>&1
./build/probe/pcie_streams > $O/pcie.txt 2>&1
./build/probe/valu_rates2 > $O/valu2.txt 2>&1
python tools/host_api_rate.py 10000 > $O/host10k.txt 2>&1
