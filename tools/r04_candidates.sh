#!/bin/bash
# Round-4 GPU call 1: settle the kernel candidates that round 3 prepared on the CPU interpreter, and take
# the first numbers of the new recalibration kernel.  Build the variants first on the build box: tools/build_candidates.sh.
#   gpurun --timeout 1200 -- 'bash tools/r04_candidates.sh'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04cand; mkdir -p $O
# 0. the new abea_recalib_kernel (LDS-DMA ring) against the oracle and the printed goldens, before anything is timed
timeout 600 python -m pytest tests -m gpu -x -q -k "scaling or recalib or 111_reads or fused or shim" > $O/scaling_tests.log 2>&1; tail -3 $O/scaling_tests.log
L="ship=f5c_amd/libabea_hip.so fifo=build/libabea_fifo.so walk2=build/libabea_walk2.so early=build/libabea_early.so w2e=build/libabea_w2e.so fifo_walk2=build/libabea_r4cand.so all=build/libabea_r4all.so ship2=f5c_amd/libabea_hip.so"
# 1. configs[1] and configs[2], >= 10 launches per variant, every output bit compared with the shipped build's
timeout 300 python tools/ab_quick.py $L --launches 11 > $O/ab_10k.log 2> $O/ab_10k.err; cat $O/ab_10k.log
timeout 900 python tools/ab_quick.py $L --config r9_100k_mixed --launches 11 > $O/ab_100k.log 2> $O/ab_100k.err; cat $O/ab_100k.log
# 2. latency of a small batch (f5c's default -K 512 -B 2M: 159 reads): the longest read alone on its SIMD decides
for v in ship sched early fifo r4all; do
  L=build/libabea_$v.so; [ $v = ship ] && L=f5c_amd/libabea_hip.so
  ABEA_LIB_PATH=$L MODES=pairs DEVICE=0 timeout 60 python tools/host_api_rate.py 512 6 2>/dev/null | grep "rep [2-5]" | sed "s/^/$v /" >> $O/small_batch.log
done
cat $O/small_batch.log
# 3. the fused align + scaling_single call next to the align-only call, with the per-kernel split
rocprofv3 --kernel-trace --stats -d $O/fused_prof -o fused -- env MODES=pairs,fused DEVICE=0 python tools/host_api_rate.py r9_100k_mixed 3 > $O/fused_rate.log 2> $O/fused_rate.err
cat $O/fused_rate.log
find $O/fused_prof -name "*kernel_stats.csv" -exec head -12 {} \;
