#!/bin/bash
# Round-4 GPU call B: the two fixed tests, the kernel candidates on configs[1] (>= 10 launches each), small-batch latency.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 200 python -m pytest tests/test_process_chain.py tests/test_host_pipeline.py -m gpu -x -q -k "111_reads or torchrun" > $O/t_fixed.log 2>&1; echo "fixed rc=$?" >> $O/steps.txt; tail -2 $O/t_fixed.log
L="ship=f5c_amd/libabea_hip.so fifo=build/libabea_fifo.so walk2=build/libabea_walk2.so early=build/libabea_early.so w2e=build/libabea_w2e.so fifo_walk2=build/libabea_r4cand.so all=build/libabea_r4all.so ship2=f5c_amd/libabea_hip.so"
timeout 400 python tools/ab_quick.py $L --launches 11 > $O/ab_10k.log 2> $O/ab_10k.err; echo "ab10k rc=$?" >> $O/steps.txt; cat $O/ab_10k.log
for v in ship sched early fifo r4all ship; do
  P=build/libabea_$v.so; [ $v = ship ] && P=f5c_amd/libabea_hip.so
  ABEA_LIB_PATH=$P MODES=pairs DEVICE=0 timeout 90 python tools/host_api_rate.py 512 8 2>/dev/null | grep "rep [3-7]" | sed "s/^/$v /" >> $O/small_batch.log
done
cat $O/small_batch.log; cat $O/steps.txt
