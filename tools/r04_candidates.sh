#!/bin/bash
# Round-4 opener for the kernel candidates that round 3 prepared on the CPU interpreter: build them on the
# build box first (hipcc cross-compiles):
#   ABEA_FIFO=1               python tools/gen_fill_asm.py && <hipcc line of f5c_amd/csrc/Makefile> -DABEA_EXP -DABEA_FIFO -o build/libabea_fifo.so
#   ABEA_WALK2=1              python tools/gen_fill_asm.py && <hipcc line> -DABEA_EXP               -o build/libabea_walk2.so
#   ABEA_FIFO=1 ABEA_WALK2=1  python tools/gen_fill_asm.py && <hipcc line> -DABEA_EXP -DABEA_FIFO   -o build/libabea_r4cand.so
#   ABEA_EARLY=1              python tools/gen_fill_asm.py && <hipcc line> -DABEA_EXP               -o build/libabea_early.so
#   ABEA_FIFO=1 ABEA_WALK2=1 ABEA_EARLY=1  ... -DABEA_EXP -DABEA_FIFO                                -o build/libabea_r4all.so
#   ABEA_SCHED=1              python tools/gen_fill_asm.py && <hipcc line> -DABEA_EXP               -o build/libabea_sched.so
# (the last one: independent work interleaved into the emission chain; with packed f32 on top it changed nothing at 4 waves
#  per SIMD, but it is what a LONE wave lacks: an f5c-default batch lasts as long as its longest read, 244 ns per band for a wave alone on its SIMD)
# then on the GPU:  gpurun --timeout 300 -- 'bash tools/r04_candidates.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04cand; mkdir -p $O
L="ship=f5c_amd/libabea_hip.so fifo=build/libabea_fifo.so walk2=build/libabea_walk2.so early=build/libabea_early.so fifo_walk2=build/libabea_r4cand.so all=build/libabea_r4all.so ship2=f5c_amd/libabea_hip.so"
timeout 120 python tools/ab_quick.py $L --launches 6 > $O/ab_10k.log 2> $O/ab_10k.err; cat $O/ab_10k.log
timeout 170 python tools/ab_quick.py $L --config r9_100k_mixed --reads 30000 --launches 4 > $O/ab_30k.log 2> $O/ab_30k.err; cat $O/ab_30k.log
# latency of a small batch (f5c's default -K 512): the longest read alone on its SIMD decides
for v in ship sched early fifo r4all; do
  L=build/libabea_$v.so; [ $v = ship ] && L=f5c_amd/libabea_hip.so
  ABEA_LIB_PATH=$L MODES=pairs DEVICE=0 timeout 60 python tools/host_api_rate.py 512 5 2>/dev/null | grep "rep [2-4]" | sed "s/^/$v /" >> $O/small_batch.log
done
cat $O/small_batch.log
