#!/bin/bash
# Round-4 opener for the kernel candidates that round 3 prepared on the CPU interpreter: build them on the
# build box first (hipcc cross-compiles):
#   ABEA_FIFO=1               python tools/gen_fill_asm.py && <hipcc line of f5c_amd/csrc/Makefile> -DABEA_EXP -DABEA_FIFO -o build/libabea_fifo.so
#   ABEA_WALK2=1              python tools/gen_fill_asm.py && <hipcc line> -DABEA_EXP               -o build/libabea_walk2.so
#   ABEA_FIFO=1 ABEA_WALK2=1  python tools/gen_fill_asm.py && <hipcc line> -DABEA_EXP -DABEA_FIFO   -o build/libabea_r4cand.so
# then on the GPU:  gpurun --timeout 300 -- 'bash tools/r04_candidates.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04cand; mkdir -p $O
L="ship=f5c_amd/libabea_hip.so fifo=build/libabea_fifo.so walk2=build/libabea_walk2.so both=build/libabea_r4cand.so ship2=f5c_amd/libabea_hip.so"
timeout 120 python tools/ab_quick.py $L --launches 6 > $O/ab_10k.log 2> $O/ab_10k.err; cat $O/ab_10k.log
timeout 170 python tools/ab_quick.py $L --config r9_100k_mixed --reads 30000 --launches 4 > $O/ab_30k.log 2> $O/ab_30k.err; cat $O/ab_30k.log
