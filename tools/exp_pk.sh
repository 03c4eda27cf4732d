#!/bin/bash
# packed-f32 variants of the interior band loop (ABEA_PK=1 / 2) against the shipped build: kernel time, outputs bit for bit
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pk; mkdir -p $O
timeout 100 python tools/ab_quick.py ship=f5c_amd/libabea_hip.so pk=build/libabea_pk.so pks=build/libabea_pks.so ship2=f5c_amd/libabea_hip.so > $O/pk.log 2>$O/pk.err
tail -3 $O/pk.err; cat $O/pk.log
