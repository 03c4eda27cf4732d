#!/bin/bash
# A/B/C on configs[1]: previous build (build/ab_old), working tree (asm), working tree C++ twin (build/libabea_noasm.so)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03ab2}; mkdir -p $O
ABEA_LIB_PATH=build/ab_old/libabea_old.so timeout 300 python tools/ab_compare.py run /tmp/old.npz > $O/ab.log 2>&1
timeout 300 python tools/ab_compare.py run /tmp/new.npz >> $O/ab.log 2>&1
timeout 100 python tools/ab_compare.py compare /tmp/old.npz /tmp/new.npz >> $O/ab.log 2>&1
ABEA_LIB_PATH=build/libabea_noasm.so timeout 600 python tools/ab_compare.py run /tmp/twin.npz r9_10k_8kb 2000 >> $O/ab.log 2>&1
timeout 300 python tools/ab_compare.py run /tmp/new2k.npz r9_10k_8kb 2000 >> $O/ab.log 2>&1
timeout 100 python tools/ab_compare.py compare /tmp/twin.npz /tmp/new2k.npz >> $O/ab.log 2>&1
cat $O/ab.log
