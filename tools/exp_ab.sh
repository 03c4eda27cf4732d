#!/bin/bash
# A/B of the working-tree library against build/ab_old/libabea_old.so on configs[1]: bit-identical outputs + kernel ms
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03ab}; mkdir -p $O
ABEA_LIB_PATH=build/ab_old/libabea_old.so timeout 300 python tools/ab_compare.py run /tmp/old.npz > $O/ab.log 2>&1
timeout 300 python tools/ab_compare.py run /tmp/new.npz >> $O/ab.log 2>&1
timeout 100 python tools/ab_compare.py compare /tmp/old.npz /tmp/new.npz >> $O/ab.log 2>&1
timeout 300 python tools/walk_stats.py >> $O/ab.log 2>&1
cat $O/ab.log
