#!/bin/bash
# the SALU-diet build against the previous build (bit-identical?) and the same loop at 5 / 6 / 7 waves per SIMD
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03occ; mkdir -p $O; : > $O/occ.log
ABEA_LIB_PATH=build/ab_old/libabea_old.so timeout 300 python tools/ab_compare.py run /tmp/old.npz 2>/dev/null | grep "kernel ms" | sed "s/^/r02 /" >> $O/occ.log
for rep in 1 2; do for v in diet tied38 tied22 tied14; do
  L=build/libabea_$v.so; [ $v = diet ] && L=f5c_amd/libabea_hip.so
  ABEA_LIB_PATH=$L timeout 200 python tools/ab_compare.py run /tmp/$v.npz 2>/dev/null | grep "kernel ms" | sed "s/^/$v /" >> $O/occ.log
done; done
for v in diet tied38 tied22 tied14; do timeout 60 python tools/ab_compare.py compare /tmp/old.npz /tmp/$v.npz >> $O/occ.log 2>&1; done
cat $O/occ.log
