#!/bin/bash
# in-situ marginal cost of a fast-class VALU, a slow-class VALU and a SALU instruction in the interior fill loop
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pad; mkdir -p $O; : > $O/pad.log
for rep in 1 2; do for v in base fast8 slow4 salu8; do
  ABEA_LIB_PATH=build/libabea_pad_$v.so timeout 200 python tools/ab_compare.py run /tmp/$v.npz 2>/dev/null | grep "kernel ms" | sed "s/^/$v /" >> $O/pad.log
done; done
for v in fast8 slow4 salu8; do timeout 60 python tools/ab_compare.py compare /tmp/base.npz /tmp/$v.npz >> $O/pad.log 2>&1; done
cat $O/pad.log
