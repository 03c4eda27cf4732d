#!/bin/bash
# Round-4 GPU call E: configs[1] profile set of the shipped build, the N2 kernel summary, the fused call's kernel timeline.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_r04.sh k10 r04e
bash tools/profile_r04.sh n2 r04e
O=gpurun_out/r04e
timeout -k 10 200 python tools/fused_trace.py 20000 /tmp/ft > $O/ft_gen.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused -o fused -- python tools/fused_trace.py 20000 /tmp/ft > $O/fused.log 2>&1
echo "fused trace rc=$?" >> $O/steps.txt; cat $O/fused.log | grep -v "^W2026"; cat $O/n2.log | grep parameters; cat $O/steps.txt
