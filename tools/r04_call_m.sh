#!/bin/bash
# Round-4 GPU call M (the last 70 s of the budget): the fused / scaling / process-chain tests with the map crossing PCIe as one
# count byte per k-mer
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
timeout -k 3 50 python -m pytest tests -m gpu -x -q -k "(fused or process or shim or 111_reads or scaling or submit) and not fuzz and not two_ranks" > $O/t_fused.log 2>&1; echo "tests rc=$?" >> $O/steps.txt; tail -5 $O/t_fused.log; cat $O/steps.txt
