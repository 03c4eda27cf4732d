#!/bin/bash
# Round-4 GPU call K (last seconds of the budget): everything of the GPU suite except the full-size and fuzz tests, on the final build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
timeout -k 5 170 python -m pytest tests -m gpu -x -q -k "not full_size and not fuzz and not two_ranks" > $O/gpu_tests_fast.log 2>&1; echo "tests rc=$?" >> $O/steps.txt; tail -4 $O/gpu_tests_fast.log; cat $O/steps.txt
