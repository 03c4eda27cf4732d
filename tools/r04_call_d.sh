#!/bin/bash
# Round-4 GPU call D: the whole GPU suite on the build that ships, then the bench line (driver's default flags).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 840 python -m pytest tests -m gpu -x -q --durations=12 > $O/gpu_tests.log 2>&1; echo "tests rc=$?" >> $O/steps.txt; tail -18 $O/gpu_tests.log
timeout -k 10 330 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/steps.txt; tail -c 1500 $O/bench.json
cat $O/steps.txt
