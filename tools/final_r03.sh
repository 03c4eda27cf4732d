#!/bin/bash
# Round-3 closing run on ONE box: the whole gpu test suite, the bench line of the shipped build, then the profiles.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -4 $O/gputests.log
python bench.py --steps 8 --warmup 2 > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
bash tools/profile_r03.sh k10 r03f > $O/k10.log 2>&1
bash tools/profile_r03.sh k100 r03f > $O/k100.log 2>&1
find $O -name "*.csv" | wc -l
