#!/bin/bash
# First GPU call of a round: everything the records of the SHIPPED build need, in one gpurun call (about 20 GPU-minutes):
#   1. the whole GPU suite                     -> gpurun_out/$TAG/gpu_tests.log
#   2. the bench line (driver's default flags)  -> gpurun_out/$TAG/bench.json
#   3. the rocprofv3 passes of tools/profile_r03.sh on configs[1] and configs[2] (kernel trace + SQ A/B + FETCH + WRITE)
# Then, on the build box: copy the CSVs to profiles/<round>/ and run profiles/make_pmc_traffic.py (see its header).
#   gpurun --timeout 1500 -- 'bash tools/open_round.sh r04a'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r04a}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
bash tools/profile_r03.sh k10 $TAG > $O/k10.log 2>&1
bash tools/profile_r03.sh k100 $TAG > $O/k100.log 2>&1
find $O -name "*counter_collection.csv" | wc -l
