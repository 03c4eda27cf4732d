#!/bin/bash
# First GPU call of a round: everything the records of the SHIPPED build need, in one gpurun call (about 20 GPU-minutes):
#   1. the whole GPU suite                     -> gpurun_out/$TAG/gpu_tests.log
#   2. the bench line (driver's default flags)  -> gpurun_out/$TAG/bench.json
# Then, on the build box: copy the CSVs to profiles/<round>/ and run profiles/make_pmc_traffic.py (see its header).
#   gpurun --timeout 1500 -- 'bash tools/open_round.sh r04a'
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r04a}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 840 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout -k 10 330 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
# the rocprofv3 passes are separate, bounded calls since round 4 (a 25-minute call lost a box): tools/profile_r04.sh k10 | k100a | k100b
