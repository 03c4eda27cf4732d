#!/bin/bash
# re-collection of the profile set on the shipped build after the walk window went from the whole group to radius 12
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_r03.sh k100 r03g > gpurun_out/r03g_k100.log 2>&1
bash tools/profile_r03.sh k10 r03g > gpurun_out/r03g_k10.log 2>&1
find gpurun_out/r03g -name "*counter_collection.csv" | wc -l
