#!/bin/bash
# Round-4 GPU call L: the fuzz slices and the two-rank bench test on the final build (the full-size tests ran on the build before
# the last change of the fused phase, profiles/r04/i_gpu_tests_shipped_build.log; that change leaves the alignment-only path's outputs
# bit-identical, profiles/r04/j_map_from_walk_ab_10k.txt)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
timeout -k 5 150 python -m pytest tests -m gpu -x -q -k "fuzz or two_ranks" > $O/gpu_tests_fuzz.log 2>&1; echo "tests rc=$?" >> $O/steps.txt; tail -3 $O/gpu_tests_fuzz.log; cat $O/steps.txt
