#!/bin/bash
# Round-4 GPU call H2: configs[2] SQ passes of the shipped build, then the configs[2] A/B against the round-3 library and the
# device-resident cost of the fused scaling_single phase.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_r04.sh k100b r04h
bash tools/profile_r04.sh ab100 r04h
