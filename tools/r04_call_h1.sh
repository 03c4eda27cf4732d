#!/bin/bash
# Round-4 GPU call H1: the profile set of the SHIPPED build: configs[1] (all five passes), configs[2] (kernel trace, FETCH, WRITE)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/profile_r04.sh k10 r04h
bash tools/profile_r04.sh k100a r04h
