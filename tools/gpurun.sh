#!/bin/bash
# Launch stages of tools/gpu_call.sh on the GPU box with the tree identified: the snapshot gpurun ships carries no .git, so the
# commit (and whether the working tree differs from it) is stamped into .gpurun_head first; gpu_call.sh copies it into its logs.
#   bash tools/gpurun.sh <timeout-seconds> <stage[,stage...]> [tag-prefix]      (environment for the stages: ENV="A=1 B=2")
cd "$(dirname "$0")/.." || exit 1
T=${1:?timeout}; STAGES=${2:?stage}; PRE=${3:-r06}
DIRTY=$(git status --porcelain -- f5c_amd include tests bench.py __graft_entry__.py oracle tools | grep -v '^??' | wc -l)
echo "commit $(git rev-parse HEAD) ($(git log -1 --format=%s | cut -c1-60)); tracked files differing from it: $DIRTY" > .gpurun_head
CMD=""
for s in ${STAGES//,/ }; do CMD="$CMD $ENV bash tools/gpu_call.sh $s ${PRE}_$s;"; done
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$CMD"
