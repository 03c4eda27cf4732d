#!/bin/bash
# Build the experiment variants of the library for tools/r04_candidates.sh (build box; hipcc cross-compiles).
# Each variant = generator switches -> abea_fill_exp.inc / abea_walk_exp.inc -> build/libabea_<name>.so (-DABEA_EXP).
set -e
cd "$(dirname "$0")/.."
mkdir -p build
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value"
SRCS="abea_kernels.hip abea_hmm.hip abea_capi.cpp abea_host.cpp abea_hmm.cpp abea_rsq.cpp abea_process.cpp f5c_shim.cpp"
build() {  # name, extra -D flags, generator env...
  name=$1; defs=$2; shift 2
  env "$@" python tools/gen_fill_asm.py > /dev/null
  (cd f5c_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -DABEA_EXP $defs -x hip -shared $SRCS -o ../../build/libabea_$name.so -lpthread)
  echo built $name
}
build fifo   "-DABEA_FIFO" ABEA_FIFO=1
build walk2  ""            ABEA_WALK2=1
build early  ""            ABEA_EARLY=1
build sched  ""            ABEA_SCHED=1
build r4cand "-DABEA_FIFO" ABEA_FIFO=1 ABEA_WALK2=1
build r4all  "-DABEA_FIFO" ABEA_FIFO=1 ABEA_WALK2=1 ABEA_EARLY=1
build w2e    ""            ABEA_WALK2=1 ABEA_EARLY=1
