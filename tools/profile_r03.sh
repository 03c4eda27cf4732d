#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun).  Stages: list | k10 | k100 | all
#   list : rocprofv3 -L counter list (which SQ_* names this gfx950 build knows)
#   k10  : configs[1] device-resident command: kernel trace + SQ passes A/B + FETCH + WRITE
#   k100 : configs[2] (the bench workload) the same five passes; the batch generator spawns clean workers under
#          rocprofv3 (f5c_amd/synth.py), so no pass depends on a forked pool any more
# Output: gpurun_out/$TAG/ (TAG defaults to r03); copy what is to be judged into profiles/r03/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
STAGE=${1:-k10}; TAG=${2:-r03}
O=gpurun_out/$TAG; mkdir -p $O
SQA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"
SQB="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"
passes() {   # $1 = name suffix, $2 = per-pass timeout, rest = command
  local sfx=$1 to=$2; shift 2
  timeout $to rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$sfx -o kt -- "$@" > $O/kt$sfx.log 2>&1
  timeout $to rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d $O/pmc_sqa$sfx -o pmc -- "$@" > $O/pmc_sqa$sfx.log 2>&1
  timeout $to rocprofv3 --kernel-trace --pmc $SQB GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sqb$sfx -o pmc -- "$@" > $O/pmc_sqb$sfx.log 2>&1 \
    || { for c in $SQB; do timeout $to rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_sqb1_${c}$sfx -o pmc -- "$@" > $O/pmc_sqb1_${c}$sfx.log 2>&1; done; }
  timeout $to rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch$sfx -o pmc -- "$@" > $O/pmc_fetch$sfx.log 2>&1
  timeout $to rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write$sfx -o pmc -- "$@" > $O/pmc_write$sfx.log 2>&1
}
if [ "$STAGE" = "list" ] || [ "$STAGE" = "all" ]; then
  timeout 60 rocprofv3 -L > $O/counters_list.txt 2>&1
  grep -c . $O/counters_list.txt
fi
if [ "$STAGE" = "k10" ] || [ "$STAGE" = "all" ]; then
  DEV10="python bench.py --mode device --config r9_10k_8kb --device-steps 3 --no-cpu-baseline --arena-gib 40 --batch-cache /tmp/bc"
  timeout 200 $DEV10 > $O/dev10k.json 2> $O/dev10k.err
  passes 10k 200 $DEV10
fi
if [ "$STAGE" = "k100" ] || [ "$STAGE" = "all" ]; then
  DEV100="python bench.py --mode device --device-steps 2 --no-cpu-baseline"
  passes 100k 420 $DEV100
fi
# keep the merged output small: per-dispatch kernel traces of PMC runs above 20 MB are not needed
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*.csv" | head -60
