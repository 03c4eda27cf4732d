#!/bin/bash
# Round-4 profiles of the SHIPPED build (run on the GPU box through gpurun), one stage per call so that no call is long:
#   k10   configs[1] device-resident command: kernel trace + SQ passes A/B + FETCH + WRITE (batch cached in /tmp after the first run)
#   k100a configs[2] (the bench workload): kernel trace, FETCH, WRITE         k100b: SQ passes A/B
#   n2    event detection (DNA and RNA) kernel trace, single-process generator (no pool under rocprofv3)
#   ab100 configs[2] A/B: the round-3 shipped library (build/libabea_r03.so, built from commit f6e5513) against this one
# Every pass is bounded by `timeout -k 10` (TERM to the process group, KILL 10 s later: a hung profiler must not hold the box).
# Output: gpurun_out/$TAG/; copy what is to be judged into profiles/r04/ and run profiles/make_pmc_traffic.py.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
STAGE=${1:-k10}; TAG=${2:-r04e}
O=gpurun_out/$TAG; mkdir -p $O
SQA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"
SQB="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"
pass() {   # name, timeout, rocprofv3 options ..., -- command
  local name=$1 to=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $to rocprofv3 --kernel-trace --output-format csv -d $O/$name -o $name "$@" > $O/$name.log 2>&1
  echo "$name rc=$? $(( $(date +%s) - t0 )) s" >> $O/steps.txt
}
if [ "$STAGE" = "k10" ]; then
  DEV10="python bench.py --mode device --config r9_10k_8kb --device-steps 3 --no-cpu-baseline --arena-gib 40 --batch-cache /tmp/bc"
  timeout -k 10 200 $DEV10 > $O/dev10k.json 2> $O/dev10k.err
  pass kt10k 150 --stats -- $DEV10
  pass pmc_sqa10k 150 --pmc $SQA -- $DEV10
  pass pmc_sqb10k 150 --pmc $SQB GRBM_GUI_ACTIVE -- $DEV10
  pass pmc_fetch10k 150 --pmc FETCH_SIZE -- $DEV10
  pass pmc_write10k 150 --pmc WRITE_SIZE -- $DEV10
fi
DEV100="python bench.py --mode device --device-steps 2 --no-cpu-baseline"
if [ "$STAGE" = "k100a" ]; then
  pass kt100k 330 --stats -- $DEV100
  pass pmc_fetch100k 330 --pmc FETCH_SIZE -- $DEV100
  pass pmc_write100k 330 --pmc WRITE_SIZE -- $DEV100
fi
if [ "$STAGE" = "k100b" ]; then
  pass pmc_sqa100k 330 --pmc $SQA -- $DEV100
  pass pmc_sqb100k 330 --pmc $SQB GRBM_GUI_ACTIVE -- $DEV100
fi
if [ "$STAGE" = "n2" ]; then
  pass n2 240 --stats -- python tools/n2_profile.py 2048
fi
if [ "$STAGE" = "ab100" ]; then
  timeout -k 10 600 python tools/ab_quick.py r03=build/libabea_r03.so r04=f5c_amd/libabea_hip.so r03b=build/libabea_r03.so --config r9_100k_mixed --launches 11 --scaling-launches 4 > $O/ab_100k.log 2> $O/ab_100k.err
  echo "ab100 rc=$?" >> $O/steps.txt; cat $O/ab_100k.log
fi
find $O -name "*kernel_trace.csv" -size +20M -delete
cat $O/steps.txt; find $O -name "*.csv" | head -40
