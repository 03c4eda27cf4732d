#!/bin/bash
# GPU calls, one parameterised script:
#   bash tools/gpurun.sh <timeout-s> <stage> [tag]          (from the build container: stamps the commit, then calls gpurun)
#   gpurun --timeout T -- 'bash tools/gpu_call.sh <stage> [tag]'
# Every step runs under its own `timeout -k 10` (TERM to the process group, KILL 10 s later: a hung kernel or profiler
# must not hold the box).  Output: gpurun_out/<tag>/ ; what is to be judged is copied into profiles/r06/ afterwards.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
STAGE=${1:-tests}; TAG=${2:-r06_$STAGE}
O=gpurun_out/$TAG; mkdir -p $O
step() {   # name, timeout, command ...
  local name=$1 to=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $to "$@" > $O/$name.log 2>&1
  echo "$name rc=$? $(( $(date +%s) - t0 )) s" >> $O/steps.txt
}
stamp() {  # which tree ran: the commit tools/gpurun.sh stamped (the snapshot carries no .git) + digests of what the suite exercises
  { echo "tree: $(cat .gpurun_head 2>/dev/null || echo 'no .gpurun_head (not launched through tools/gpurun.sh)')"
    echo "sha256 of the product + test sources as they ran:"
    find f5c_amd include tests bench.py __graft_entry__.py oracle -type f \( -name '*.py' -o -name '*.cpp' -o -name '*.hip' -o -name '*.h' -o -name '*.inc' -o -name '*.c' \) \
      | grep -v __pycache__ | sort | xargs sha256sum | sha256sum | cut -c1-64
    sha256sum f5c_amd/libabea_hip.so oracle/libabea_oracle.so 2>/dev/null; } > $1
}
case $STAGE in
  chain)      # what bounds the raw-signal chain (round-5 verdict item 2): link ceilings, table form x mover A/B, GPU-clock timeline, kernel trace
    step gen 300 python tools/chain_trace.py ${CHAIN_READS:-10000} /tmp/ct
    step chain_ab 900 python tools/chain_trace.py ${CHAIN_READS:-10000} /tmp/ct ${CHAIN_MODES:-full:kernel:trace packed:kernel:trace full:engine packed:engine:trace}
    grep -v "^\[abea\|^----" $O/chain_ab.log | tail -40
    step chain_kt 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain_kt -o chain -- python tools/chain_trace.py ${CHAIN_READS:-10000} /tmp/ct ${CHAIN_PROFILE_MODE:-packed:kernel}
    find $O/chain_kt -name "*kernel_stats.csv" | head -2
    ;;
  phase4)     # where the fused scaling_single phase spends its wave time (profile build: make -C f5c_amd/csrc prof)
    ABEA_LIB_PATH=build/libabea_prof.so step phase_profile 300 python tools/phase_profile.py --scaling --reads 10000
    grep -v "^  t=\|^  [0-9]" $O/phase_profile.log | tail -12
    ABEA_LIB_PATH=build/libabea_prof.so step phase_profile_align_only 300 python tools/phase_profile.py --reads 10000
    grep "^kernel\|^sum" $O/phase_profile_align_only.log
    ;;
  n2prof)     # the detector alone (device entry, 2048 reads): kernel trace + HBM counters of its kernels (separate passes)
    step n2_kt 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n2 -o n2 -- python tools/n2_profile.py 2048
    grep "parameters" $O/n2_kt.log
    step n2_fetch 300 rocprofv3 --kernel-trace --output-format csv -d $O/n2_fetch -o n2 --pmc FETCH_SIZE -- python tools/n2_profile.py 2048
    step n2_write 300 rocprofv3 --kernel-trace --output-format csv -d $O/n2_write -o n2 --pmc WRITE_SIZE -- python tools/n2_profile.py 2048
    step n2_sqa 300 rocprofv3 --kernel-trace --output-format csv -d $O/n2_sqa -o n2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -- python tools/n2_profile.py 2048
    step n2_sqb 300 rocprofv3 --kernel-trace --output-format csv -d $O/n2_sqb -o n2 --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- python tools/n2_profile.py 2048
    find $O -name "*.csv" | head
    ;;
  n2ab)       # the detector alone, no profiler: the fused common path against the array form (ABEA_EV_PATH=arrays), same process layout
    step n2_fused 200 python tools/n2_profile.py 2048
    ABEA_EV_PATH=arrays step n2_arrays 200 python tools/n2_profile.py 2048
    grep -H "parameters" $O/n2_fused.log $O/n2_arrays.log
    ;;
  chainslots) # the raw-signal entries against the number of chunk slots (chunks in flight): long reads hold a slot for their serial band chain
    step gen 300 python tools/chain_trace.py ${CHAIN_READS:-10000} /tmp/ct
    for n in ${SLOT_COUNTS:-6 9 12 16}; do
      ABEA_CHAIN_SLOTS=$n step chain_slots_$n 300 python tools/chain_trace.py ${CHAIN_READS:-10000} /tmp/ct packed:engine
      echo "slots $n"; grep "rep [12]" $O/chain_slots_$n.log | cut -c1-200
    done
    ;;
  stress)     # a test file over and over in fresh processes (a data race shows once in N runs): STRESS_FILE, STRESS_N
    ok=0; bad=0
    for i in $(seq 1 ${STRESS_N:-20}); do
      timeout -k 10 300 python -m pytest ${STRESS_FILE:-tests/test_host_plan_and_async.py} -m gpu -x -q -p no:cacheprovider > $O/stress_$i.log 2>&1
      rc=$?; if [ $rc = 0 ]; then ok=$((ok+1)); rm -f $O/stress_$i.log; else bad=$((bad+1)); echo "run $i rc=$rc"; tail -5 $O/stress_$i.log; fi
    done
    echo "stress ${STRESS_FILE:-tests/test_host_plan_and_async.py}: $ok ok, $bad failed" | tee $O/stress.txt
    ;;
  smoke)      # what the driver runs before the bench: __graft_entry__.build() (prebuilt files) and smoke()
    step smoke 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')"
    tail -3 $O/smoke.log
    ;;
  fuzz)       # long randomised sweeps of the detector against the oracle: device entry and host entry, two seeds each
    for seed in ${FUZZ_SEEDS:-61 62}; do
      step fuzz_dev_$seed $(( ${FUZZ_S:-150} + 120 )) python tools/fuzz_events.py ${FUZZ_S:-150} $seed
      step fuzz_host_$seed $(( ${FUZZ_S:-150} + 120 )) python tools/fuzz_events.py ${FUZZ_S:-150} $seed host
    done
    grep -h "fuzz OK\|Error\|assert" $O/fuzz_*.log | tee $O/fuzz_sweeps.txt
    ;;
  tests)      # the whole GPU suite at the stamped commit (the round's gate: parity tests first, infrastructure last — tests/conftest.py)
    stamp $O/gpu_tests_tree.txt; cat $O/gpu_tests_tree.txt
    step gpu_tests ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q --durations=12 -p no:cacheprovider
    cat $O/gpu_tests_tree.txt $O/gpu_tests.log > $O/gpu_tests_full_suite.log
    tail -22 $O/gpu_tests.log
    if [ -z "$NO_BENCH" ]; then
      t0=$(date +%s); timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s" >> $O/steps.txt
      python tools/bench_summary.py $O/bench.json
    fi
    ;;
  bench)      # the bench line alone (the driver's command)
    t0=$(date +%s); timeout -k 10 600 python bench.py ${BENCH_ARGS:-} > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s" >> $O/steps.txt
    tail -3 $O/bench.err; python tools/bench_summary.py $O/bench.json
    ;;
  quick)      # the GPU suite without the full-size configs
    step gpu_tests 600 python -m pytest tests -m gpu -x -q --durations=10 --deselect tests/test_full_size.py ${PYTEST_ARGS:-}
    tail -16 $O/gpu_tests.log
    ;;
  prof100a|prof100b|prof10)   # rocprofv3 passes of `bench.py --mode device` on the build that ships (what profiles/pmc_traffic.json is made from):
              # prof100a = configs[2] kernel trace + FETCH + WRITE, prof100b = configs[2] SQ passes A / B, prof10 = all five on configs[1].
              # PMC passes carry only --kernel-trace (gpurun refuses counters combined with the other trace domains).
    SQA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"
    SQB="SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"
    pass() {   # name, timeout, rocprofv3 options ..., -- command
      local name=$1 to=$2; shift 2
      local t0=$(date +%s)
      timeout -k 10 $to rocprofv3 --kernel-trace --output-format csv -d $O/$name -o $name "$@" > $O/$name.log 2>&1
      echo "$name rc=$? $(( $(date +%s) - t0 )) s" >> $O/steps.txt
    }
    DEV100="python bench.py --mode device --device-steps 2 --no-cpu-baseline"
    DEV10="python bench.py --mode device --config r9_10k_8kb --device-steps 3 --no-cpu-baseline --arena-gib 40"
    if [ $STAGE = prof100a ]; then
      pass kt100k 330 --stats -- $DEV100
      pass pmc_fetch100k 330 --pmc FETCH_SIZE -- $DEV100
      pass pmc_write100k 330 --pmc WRITE_SIZE -- $DEV100
    elif [ $STAGE = prof100b ]; then
      pass pmc_sqa100k 330 --pmc $SQA -- $DEV100
      pass pmc_sqb100k 330 --pmc $SQB GRBM_GUI_ACTIVE -- $DEV100
    else
      pass kt10k 150 --stats -- $DEV10
      pass pmc_sqa10k 150 --pmc $SQA -- $DEV10
      pass pmc_sqb10k 150 --pmc $SQB GRBM_GUI_ACTIVE -- $DEV10
      pass pmc_fetch10k 150 --pmc FETCH_SIZE -- $DEV10
      pass pmc_write10k 150 --pmc WRITE_SIZE -- $DEV10
    fi
    find $O -name "*kernel_trace.csv" -size +20M -delete
    find $O -name "*.csv" | head -40
    ;;
  fusedtrace) # the fused align + scaling_single HOST call under rocprofv3: kernel timeline (8 chunks in flight) and one SQ pass
    step ft_gen 200 python tools/fused_trace.py 20000 /tmp/ft
    step fused_kt 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fused_kt -o fused -- python tools/fused_trace.py 20000 /tmp/ft
    grep "rep" $O/fused_kt.log
    step fused_sq 200 rocprofv3 --kernel-trace --output-format csv -d $O/fused_sq -o fused --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -- python tools/fused_trace.py 20000 /tmp/ft
    find $O -name "*.csv" | head
    ;;
  cmd)        # an ad-hoc command line (quoted by the caller) under a timeout
    step cmd ${CMD_TIMEOUT:-600} bash -c "$CMD"
    tail -40 $O/cmd.log
    ;;
esac
cat $O/steps.txt 2>/dev/null; true
