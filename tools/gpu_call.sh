#!/bin/bash
# Round-5 GPU calls, one parameterised script (replaces the per-call tools/r04_call_*.sh one-offs):
#   gpurun --timeout T -- 'bash tools/gpu_call.sh <stage> [tag]'
# Every step runs under its own `timeout -k 10` (TERM to the process group, KILL 10 s later: a hung kernel or profiler
# must not hold the box).  Output: gpurun_out/<tag>/ ; what is to be judged is copied into profiles/r05/ afterwards.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
STAGE=${1:-sweep}; TAG=${2:-r05_$STAGE}
O=gpurun_out/$TAG; mkdir -p $O
step() {   # name, timeout, command ...
  local name=$1 to=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $to "$@" > $O/$name.log 2>&1
  echo "$name rc=$? $(( $(date +%s) - t0 )) s" >> $O/steps.txt
}
case $STAGE in
  sweep)      # host-entry settings on ONE generated batch + the flatten loop alone + the changed host code against the oracle
    step t_host 400 python -m pytest tests/test_host_pipeline.py -m gpu -x -q -k "not torchrun and not two_ranks"
    tail -3 $O/t_host.log
    step sweep 700 python tools/host_sweep.py --out $O ${SWEEP_ARGS:-}
    grep -v "^first-touch\|^interleave" $O/sweep.log | tail -40
    ;;
  hwq)        # the same sweep under GPU_MAX_HW_QUEUES = 8 / 16 / 24 (read by the HIP runtime at initialisation: one process each)
    step t_changed 400 python -m pytest tests/test_host_pipeline.py tests/test_process_chain.py tests/test_rna_events.py -m gpu -x -q -k "not torchrun and not two_ranks"
    tail -3 $O/t_changed.log
    for q in ${HWQS:-8 16 24}; do
      GPU_MAX_HW_QUEUES=$q step sweep_q$q 400 python tools/host_sweep.py --out $O/q$q --no-flatten-probe --steps 3 --only ${ONLY:-base,slots16,chunk24M_slots16,chunk96M,chunk96M_slots16,rmin512,rmin512_slots16,fused_base,fused_slots16,fused_rmin512_slots16,base_again}
      grep '"name"' $O/sweep_q$q.log | python3 -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('q$q %-24s %7.1f ms  flat %6.1f unfl %5.1f wait %6.1f plan %4.1f  fill_sum %7.1f' % (r['name'], r['ms_per_step'], r['flatten_ms'], r['unflatten_ms'], r['wait_ms'], r['plan_ms'], r['fill_ms']))"
    done
    ;;
  chain)      # round-5 rewrite of the raw-signal entries + the ramp launch order + where GPU_MAX_HW_QUEUES is set
    step t_changed 500 python -m pytest tests/test_host_pipeline.py tests/test_process_chain.py tests/test_rna_events.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not torchrun and not two_ranks" --durations=5
    tail -9 $O/t_changed.log
    step bench10k 400 python bench.py --config r9_10k_8kb --steps 3 --warmup 1
    tail -c 2500 $O/bench10k.log
    step sweep 400 python tools/host_sweep.py --out $O/py --no-flatten-probe --steps 3 --only base,lpt,fused_base,fused_lpt,base_again
    ABEA_KEEP_HW_QUEUES=1 step sweep_lib 400 python tools/host_sweep.py --out $O/lib --no-flatten-probe --steps 3 --only base
    for f in sweep sweep_lib; do grep '"name"\|GPU_MAX' $O/$f.log | python3 -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln)
    if 'name' not in r: print('$f', r); continue
    print('$f %-12s %7.1f ms  flat %6.1f unfl %5.1f wait %6.1f plan %4.1f  gpu_busy %7.1f' % (r['name'], r['ms_per_step'], r['flatten_ms'], r['unflatten_ms'], r['wait_ms'], r['plan_ms'], r.get('gpu_busy_ms', 0)))"; done
    ;;
  exp)        # round-5 experiments: where GPU_MAX_HW_QUEUES takes effect for a C++ caller, the method-of-moments kernel, the Markstein quotient
    step t_n2 500 python -m pytest tests/test_rna_events.py tests/test_process_chain.py tests/test_oracle_ecoli.py tests/test_fuzz_gpu.py -m gpu -x -q -k "not alignment_and_scaling" --durations=5
    tail -8 $O/t_n2.log
    step ab_markstein 300 python tools/ab_quick.py ship=f5c_amd/libabea_hip.so markstein=build/libabea_markstein.so ship2=f5c_amd/libabea_hip.so markstein2=build/libabea_markstein.so --config r9_10k_8kb --launches 6
    grep "kernel ms" $O/ab_markstein.log
    step n2prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n2 -o n2 -- python tools/n2_profile.py 2048
    grep "parameters" $O/n2prof.log; find $O/n2 -name "*kernel_stats.csv" | head -2
    g++ -std=c++11 -O2 tests/shim_driver.cpp -o /tmp/shim_driver -Lf5c_amd -labea_hip -Wl,-rpath,$GRAFT_REPO_ROOT/f5c_amd
    step dump 200 python tools/probe/dump_batch.py r9_10k_8kb /tmp/b10k.bin
    for q in lib 4 16; do
      if [ $q = lib ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      SHIM_REPS=6 SHIM_NOPRINT=1 timeout -k 10 200 /tmp/shim_driver /tmp/b10k.bin /dev/null 2> $O/shim_q$q.log
      echo "C++ caller, GPU_MAX_HW_QUEUES=$q: $(grep wall $O/shim_q$q.log | awk '{print $4}' | tr '\n' ' ')"
    done | tee $O/hw_queues_cpp_caller.txt
    unset GPU_MAX_HW_QUEUES
    ;;
  exp2)       # the two-wave-per-read model of a band (tools/ubench/two_wave_band.hip) + the method-of-moments kernel again
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/ubench/two_wave_band.hip -o /tmp/two_wave_band && step two_wave 120 /tmp/two_wave_band
    cat $O/two_wave.log
    step t_n2 300 python -m pytest tests/test_rna_events.py tests/test_process_chain.py tests/test_oracle_ecoli.py -m gpu -x -q
    tail -3 $O/t_n2.log
    step n2prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/n2 -o n2 -- python tools/n2_profile.py 2048
    grep "parameters" $O/n2prof.log; grep "scalings\|copyBuffer" $O/n2/n2_kernel_stats.csv | cut -d, -f1-6
    ;;
  tests)      # the whole GPU suite + the bench line of the build that ships
    step gpu_tests 900 python -m pytest tests -m gpu -x -q --durations=10
    tail -16 $O/gpu_tests.log
    t0=$(date +%s); timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s" >> $O/steps.txt
    python tools/bench_summary.py $O/bench.json
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
  phase4)     # A/B of the fused scaling_single phase (device-resident launches, one process) + the GPU suite without the full-size configs
    step ab_phase4 400 python tools/ab_quick.py r04before=build/libabea_r05_before_phase4_prefetch.so r04after=f5c_amd/libabea_hip.so r04before2=build/libabea_r05_before_phase4_prefetch.so r04after2=f5c_amd/libabea_hip.so --config r9_10k_8kb --launches 4 --scaling-launches 5
    grep "kernel ms" $O/ab_phase4.log
    step gpu_tests 600 python -m pytest tests -m gpu -x -q --durations=6 --deselect tests/test_full_size.py
    tail -12 $O/gpu_tests.log
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
cat $O/steps.txt
