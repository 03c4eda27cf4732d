#!/bin/bash
# Round-4 GPU call A: the new code only, each step under its own short timeout (a hung kernel must not hold the box).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
free -g | head -2 > $O/sys.txt; rocm-smi --showmeminfo vram >> $O/sys.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scaling" > $O/t1_scaling.log 2>&1; echo "t1 rc=$?" >> $O/steps.txt; tail -2 $O/t1_scaling.log
timeout 240 python -m pytest tests/test_oracle_ecoli.py tests/test_host_pipeline.py -m gpu -x -q -k "recalibrated or 111_reads or fused" > $O/t2_fused.log 2>&1; echo "t2 rc=$?" >> $O/steps.txt; tail -2 $O/t2_fused.log
timeout 240 python -m pytest tests/test_process_chain.py -m gpu -x -q > $O/t3_process.log 2>&1; echo "t3 rc=$?" >> $O/steps.txt; tail -4 $O/t3_process.log
timeout 200 python -m pytest tests/test_hmm_gpu.py tests/test_hmm_pin.py -m gpu -x -q > $O/t4_hmm.log 2>&1; echo "t4 rc=$?" >> $O/steps.txt; tail -2 $O/t4_hmm.log
timeout 200 python -m pytest tests/test_host_pipeline.py -m gpu -x -q -k "torchrun or two_ranks" > $O/t5_dist.log 2>&1; echo "t5 rc=$?" >> $O/steps.txt; tail -3 $O/t5_dist.log
dmesg 2>/dev/null | tail -5 >> $O/sys.txt
cat $O/steps.txt
