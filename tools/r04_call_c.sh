#!/bin/bash
# Round-4 GPU call C: N2 after the {S,Q} pair layout (parity, then the per-kernel profile, DNA and RNA), then the fused
# align + scaling_single call next to the align-only call on configs[2] with the per-kernel split.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 300 python -m pytest tests/test_oracle_ecoli.py tests/test_rna_events.py tests/test_fuzz_gpu.py tests/test_process_chain.py -m gpu -x -q > $O/t_n2.log 2>&1; echo "n2 tests rc=$?" >> $O/steps.txt; tail -2 $O/t_n2.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/n2_dna -o n2_dna -- python tools/event_rate.py 4096 > $O/n2_dna.log 2> $O/n2_dna.err; echo "n2 dna rc=$?" >> $O/steps.txt; cat $O/n2_dna.log
RNA=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/n2_rna -o n2_rna -- python tools/event_rate.py 4096 > $O/n2_rna.log 2> $O/n2_rna.err; echo "n2 rna rc=$?" >> $O/steps.txt; cat $O/n2_rna.log
free -g | head -2 >> $O/steps.txt
MODES=pairs,fused DEVICE=0 timeout 600 python tools/host_api_rate.py r9_100k_mixed 3 > $O/fused_rate.log 2> $O/fused_rate.err; echo "fused rc=$?" >> $O/steps.txt; cat $O/fused_rate.log
find $O -name "*kernel_stats.csv" | while read f; do echo $f; head -14 $f; done
cat $O/steps.txt
