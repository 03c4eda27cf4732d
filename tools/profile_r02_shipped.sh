#!/bin/bash
# Per-kernel time and SQ counters of the SHIPPED round-2 build (after the k-mer quad rotation), device-resident command.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
DEV10="python bench.py --mode device --config r9_10k_8kb --device-steps 3 --no-cpu-baseline --arena-gib 40 --batch-cache /tmp/bc"
timeout 120 $DEV10 > $O/dev10k.json 2> $O/dev10k.err
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt10k -o kt -- $DEV10 > $O/kt10k.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq10k -o pmc -- $DEV10 > $O/pmc_sq10k.log 2>&1
cat $O/kt10k/kt_kernel_stats.csv | head -3
