#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): per-kernel time of the device-resident command, the PMC passes
# that feed profiles/pmc_traffic.json, and the per-kernel time of the default bench command.  Output: gpurun_out/r02/.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
DEV10="python bench.py --mode device --config r9_10k_8kb --device-steps 3 --no-cpu-baseline --arena-gib 40 --batch-cache /tmp/bc"
$DEV10 > $O/dev10k.json 2> $O/dev10k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt10k -o kt -- $DEV10 > $O/kt10k.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq10k -o pmc -- $DEV10 > $O/pmc_sq10k.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch10k -o pmc -- $DEV10 > $O/pmc_fetch10k.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write10k -o pmc -- $DEV10 > $O/pmc_write10k.log 2>&1
if [ "$1" = "full" ]; then
  DEV100="python bench.py --mode device --device-steps 2 --no-cpu-baseline"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch100k -o pmc -- $DEV100 > $O/pmc_fetch100k.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write100k -o pmc -- $DEV100 > $O/pmc_write100k.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt100k_dev -o kt -- $DEV100 > $O/kt100k_dev.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt100k_default -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt100k_default.log 2>&1
fi
find $O -name "*.csv" | head -40
# keep the merged output small: per-dispatch kernel traces of the PMC runs are not needed
find $O -name "*kernel_trace.csv" -size +20M -delete
