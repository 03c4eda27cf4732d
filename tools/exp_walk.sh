#!/bin/bash
# A/B of the windowed traceback walk + masked trace store against the previous build (build/ab_old/libabea_old.so)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03walk; mkdir -p $O
ABEA_LIB_PATH=build/ab_old/libabea_old.so timeout 300 python tools/ab_compare.py run /tmp/old.npz > $O/ab.log 2>&1
timeout 300 python tools/ab_compare.py run /tmp/new.npz >> $O/ab.log 2>&1
timeout 100 python tools/ab_compare.py compare /tmp/old.npz /tmp/new.npz >> $O/ab.log 2>&1
timeout 300 python tools/walk_stats.py >> $O/ab.log 2>&1
cat $O/ab.log
DEV10="python bench.py --mode device --config r9_10k_8kb --device-steps 3 --no-cpu-baseline --arena-gib 40 --batch-cache /tmp/bc"
timeout 200 $DEV10 > $O/dev10k.json 2> $O/dev10k.err
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch10k -o pmc -- $DEV10 > $O/pmc_fetch10k.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write10k -o pmc -- $DEV10 > $O/pmc_write10k.log 2>&1
python - <<'PY'
import csv, collections
for f in ["pmc_fetch10k","pmc_write10k"]:
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(f"gpurun_out/r03walk/{f}/pmc_counter_collection.csv")):
        if "align" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
    for c in acc: print(f, c, acc[c]/cnt[c], "KB per launch;", acc[c]/cnt[c]*1024/158727291, "B/event (x2 for FETCH)")
PY
