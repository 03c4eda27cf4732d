#!/bin/bash
# windowed trace read-back at radius 8 / 12 against the shipped build (radius 26 = whole group): time, reloads, FETCH
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03rad; mkdir -p $O; : > $O/rad.log
for rep in 1 2; do for v in ship r12 r8; do
  L=build/libabea_$v.so; [ $v = ship ] && L=f5c_amd/libabea_hip.so
  ABEA_LIB_PATH=$L timeout 200 python tools/ab_compare.py run /tmp/$v.npz 2>/dev/null | grep "kernel ms" | sed "s/^/$v /" >> $O/rad.log
done; done
for v in r12 r8; do timeout 60 python tools/ab_compare.py compare /tmp/ship.npz /tmp/$v.npz >> $O/rad.log 2>&1; ABEA_LIB_PATH=build/libabea_$v.so timeout 120 python tools/walk_stats.py 2>/dev/null | sed "s/^/$v /" >> $O/rad.log; done
DEV10="python bench.py --mode device --config r9_10k_8kb --device-steps 3 --no-cpu-baseline --arena-gib 40 --batch-cache /tmp/bc"
timeout 200 $DEV10 > /dev/null 2>&1
for v in r12 r8; do ABEA_LIB_PATH=build/libabea_$v.so timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -o pmc -- $DEV10 > /dev/null 2>&1
python - <<PY >> $O/rad.log
import csv
a=[float(r["Counter_Value"]) for r in csv.DictReader(open("$O/fetch_$v/pmc_counter_collection.csv")) if r["Kernel_Name"].startswith("abea_align")]
print("$v FETCH_SIZE x2 B/event:", sum(a)/len(a)*2048/158727291)
PY
done
cat $O/rad.log
