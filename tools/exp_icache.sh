# round-6 experiment: instruction-cache counters of abea_align_kernel, alignment-only launches against fused (scaling_single) launches
# of the same batch (tools/fused_trace.py: 3 host calls with pairs, then 3 fused), to test whether the fused launch's longer FILL
# wave time (profiles/r06/e_phase_profile_*.log) is instruction-cache pressure.  gpurun -- 'bash tools/exp_icache.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h_icache; mkdir -p $O
timeout -k 10 200 python tools/fused_trace.py 20000 /tmp/ft > $O/gen.log 2>&1; echo "gen rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ic -o ic --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_BUSY_CYCLES -- python tools/fused_trace.py 20000 /tmp/ft > $O/ic.log 2>&1
echo "ic rc=$?"; grep "rep" $O/ic.log
python3 - <<PY
import csv, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open("$O/ic/ic_counter_collection.csv")):
    if r["Kernel_Name"].startswith("abea_align_kernel"):
        d = rows[int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"]); d["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
ids = sorted(rows)
half = len(ids) // 2
for name, sel in (("alignment-only launches", ids[:half]), ("fused launches", ids[half:])):
    acc = collections.defaultdict(float)
    for i in sel:
        for k, v in rows[i].items(): acc[k] += v
    print(name, len(sel), {k: round(v) for k, v in acc.items()}, "miss rate %.4f" % (acc["SQC_ICACHE_MISSES"] / max(1, acc["SQC_ICACHE_REQ"])))
PY
