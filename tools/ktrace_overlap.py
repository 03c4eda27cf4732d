import csv,sys,glob
f=glob.glob("/root/repo/gpurun_out/hp/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "abea_align_kernel" in r["Kernel_Name"] or "abea_pre" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
for r in rows[:24]:
    print(r["Kernel_Name"][:20], r["Queue_Id"], round((int(r["Start_Timestamp"])-t0)/1e6,2), round((int(r["End_Timestamp"])-t0)/1e6,2))
