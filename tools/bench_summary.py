#!/usr/bin/env python3
"""One-screen digest of a bench.py JSON line:  python tools/bench_summary.py gpurun_out/<tag>/bench.json"""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = j["host_to_host"]
print("value", j["value"], "ms/step", j["ms_per_step"], "bound", j["bound"], h["host_ms_per_step"], "gpu idle", h["gpu_idle_frac"])
r = j["roofline"]
print("device_resident", j["device_resident"]["mevents_per_s"], "kernel_only", j["kernel_only"]["mevents_per_s"], j["kernel_only"]["ms_per_step"],
      "roofline", r["frac"], "traffic", r["traffic"], "stale", (r.get("limiter") or {}).get("stale"))
f = j.get("fused_scaling")
if f:
    print("fused", f["mevents_per_s"], f["ms_per_step"], f["host_ms_per_step"], "gpu busy", f["gpu_busy_ms_per_step"])
p = j.get("process_chain")
if p:
    print("chain", {k: p[k] for k in ("reads", "ms_per_call", "msamples_per_s", "mevents_per_s", "gpu_idle_frac", "host_ms_per_call", "kernels_ms_sum_over_chunks",
                                      "pcie_bytes_per_call", "gpu_bit_exact_on_cpu_sample") if k in p})
    print("event_db", p["event_db_alone"])
s = j.get("f5c_default_batch")
if s:
    print("small", s["mevents_per_s"], s["in_flight"]["lanes"])
c = j.get("cpu_baseline")
if c:
    print("cpu", c["value"], c["cores"], c["gpu_bit_exact_on_sample"])
print("throttle", j.get("cgroup_cpu", {}).get("in_timed_region"), "hwq", j.get("hip_runtime", {}).get("GPU_MAX_HW_QUEUES"), "gen_s", j.get("gen_s"))
