#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (abea_align_batch_host: flatten + H2D + kernels + D2H +
un-flatten) next to the device-resident entry; numbers quoted in DESIGN.md §6.
usage: host_api_rate.py [config|n_reads] [reps]   env: ABEA_HOST_* knobs (see abea_host.cpp)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5c_amd import abea, synth, load_model_f32, synthetic_model
arg = sys.argv[1] if len(sys.argv) > 1 else "2000"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if arg in synth.CONFIGS:
    cfg = synth.CONFIGS[arg]
else:
    cfg = dict(n_reads=int(arg), seed=20250002, law="gamma8k", k=6)
k = cfg["k"]
model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))[1] if k == 6 else synthetic_model(k, seed=9)
t0 = time.time()
b = synth.make_batch(cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=16)
ev = int(b["n_events"].sum())
print(f"{arg}: {len(b['read_len'])} reads, {ev/1e6:.1f} M events, generated in {time.time()-t0:.1f} s", flush=True)
ctx = abea.AbeaContext(model, k, max_arena_bytes=int(float(os.environ.get("ARENA_GIB", "0")) * (1 << 30)))
modes = os.environ.get("MODES", "pairs,pairs-device,fused,fused+pairs").split(",")
for mode in modes:
    if mode == "pairs-device":
        os.environ["ABEA_HOST_PAIRS"] = "device"
    else:
        os.environ.pop("ABEA_HOST_PAIRS", None)
    v = ctx.host_view(b, scaling=mode.startswith("fused"), want_pairs=(mode != "fused"))
    for rep in range(reps):
        t0 = time.perf_counter(); ctx.align_view(v); t = time.perf_counter() - t0
        st = ctx.stats()
        print(f"host {mode:13s} rep {rep}: {ev/t/1e6:8.1f} Mevents/s  wall {t*1e3:7.1f} ms | flatten {st['flatten_ms']:6.1f} unflatten {st['unflatten_ms']:6.1f} "
              f"wait {st['wait_ms']:6.1f} | kernels(sum over chunks) pre {st['pre_ms']:.1f} align {st['fill_ms']:.1f} scaling {st['trace_ms']:.1f} | "
              f"{st['n_sub_batches']} chunks, h2d {st['h2d_bytes']/1e9:.2f} GB d2h {st['d2h_bytes']/1e9:.2f} GB, {st['host_threads']} threads", flush=True)
    del v
if os.environ.get("DEVICE", "1") == "1":
    d = abea.AbeaContext.upload(b)
    for sc in (False, True):
        ctx.align_db_device(d, want_diag=False, scaling=sc); ctx.align_db_device(d, want_diag=False, scaling=sc)
        st = ctx.stats()
        print(f"device API scaling={sc}: {ev/st['total_ms']/1e3:.1f} Mevents/s; align {st['fill_ms']:.2f} ms, pre {st['pre_ms']:.2f} ms, scaling kernel {st['trace_ms']:.2f} ms, call {st['total_ms']:.2f} ms")
