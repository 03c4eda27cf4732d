#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (abea_align_batch_host: flatten + H2D + kernels + D2H +
un-flatten) and the cost of the optional scaling kernel; numbers quoted in DESIGN.md §6."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5c_amd import abea, synth, load_model_f32
k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
b = synth.make_batch(n, model, k, seed=20250002, law="gamma8k", workers=16)
ev = int(b["n_events"].sum())
ctx = abea.AbeaContext(model, k)
seqs, evs = [], []
for i in range(n):
    s = int(b["read_ptr"][i]); L = int(b["read_len"][i]); seqs.append(b["reads"][s:s + L].tobytes())
    s = int(b["event_ptr"][i]); E = int(b["n_events"][i]); evs.append(b["events"][s:s + E])
for rep in range(3):
    t0 = time.perf_counter(); ctx.align_db_host(seqs, evs, b["scalings"], want_diag=False); t = time.perf_counter() - t0
    st = ctx.stats()
    print(f"host API rep {rep}: {ev/st['total_ms']/1e3:.1f} Mevents/s inside the call (total {st['total_ms']:.1f} ms: h2d {st['h2d_ms']:.1f} "
          f"d2h {st['d2h_ms']:.1f} host flatten/unflatten {st['host_ms']:.1f} kernels {st['pre_ms']+st['fill_ms']:.1f}); "
          f"python wrapper wall {t*1e3:.0f} ms")
d = abea.AbeaContext.upload(b)
for sc in (False, True):
    ctx.align_db_device(d, want_diag=False, scaling=sc); ctx.align_db_device(d, want_diag=False, scaling=sc)
    st = ctx.stats()
    print(f"device API scaling={sc}: align {st['fill_ms']:.2f} ms, pre {st['pre_ms']:.2f} ms, scaling kernel {st['trace_ms']:.2f} ms, call {st['total_ms']:.2f} ms")
