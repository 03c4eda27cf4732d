#!/usr/bin/env python3
"""Throughput of the device event-detection kernel (row N2) next to the oracle's CPU getevents."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5c_amd import abea, synth, load_model_f32
from oracle import orc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
b = synth.make_batch(n, model, k, seed=20250002, law="gamma8k", workers=int(os.environ.get("ABEA_WORKERS", "16")))
sigs, sc = synth.make_signals(b, seed=1)
seqs = [b["reads"][int(b["read_ptr"][i]):int(b["read_ptr"][i]) + int(b["read_len"][i])].tobytes() for i in range(n)]
ns = sum(len(s) for s in sigs)
ctx = abea.AbeaContext(model, k, mem_frac=0.6)
RNA = os.environ.get("RNA", "0") == "1"      # the RNA parameter set (events.c:59-65) on the same signals: kernel times only
for rep in range(2):
    evs, ne, scal = ctx.detect_events_device(sigs, sc, seqs=seqs, rna=RNA)
    ms = ctx.stats()["event_ms"]
    print(f"device: {n} reads, {ns/1e6:.1f} Msamples, {int(ne.sum())/1e6:.2f} Mevents: kernel {ms:.2f} ms = {ns/ms/1e3:.1f} Msamples/s, {ne.sum()/ms/1e3:.1f} Mevents/s")
if RNA:
    sys.exit(0)
# whole device chain: raw signal -> events + scalings -> ABEA -> scaling_single, event tables resident in HBM
import torch
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = ctx.signals_to_device_batch(sigs, sc, seqs)
    t1 = time.perf_counter()
    ctx.align_db_device(d, want_diag=False, scaling=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = ctx.stats()
    print(f"chain: detect call {1e3*(t1-t0):.0f} ms wall (incl. python flatten + H2D of the signal), align+scaling kernels "
          f"{st['pre_ms']+st['fill_ms']+st['trace_ms']:.1f} ms (call {1e3*(t2-t1):.0f} ms): device time ~{ms + st['pre_ms']+st['fill_ms']+st['trace_ms']:.0f} ms "
          f"for {n} reads = {n/(ms + st['pre_ms']+st['fill_ms']+st['trace_ms'])*1e3:.0f} reads/s, {ns/(ms + st['pre_ms']+st['fill_ms']+st['trace_ms'])/1e3:.0f} Msamples/s")
t0 = time.perf_counter()
m = 0
for i in range(0, n, max(1, n // 16)):
    orc.getevents(sigs[i], *sc[i]); m += len(sigs[i])
t = time.perf_counter() - t0
print(f"oracle getevents, 1 thread: {m/t/1e6:.1f} Msamples/s")
