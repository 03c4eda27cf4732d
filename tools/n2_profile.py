#!/usr/bin/env python3
"""Event detection (row N2) for a rocprofv3 kernel trace: DNA twice, RNA twice, nothing else — no generator pool (a forked or
spawned pool under the profiler has hung two GPU calls), no oracle, no alignment.  usage: n2_profile.py [n_reads]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5c_amd import abea, synth, load_model_f32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
b = synth.make_batch(n, model, k, seed=20250002, law="gamma8k", workers=1)
sigs, sc = synth.make_signals(b, seed=1)
seqs = [b["reads"][int(b["read_ptr"][i]):int(b["read_ptr"][i]) + int(b["read_len"][i])].tobytes() for i in range(n)]
ns = sum(len(s) for s in sigs)
ctx = abea.AbeaContext(model, k, mem_frac=0.5)
for rna in (False, False, True, True):
    evs, ne, scal = ctx.detect_events_device(sigs, sc, seqs=seqs, rna=rna)
    ms = ctx.stats()["event_ms"]
    print(f"{'RNA' if rna else 'DNA'} parameters: {n} reads, samples={ns} ({ns/1e6:.1f} Msamples), {int(ne.sum())/1e6:.2f} Mevents: kernels {ms:.2f} ms = "
          f"{ns/ms/1e3:.1f} Msamples/s, {ne.sum()/ms/1e3:.1f} Mevents/s", flush=True)
ctx.close()
