#!/usr/bin/env python3
"""Experiment helper: per-read phase timeline of abea_align_kernel (needs the ABEA_PROFILE_PHASES build:
   ABEA_LIB_PATH=build/libabea_prof.so python tools/phase_profile.py [--reads N] [--config C])."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f5c_amd import abea, synth, load_model_f32
ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=10000); ap.add_argument("--config", default="r9_10k_8kb")
ap.add_argument("--scaling", action="store_true", help="fused scaling_single on: reports phase 4's share of the wave time (diag.pad of the profile build)")
a = ap.parse_args()
cfg = synth.CONFIGS[a.config]
k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
b = synth.make_batch(a.reads, model, k, seed=cfg["seed"], law=cfg["law"], workers=32)
d = abea.AbeaContext.upload(b)
ctx = abea.AbeaContext(model, k)
ctx.align_db_device(d, scaling=a.scaling); ctx.align_db_device(d, scaling=a.scaling)
st = ctx.stats()
_, n_pairs, dg = ctx.download(d)
ok = dg["n_aligned"] > 0
t0 = dg["sum_emission"][ok]; fill = dg["best_event"][ok].astype(np.float64); walk = dg["max_gap"][ok].astype(np.float64); exp = dg["spanned"][ok].astype(np.float64)
bands = (b["n_events"] + b["read_len"] - k + 3)[ok].astype(np.float64); steps = dg["n_aligned"][ok].astype(np.float64)
base = t0.min(); end = (t0 + fill + walk + exp)
tick = 10e-9
print(f"kernel {st['fill_ms']:.2f} ms; span by timestamps {(end.max()-base)*tick*1e3:.2f} ms; reads {ok.sum()}")
print(f"sum fill {fill.sum()*tick*1e3:.1f} ms-wave, walk {walk.sum()*tick*1e3:.1f}, expand {exp.sum()*tick*1e3:.1f}")
if a.scaling:
    sel = n_pairs[ok] > 0
    p4 = (dg["pad"][ok].astype(np.float64) - exp)[sel]
    kk = (b["read_len"] - k + 1)[ok][sel].astype(np.float64)
    tot = fill.sum() + walk.sum() + exp.sum() + p4.sum()
    print(f"phase 4 (scaling_single): sum {p4.sum()*tick*1e3:.1f} ms-wave = {100*p4.sum()/tot:.2f} % of the wave time; per k-mer median {np.median(p4/kk)*10:.1f} ns, "
          f"per 64 k-mers {np.median(p4/kk)*640:.0f} ns")
print(f"per band: median {np.median(fill/bands)*10:.1f} ns, p10 {np.percentile(fill/bands,10)*10:.1f}, p90 {np.percentile(fill/bands,90)*10:.1f}")
print(f"per step: median {np.median(walk/steps)*10:.1f} ns, p10 {np.percentile(walk/steps,10)*10:.1f}, p90 {np.percentile(walk/steps,90)*10:.1f}")
print(f"expand per step: median {np.median(exp/steps)*10:.1f} ns")
T = (end.max() - base)
for q in range(10):
    lo, hi = base + T * q / 10, base + T * (q + 1) / 10
    mid = (lo + hi) / 2
    nf = ((t0 <= mid) & (t0 + fill > mid)).sum(); nw = ((t0 + fill <= mid) & (t0 + fill + walk > mid)).sum()
    ne = ((t0 + fill + walk <= mid) & (end > mid)).sum()
    print(f"  t={q*10+5:3d}%: waves filling {nf:5d} walking {nw:5d} expanding {ne:5d}")
i = np.argmax(bands)
print(f"longest read: {bands[i]:.0f} bands fill {fill[i]*tick*1e3:.2f} ms ({fill[i]/bands[i]*10:.0f} ns/band) walk {walk[i]*tick*1e3:.2f} ms ({walk[i]/steps[i]*10:.0f} ns/step) start {(t0[i]-base)*tick*1e3:.2f} end {(end[i]-base)*tick*1e3:.2f}")
order = np.argsort(-(end))[:12]
idx_ok = np.nonzero(ok)[0]
print("last finishing reads: (bands, fill ms, ns/band, walk ms, start ms, end ms, qc_pass, flagged_bad)")
for j in order:
    r = idx_ok[j]
    print(f"  {bands[j]:.0f} {fill[j]*tick*1e3:.2f} {fill[j]/bands[j]*10:.0f} {walk[j]*tick*1e3:.2f} {(t0[j]-base)*tick*1e3:.2f} {(end[j]-base)*tick*1e3:.2f} {int(n_pairs[r]>0)} {int(b['bad'][r])}")
