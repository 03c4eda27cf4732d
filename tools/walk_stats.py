#!/usr/bin/env python3
"""How often the traceback walk leaves the prefetched window of a trace group (abea_read_diag.pad = groups re-loaded
whole) on a synthetic config:  python tools/walk_stats.py [config] [reads]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5c_amd import abea, synth, load_model_f32
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "r9_10k_8kb"]
k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
b = synth.make_batch(int(sys.argv[2]) if len(sys.argv) > 2 else cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=16)
d = abea.AbeaContext.upload(b)
ctx = abea.AbeaContext(model, k)
ctx.align_db_device(d); ctx.align_db_device(d)
_, n_pairs, dg = ctx.download(d)
ok = dg["n_aligned"] > 0
groups = (dg["n_aligned"][ok].astype(np.float64) * 1.5 / 32)          # ~1.5 bands per walk step
rl = dg["pad"][ok].astype(np.float64)
print(f"kernel {ctx.stats()['fill_ms']:.2f} ms; reads walked {ok.sum()}; groups ~{groups.sum():.0f}; whole-group reloads {rl.sum():.0f} "
      f"({100 * rl.sum() / groups.sum():.3f} % of groups); reads with any reload {int((rl > 0).sum())}; max per read {rl.max():.0f}")
good = ok & (n_pairs > 0)
print(f"QC-pass reads: reloads per 1000 groups {1000 * dg['pad'][good].sum() / (dg['n_aligned'][good].sum() * 1.5 / 32):.3f}; "
      f"QC-fail reads: {1000 * dg['pad'][ok & (n_pairs == 0)].sum() / max(1.0, dg['n_aligned'][ok & (n_pairs == 0)].sum() * 1.5 / 32):.3f}")
