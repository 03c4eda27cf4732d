#!/usr/bin/env python3
"""Experiment helper: ONE process, one synthetic batch, several builds of the library (name=path ...): per build the
abea_align_kernel time of a few launches, and every output bit compared with the first build's.
  python tools/ab_quick.py ship=f5c_amd/libabea_hip.so pk=build/libabea_pk.so [--config r9_10k_8kb] [--reads N] [--launches 4]"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5c_amd import abea, synth, load_model_f32

args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
cfg_name = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "r9_10k_8kb"
launches = int(sys.argv[sys.argv.index("--launches") + 1]) if "--launches" in sys.argv else 4
cfg = dict(synth.CONFIGS[cfg_name])
if "--reads" in sys.argv:
    cfg["n_reads"] = int(sys.argv[sys.argv.index("--reads") + 1])
k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
b = synth.make_batch(cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=32)
d = abea.AbeaContext.upload(b)
ref = None
for a in args:
    name, path = a.split("=", 1)
    abea._LIB = None; abea.LIB_PATH = os.path.abspath(path)
    ctx = abea.AbeaContext(model, k)
    ms = []
    for _ in range(launches):
        ctx.align_db_device(d); ms.append(ctx.stats()["fill_ms"])
    pairs, n_pairs, dg = ctx.download(d)
    out = dict(n_pairs=n_pairs.copy(), sum_emission=dg["sum_emission"].copy(), max_score=dg["max_score"].copy())
    pv = pairs.view(np.int32).reshape(-1, 2)
    keep = np.zeros(len(pv), bool)                               # only the pairs of reads that passed QC are defined
    for s, n in zip(b["pair_ptr"], n_pairs):
        keep[s:s + n] = True
    out["pairs_sha"] = hashlib.sha256(np.ascontiguousarray(pv[keep]).tobytes()).hexdigest()
    del ctx
    line = f"{name} kernel ms " + " ".join(f"{x:.3f}" for x in ms) + f" | min {min(ms[1:]):.3f}"
    if ref is None:
        ref = out
        line += f" | {len(n_pairs)} reads, {int(n_pairs.sum())} pairs, {(n_pairs > 0).sum()} pass QC"
    else:
        same = ((out["n_pairs"] == ref["n_pairs"]).all() and out["pairs_sha"] == ref["pairs_sha"]
                and (out["sum_emission"] == ref["sum_emission"]).all() and (out["max_score"] == ref["max_score"]).all())
        line += " | outputs " + ("IDENTICAL" if same else "DIFFER")
    print(line, flush=True)
