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
# --scaling-launches N: after the plain launches of a build whose name starts with "r04" (the in-kernel scaling_single of round 4),
# N more device-resident launches with scaling_single fused: what the last phase of abea_align_kernel costs on the GPU alone
scal_launches = int(sys.argv[sys.argv.index("--scaling-launches") + 1]) if "--scaling-launches" in sys.argv else 0
cfg = dict(synth.CONFIGS[cfg_name])
if "--reads" in sys.argv:
    cfg["n_reads"] = int(sys.argv[sys.argv.index("--reads") + 1])
k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
b = synth.make_batch(cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=32)
d = abea.AbeaContext.upload(b)
# host memory: the 100k-read batch is 60 GB of event tables + 30 GB of pair buffers; a first version also built two more
# 21-GB copies of the pairs per build and took the box down.  After the upload only the index arrays are needed here, and the
# pair lists are compared through the blocked per-read hash of tests/pairhash.py.
pair_ptr = b["pair_ptr"].copy()
for key in ("events", "reads"):
    b[key] = None
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pairhash import hash_pair_lists
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
    out["pairs_sha"] = hashlib.sha256(hash_pair_lists(pairs, pair_ptr, n_pairs).tobytes()).hexdigest()   # only defined pairs
    del pairs
    if scal_launches and name.startswith("r04"):
        sm = []
        for _ in range(scal_launches):
            ctx.align_db_device(d, want_diag=False, scaling=True); sm.append(ctx.stats()["fill_ms"])
        cal = int(((d["read_stat_flag"].cpu().numpy()[:len(n_pairs)] & 1) == 0).sum())
        print(f"{name} with scaling_single fused, kernel ms " + " ".join(f"{x:.3f}" for x in sm) + f" | {cal} reads calibrated", flush=True)
        for key in ("b2e", "scalings_io", "events_per_base", "read_stat_flag", "n_event_alignment"):
            d.pop(key, None)
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
