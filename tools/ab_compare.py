#!/usr/bin/env python3
"""Experiment helper: run the same batch through two builds of the library (ABEA_LIB_PATH=a vs b, one per
process) and compare every output bit:  python tools/ab_compare.py run out.npz [config] [reads];
python tools/ab_compare.py compare a.npz b.npz"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from f5c_amd import abea, synth, load_model_f32
    cfg = synth.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "r9_10k_8kb"]
    k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
    b = synth.make_batch(int(sys.argv[4]) if len(sys.argv) > 4 else cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=32)
    d = abea.AbeaContext.upload(b)
    ctx = abea.AbeaContext(model, k)
    ctx.align_db_device(d); ctx.align_db_device(d)
    print(sys.argv[2], "kernel ms", ctx.stats()["fill_ms"])
    pairs, n_pairs, dg = ctx.download(d)
    np.savez(sys.argv[2], pairs=pairs.view(np.int32), n_pairs=n_pairs, sum_emission=dg["sum_emission"], max_score=dg["max_score"], pair_ptr=b["pair_ptr"])
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    assert (a["n_pairs"] == b["n_pairs"]).all(), "n_pairs differ"
    pa, pb = a["pairs"].reshape(-1, 2), b["pairs"].reshape(-1, 2)
    bad = 0
    for i, (s, n) in enumerate(zip(a["pair_ptr"], a["n_pairs"])):
        if not (pa[s:s + n] == pb[s:s + n]).all(): bad += 1
    assert bad == 0, f"{bad} reads differ"
    assert (a["sum_emission"] == b["sum_emission"]).all() and (a["max_score"] == b["max_score"]).all()
    print(f"identical: {len(a['n_pairs'])} reads, {int(a['n_pairs'].sum())} pairs, {(a['n_pairs']>0).sum()} pass QC")
