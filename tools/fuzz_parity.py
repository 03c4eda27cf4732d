#!/usr/bin/env python3
"""Randomised parity sweep (GPU): many small batches with extreme shapes — reads of k..k+5 bases, 1-event reads,
events-per-base ratios from 0.2 to just under / over the 15.0 guard, repeated and sub-sampled event tables, odd
scalings, constant signals — through both entry points (and the device scaling_single, row N1) against the CPU oracle.  tests/test_fuzz_gpu.py runs a seeded, time-boxed slice of it
under -m gpu; for a long sweep run:  python tools/fuzz_parity.py [seconds] [seed] [k9]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5c_amd import abea, synth, load_model_f32
from f5c_amd.types import EVENT_DT
from oracle import orc



def run(budget=60.0, seed=1, k9=False, ctx=None, max_batches=None):
    """Fuzz for `budget` seconds (or max_batches); returns (batches, reads, reads passing QC).  Raises on any mismatch."""
    k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
    if k9:                                                           # R10-style 9-mer table (synthetic, BASELINE configs[4])
        from f5c_amd import synthetic_model
        k, model = 9, synthetic_model(9, seed=9)
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        ctx = abea.AbeaContext(model, k, max_arena_bytes=2 << 30)
    t0 = time.time()
    n_batches = n_reads = n_pass = 0
    while time.time() - t0 < budget and (max_batches is None or n_batches < max_batches):
        n = int(rng.integers(1, 33))
        kinds = rng.integers(0, 10, n)
        lens = np.where(kinds == 0, rng.integers(k, k + 6, n),
               np.where(kinds == 1, rng.integers(k + 6, 120, n),
               np.where(kinds == 2, rng.integers(8000, 20000, n), rng.integers(120, 3000, n))))
        base = synth.make_batch(n, model, k, seed=int(rng.integers(1 << 30)), lengths=lens, bad_frac=float(rng.choice([0, 0.1, 0.5])))
        seqs, evs, scs = [], [], []
        for i in range(n):
            s, L = int(base["read_ptr"][i]), int(base["read_len"][i])
            seqs.append(base["reads"][s:s + L].tobytes())
            s, E = int(base["event_ptr"][i]), int(base["n_events"][i])
            ev = base["events"][s:s + E].copy()
            mode = int(rng.integers(0, 12))
            if mode == 0 and E > 4:   ev = ev[::int(rng.integers(2, 12))]                        # few events per base
            elif mode == 1:           ev = np.repeat(ev, int(rng.integers(2, 8)))                # many events per base
            elif mode == 2:           ev = ev[:1]                                                # a single event
            elif mode == 3:           ev = np.repeat(ev, 8)[: max(1, int(L * 15) - int(rng.integers(0, 3)))]   # at the 15.0 guard
            elif mode == 4:           ev = np.repeat(ev, 8)[: int(L * 15) + int(rng.integers(0, 3))]
            elif mode == 5:           ev["mean"] = np.float32(rng.uniform(60, 120))              # constant signal
            elif mode == 6:           ev = ev[: max(1, E // int(rng.integers(2, 6)))]            # events run out early
            elif mode == 7:           ev = np.concatenate([ev, ev])                              # signal goes on after the read ends
            ev = np.ascontiguousarray(ev)
            evs.append(ev)
            sc = (float(base["scalings"]["scale"][i]), float(base["scalings"]["shift"][i]))
            if rng.random() < 0.2: sc = (float(rng.uniform(0.3, 2.5)), float(rng.uniform(-40, 40)))
            scs.append(sc)
        b = synth.batch_from_reads(seqs, evs, scs)
        ora = orc.align_batch(b, model, k, n_threads=16)
        d = ctx.upload(b); ctx.align_db_device(d, scaling=True)           # row N1 rides along
        pairs, n_pairs, diag = ctx.download(d)
        b2e, rsc, epb, flags, nalign = ctx.download_scaling(d)
        plist, n_pairs_h, diag_h = ctx.align_flat_host(b)
        o_pairs, o_n, o_diag = ora
        for i in range(n):
            s = int(b["pair_ptr"][i])
            tag = f"batch {n_batches} read {i} L={len(seqs[i])} E={len(evs[i])}"
            assert n_pairs[i] == o_n[i] == n_pairs_h[i], (tag, n_pairs[i], n_pairs_h[i], o_n[i])
            assert (pairs[s:s + o_n[i]] == o_pairs[s:s + o_n[i]]).all(), tag + " device pairs"
            assert (plist[i] == o_pairs[s:s + o_n[i]]).all(), tag + " host pairs"
            if (diag["flags"][i] & 3) == 0:
                for f in ("n_aligned", "best_event", "max_gap"):
                    assert diag[f][i] == o_diag[f][i] == diag_h[f][i], (tag, f)
                assert abs(diag["sum_emission"][i] - o_diag["sum_emission"][i]) <= 1e-4, tag
            if len(seqs[i]) >= k and len(evs[i]) > 0:                      # scaling_single (f5c.c:736-807) on the device
                r = orc.scaling_single(o_pairs[s:s + o_n[i]], seqs[i], evs[i], model, k, scs[i][0], scs[i][1])
                assert flags[i] == r["flag"] and nalign[i] == r["n_alignment"] and epb[i] == r["events_per_base"], (tag, "N1 scalars")
                if o_n[i] > 0:
                    K, ko = len(seqs[i]) - k + 1, int(d["kmer_ptr"][i])
                    assert (b2e[ko:ko + K, 0] == r["base_to_event_map"]["start"]).all() and \
                           (b2e[ko:ko + K, 1] == r["base_to_event_map"]["stop"]).all(), (tag, "N1 map")
                    if not (r["flag"] & 1) or r["scalings"]["var"] != 0:
                        assert rsc["shift"][i] == r["scalings"]["shift"] and rsc["scale"][i] == r["scalings"]["scale"] and \
                               rsc["var"][i] == r["scalings"]["var"], (tag, "N1 scalings", rsc[i], r["scalings"], r["flag"], scs[i])
        n_batches += 1; n_reads += n; n_pass += int((o_n > 0).sum())
    if own:
        ctx.close()
    return n_batches, n_reads, n_pass, time.time() - t0


if __name__ == "__main__":
    nb, nr, npass, dt = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                            k9=len(sys.argv) > 3 and sys.argv[3] == "k9")
    print(f"fuzz OK: {nb} batches, {nr} reads ({npass} pass QC) bit-exact through both entry points in {dt:.0f} s")
