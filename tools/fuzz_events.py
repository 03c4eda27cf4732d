#!/usr/bin/env python3
"""Randomised parity sweep (GPU) of device event detection (row N2) against the oracle's getevents: signals of every
length from 1 sample up, event-like steps with assorted dwell times and noise, plateaus, ramps, pure noise, full-range
ADC values, channel offsets that put samples next to 0 pA (sequential-sums fallback) and odd range/digitisation.
tests/test_fuzz_gpu.py runs a seeded, time-boxed slice under -m gpu.  Long sweep:  python tools/fuzz_events.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5c_amd import abea, load_model_f32
from oracle import orc



def run(budget=60.0, seed=1, ctx=None, max_batches=None, host_entry=False):
    """Fuzz for `budget` seconds (or max_batches); returns (batches, signals, events, seconds).  Raises on any mismatch.
    host_entry: the same signals as FLOAT ADC counts through abea_events_batch_host — the chunk pipeline of abea_chain.cpp — with
    random chunk sizes, slot counts, first-guess table capacities (n/1 .. n/64 + 16 events: overflowing tables take the redo path) and both
    forms / movers of the tables' trip down (ABEA_CHAIN_TABLE_FORMAT, ABEA_CHAIN_TABLE_COPY)."""
    k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        ctx = abea.AbeaContext(model, k, max_arena_bytes=4 << 30)

    def signal():
        kind = int(rng.integers(0, 9))
        n = int(rng.choice([rng.integers(1, 30), rng.integers(30, 1100), rng.integers(1100, 40000), rng.integers(40000, 300000)],
                           p=[0.2, 0.3, 0.4, 0.1]))
        if kind == 0:      # event-like steps
            dwell = int(rng.choice([2, 3, 5, 9, 20, 80, 600]))
            lv = rng.integers(350, 750, n // dwell + 2)
            s = np.resize(np.repeat(lv, rng.integers(1, 2 * dwell + 1, len(lv))), n) + rng.normal(0, rng.choice([0, 0.5, 3, 15]), n)
        elif kind == 1:    s = np.full(n, rng.integers(-100, 900))
        elif kind == 2:    s = rng.integers(300, 700, n)
        elif kind == 3:    s = np.linspace(rng.integers(0, 500), rng.integers(500, 1500), n)
        elif kind == 4:    s = 500 + rng.uniform(5, 300) * np.sin(np.arange(n) / rng.uniform(2, 900))
        elif kind == 5:    s = rng.integers(-32768, 32767, n)
        elif kind == 6:    s = np.where(rng.random(n) < 0.01, rng.integers(-2000, 5000, n), 520) + rng.normal(0, 1, n)
        elif kind == 7:    s = np.resize(np.repeat(rng.integers(400, 700, n // 7 + 2), 7), n) + rng.integers(-1, 2, n)
        else:              s = np.cumsum(rng.normal(0, 2, n)) + 500
        s = np.clip(np.rint(np.resize(s, n)), -32768, 32767).astype(np.int16)
        sc = [float(rng.choice([10.0, 3.0, -12.0, 0.0, -499.999, 21.5])), float(rng.choice([1467.61, 748.58, 2903.1])),
              float(rng.choice([8192.0, 2048.0]))]
        return s, sc


    t0 = time.time(); nb = ns = ne_tot = 0
    while time.time() - t0 < budget and (max_batches is None or nb < max_batches):
        m = int(rng.integers(1, 90))
        sigs, scal = zip(*[signal() for _ in range(m)])
        scal = np.array(scal, dtype=np.float32)
        seqs = None
        if nb % 2 == 1:                                                  # every other batch also asks for the scalings
            seqs = [bytes(rng.choice(list(b"ACGT"), int(rng.integers(k, 4000))).astype(np.uint8)) for _ in range(m)]
        rna = bool(nb % 3 == 2)                                          # every third batch with the RNA detector (events.c:59-65) + reversal
        if host_entry:
            os.environ["ABEA_CHAIN_CAP_DIV"] = str(int(rng.choice([1, 4, 16, 64])))
            os.environ["ABEA_CHAIN_SLOTS"] = str(int(rng.integers(1, 5)))
            os.environ["ABEA_CHAIN_CHUNK_SAMPLES"] = str(int(rng.choice([2000, 60000, 1 << 20])))
            os.environ["ABEA_CHAIN_CHUNK_READS"] = str(int(rng.integers(1, 9)))
            os.environ["ABEA_CHAIN_TABLE_FORMAT"] = str(rng.choice(["full", "packed"]))      # 24-byte event_t or 12-byte records over PCIe
            os.environ["ABEA_CHAIN_TABLE_COPY"] = str(rng.choice(["kernel", "engine"]))      # copy-out kernel or copy engine
            nsamp = np.array([len(x) for x in sigs], dtype=np.int64)
            pad = (nsamp + 7) // 8 * 8
            sp = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
            flat = np.zeros(int(pad.sum()) + 8, dtype=np.float32)
            for i, sg in enumerate(sigs):
                flat[sp[i]:sp[i] + nsamp[i]] = sg
            bt = None
            if seqs is not None:
                rl = np.array([len(x) for x in seqs], dtype=np.int32)
                rp = np.concatenate([[0], np.cumsum(rl.astype(np.int64) + 1)[:-1]]).astype(np.int64)
                rd = np.zeros(int((rl.astype(np.int64) + 1).sum()), dtype=np.uint8)
                for i, x in enumerate(seqs):
                    rd[rp[i]:rp[i] + rl[i]] = np.frombuffer(x, dtype=np.uint8)
                bt = dict(reads=rd, read_len=rl, read_ptr=rp)
            v = ctx.signal_view(flat, sp, nsamp, scal, batch=bt, rna=rna)
            ctx.events_view(v)
            evs = [ctx.view_events(v, i) for i in range(m)]
            ne = v["n_events"].astype(np.int64)
            dsc = v["scalings"]
            ctx.free_view(v)
        else:
            evs, ne, dsc = ctx.detect_events_device(list(sigs), scal, seqs=seqs, cap_div=1, rna=rna)
        for i, sg in enumerate(sigs):
            o_ev, _ = orc.getevents(sg, scal[i, 0], scal[i, 1], scal[i, 2], rna=rna)
            tag = f"batch {nb} signal {i} n={len(sg)} scaling={scal[i]}"
            assert ne[i] == len(o_ev), (tag, int(ne[i]), len(o_ev))
            o_out = orc.reverse_events(o_ev) if rna else o_ev            # f5c.c:711-719, after the scalings
            for f in ("start", "length", "mean", "stdv"):
                a, b = evs[i][f], o_out[f]
                assert ((a == b) | ((a != a) & (b != b))).all(), (tag, f, rna)
            if seqs is not None:                                         # estimate_scalings_using_mom (align.c:58-106)
                scale, shift = orc.estimate_scalings(seqs[i], model, k, o_ev)
                got = (np.float32(dsc["scale"][i]), np.float32(dsc["shift"][i])); want = (np.float32(scale), np.float32(shift))
                assert all((g == w) or (g != g and w != w) for g, w in zip(got, want)), (tag, "scalings", got, want)
        nb += 1; ns += m; ne_tot += int(ne.sum())
    if own:
        ctx.close()
    return nb, ns, ne_tot, time.time() - t0


if __name__ == "__main__":
    nb, ns, ne_tot, dt = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                             host_entry=len(sys.argv) > 3 and sys.argv[3] == "host")
    print(f"fuzz OK ({'host entry' if len(sys.argv) > 3 and sys.argv[3] == 'host' else 'device entry'}): {nb} batches, {ns} signals, "
          f"{ne_tot} events bit-exact in {dt:.0f} s")
