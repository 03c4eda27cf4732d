"""Row N2, RNA branch: getevents(nsample, rawptr, rna=1) runs the detector with event_detection_rna (src/events.c:59-65,
575-577: windows 7 / 14, thresholds 2.5 / 9.0, peak height 1.0) and event_single() reverses the table to 3'->5' after the
scalings are estimated (src/f5c.c:698-719).  UNPINNED against reference output: the reference's RNA test sets are
downloaded by test/test_eventalign.sh -e and are not in the mount; what is checked is (CPU) the oracle's RNA row against
an independent numpy restatement of the t-statistic and its behaviour on RNA-like signals, and (GPU) the device path
against the oracle bit for bit, through to the alignment with the reference's R9.4 RNA 5-mer model
(test/r9-models/r9.4_70bps.u_to_t_rna.5mer.template.model, committed as data)."""
import os
import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rna_model():
    from f5c_amd.model import _finish
    tab = np.fromfile(os.path.join(GOLD, "r9.4_70bps.rna.5mer.f32"), dtype=np.float32).reshape(-1, 2)
    assert len(tab) == 4 ** 5
    return 5, _finish(tab[:, 0], tab[:, 1])


def rna_like_reads(n, seed, lo=300, hi=1500):
    """Synthetic direct-RNA reads: the strand goes through the pore 3'->5' at ~70 bases/s sampled at 3 kHz (~40 samples per
    base, exponential dwells), so the SIGNAL is in reverse read order.  Returns (seqs, int16 signals, scaling[n,3])."""
    k, model = rna_model()
    r = np.random.default_rng(seed)
    seqs, sigs = [], []
    raw_unit = np.float32(1467.61) / np.float32(8192.0)
    for _ in range(n):
        L = int(r.integers(lo, hi))
        codes = r.integers(0, 4, L)
        seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()
        K = L - k + 1
        rank = np.zeros(K, dtype=np.int64)
        for j in range(k):
            rank = rank * 4 + codes[j:j + K]
        dwell = np.maximum(6, r.exponential(40.0, K)).astype(np.int64)
        lv = model["level_mean"][rank][::-1].astype(np.float64)                 # 3' end first
        sd = model["level_stdv"][rank][::-1].astype(np.float64)
        d = dwell[::-1]
        pa = np.repeat(lv * r.normal(1.0, 0.03) + r.normal(0.0, 4.0), d) + r.normal(0.0, 1.0, int(d.sum())) * np.repeat(sd, d) * 0.6
        sigs.append(np.clip(np.rint(pa / raw_unit - 10.0), -32768, 32767).astype(np.int16))
        seqs.append(seq)
    return seqs, sigs, np.tile(np.array([10.0, 1467.61, 8192.0], dtype=np.float32), (n, 1))


def _tstat_numpy(pa, w):
    """events.c:324-369 written with numpy prefix sums (independent of the oracle's loop)."""
    n = len(pa)
    t = np.zeros(n, dtype=np.float32)
    if n < 2 * w or w < 2:
        return t
    x = pa.astype(np.float32)
    S = np.concatenate([[0.0], np.cumsum(x.astype(np.float64))])
    Q = np.concatenate([[0.0], np.cumsum((x * x).astype(np.float64))])      # float product, double sum (events.c:311)
    i = np.arange(w, n - w + 1)
    wf = np.float32(w)
    sum1 = S[i] - np.where(i > w, S[i - w], 0.0); sq1 = Q[i] - np.where(i > w, Q[i - w], 0.0)
    sum2 = (S[i + w] - S[i]).astype(np.float32); sq2 = (Q[i + w] - Q[i]).astype(np.float32)
    mean1 = (sum1 / np.float64(wf)).astype(np.float32); mean2 = sum2 / wf
    # sumsq1 / w_lengthf is a double / float -> double; the whole expression is evaluated in double, then stored as float
    cv = (sq1 / np.float64(wf) - (mean1 * mean1).astype(np.float64) + (sq2 / wf).astype(np.float64)
          - (mean2 * mean2).astype(np.float64)).astype(np.float32)
    cv = np.maximum(cv, np.float32(np.finfo(np.float32).tiny))
    t[i] = np.abs(mean2 - mean1) / np.sqrt(cv / wf)
    return t


def test_rna_detector_row_of_the_oracle(orc):
    """The RNA parameters change what is detected the way their meaning says (wider windows, higher peak threshold:
    fewer, longer events), leave the DNA row untouched, and the reversal is the reference's swap loop."""
    seqs, sigs, sc = rna_like_reads(6, 11)
    for seq, sg in zip(seqs, sigs):
        dna, pa = orc.getevents(sg, *sc[0])
        rna, _ = orc.getevents(sg, *sc[0], rna=True)
        assert len(rna) < len(dna)
        K = len(seq) - 5 + 1
        assert 0.6 * K < len(rna) < 1.6 * K                       # about one event per k-mer at these dwells
        assert rna["start"][0] == 0 and (np.diff(rna["start"].astype(np.int64)) > 0).all()
        assert int(rna["start"][-1]) + int(rna["length"][-1]) == len(sg)
        assert (rna["length"][1:-1] >= 4).all()                   # a peak needs (i - peak_pos) > window/2 = 3
        rev = orc.reverse_events(rna)
        assert (rev == rna[::-1]).all() and (orc.reverse_events(rev) == rna).all()
        # every event boundary is a local maximum of one of the two RNA t-statistics (windows 7 and 14)
        t7, t14 = _tstat_numpy(pa, 7), _tstat_numpy(pa, 14)
        b = rna["start"][1:].astype(np.int64)
        is_max7 = (t7[b] >= t7[b - 1]) & (t7[b] >= t7[np.minimum(b + 1, len(pa) - 1)]) & (t7[b] > 2.5)
        is_max14 = (t14[b] >= t14[b - 1]) & (t14[b] >= t14[np.minimum(b + 1, len(pa) - 1)]) & (t14[b] > 9.0)
        assert (is_max7 | is_max14).all()


def test_rna_chain_on_the_oracle_aligns_reversed_tables(orc):
    """event_single's order of operations for RNA (f5c.c:702-719): scalings from the table in detection order, then the
    reversal; the reversed table aligns to the read with the RNA 5-mer model and passes QC."""
    k, model = rna_model()
    seqs, sigs, sc = rna_like_reads(8, 13)
    ok = 0
    for seq, sg in zip(seqs, sigs):
        ev, _ = orc.getevents(sg, *sc[0], rna=True)
        scale, shift = orc.estimate_scalings(seq, model, k, ev)
        pairs, d = orc.align(seq, orc.reverse_events(ev), model, k, scale, shift)
        fwd, _ = orc.align(seq, ev, model, k, scale, shift)          # without the reversal the signal runs the wrong way
        ok += len(pairs) > 0
        assert len(fwd) == 0
    assert ok >= 6


@pytest.mark.gpu
def test_gpu_rna_event_detection_and_alignment_bit_exact(orc):
    """Device path with rna=1: event tables (reversed), n_events and method-of-moments scalings equal the oracle's bit for
    bit; the device batch then aligns with the RNA 5-mer model exactly as the oracle does."""
    from f5c_amd import abea, synth
    k, model = rna_model()
    seqs, sigs, sc = rna_like_reads(70, 17)
    # degenerate lengths next to the RNA windows (2*7, 2*14 samples) and a constant signal
    extra = [np.full(n, 500, np.int16) for n in (1, 13, 14, 15, 27, 28, 29, 600)]
    r = np.random.default_rng(3)
    extra += [r.integers(300, 700, n).astype(np.int16) for n in (14, 28, 29, 513, 5000)]
    with abea.AbeaContext(model, k, max_arena_bytes=3 << 30) as ctx:
        evs, ne, dsc = ctx.detect_events_device(sigs, sc, seqs=seqs, cap_div=2, rna=True)
        o_evs, o_sc = [], []
        for i, sg in enumerate(sigs):
            o_ev, _ = orc.getevents(sg, *sc[i], rna=True)
            assert ne[i] == len(o_ev), (i, ne[i], len(o_ev))
            rev = orc.reverse_events(o_ev)
            for f in ("start", "length", "mean", "stdv"):
                assert (evs[i][f] == rev[f]).all(), (i, f)
            scale, shift = orc.estimate_scalings(seqs[i], model, k, o_ev)     # detection order (f5c.c:707-709)
            assert dsc["scale"][i] == np.float32(scale) and dsc["shift"][i] == np.float32(shift)
            o_evs.append(rev); o_sc.append((scale, shift))
        sc2 = np.tile(sc[0], (len(extra), 1))
        evs2, ne2, _ = ctx.detect_events_device(extra, sc2, cap_div=1, rna=True)
        for i, sg in enumerate(extra):
            o_ev, _ = orc.getevents(sg, *sc2[i], rna=True)
            assert ne2[i] == len(o_ev)
            rev = orc.reverse_events(o_ev)
            for f in ("start", "length", "mean", "stdv"):
                assert (evs2[i][f] == rev[f]).all(), (i, f)
        # raw signal -> events stay in HBM -> alignment + scaling_single with the RNA model
        d = ctx.signals_to_device_batch(sigs, sc, seqs, cap_div=2, rna=True)
        ctx.align_db_device(d, scaling=True)
        pairs, n_pairs, diag = ctx.download(d)
        batch = synth.batch_from_reads(seqs, o_evs, o_sc)
        o_pairs, o_n, o_diag = orc.align_batch(batch, model, k, n_threads=8)
        assert (n_pairs == o_n).all() and (o_n > 0).mean() > 0.7
        for i in range(len(o_n)):
            a, b = int(d["pair_ptr"][i]), int(batch["pair_ptr"][i])
            assert (pairs[a:a + o_n[i]] == o_pairs[b:b + o_n[i]]).all()
        assert (diag["sum_emission"][o_n > 0] == o_diag["sum_emission"][o_n > 0]).all()
