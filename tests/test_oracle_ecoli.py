"""CPU: the oracle's per-read chain (event detection -> scalings -> ABEA -> recalibration) against the
reference's goldens for test/ecoli_2kb_region (adaptive.exp, est_scalings.exp, recalib_scalings.exp)."""
import os
import sys
import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _reads():
    z = np.load(os.path.join(GOLD, "ecoli_reads.npz"))
    for i in range(int(z["n"])):
        off, rng, dig = z["scaling"][i]
        yield dict(sig=z[f"sig{i}"], seq=z[f"seq{i}"].tobytes(), offset=off, range=rng, digitisation=dig,
                   ada=str(z["ada"][i]), est=str(z["est"][i]), rec=str(z["rec"][i]), n_events=int(z["n_events"][i]),
                   read_id=str(z["read_id"][i]))


def test_real_signal_chain_matches_reference_goldens(orc, r9):
    """10 real reads (raw FAST5 signal committed as data): printed est_scalings / recalib_scalings lines equal
    the reference's, n_aligned_events exact, sum_emission to 1e-6 relative."""
    k, model = r9
    n = 0
    for r in _reads():
        ev, _ = orc.getevents(r["sig"], r["offset"], r["range"], r["digitisation"])
        assert len(ev) == r["n_events"]
        scale, shift = orc.estimate_scalings(r["seq"], model, k, ev)
        assert "%.2f %.2f" % (shift, scale) == r["est"]                 # est_scalings.exp
        pairs, d = orc.align(r["seq"], ev, model, k, scale, shift)
        gsum, gn = r["ada"].split()
        assert d["n_aligned"] == int(gn) == len(pairs)                   # adaptive.exp
        assert abs(d["sum_emission"] - float(gsum)) <= 1e-6 * abs(float(gsum)) + 1e-6
        rec = orc.scaling_single(pairs, r["seq"], ev, model, k, scale, shift)
        assert "%.2f %.2f %.2f" % (rec["scalings"]["shift"], rec["scalings"]["scale"], rec["scalings"]["var"]) == r["rec"]
        n += 1
    assert n == 10


@pytest.mark.skipif(not (os.path.exists("/root/reference/test/ecoli_2kb_region/fast5_files")
                         and os.path.exists("/opt/conda/bin/h5dump")), reason="needs the reference data + h5dump")
def test_all_112_ecoli_reads_against_goldens():
    """Build container only: every FAST5 of the reference's test set through the oracle chain; all 111 unique
    adaptive.exp records are reproduced (the 112th read is not in the BAM)."""
    sys.path.insert(0, GOLD)
    import make_ecoli_golden as M
    r = M.main(write=False)
    assert r["n"] == 112 and r["gold_unique"] == 111
    assert r["covered"] == 111 and r["adaptive"] == 111
    assert r["est"] >= 111 and r["recalib"] >= 110


@pytest.mark.gpu
def test_gpu_bit_exact_on_real_signal_reads(ctx, orc, r9):
    """HIP path on real nanopore reads (events from the oracle's event detection) vs the oracle."""
    from f5c_amd import synth
    k, model = r9
    seqs, evs, scs = [], [], []
    for r in _reads():
        ev, _ = orc.getevents(r["sig"], r["offset"], r["range"], r["digitisation"])
        scale, shift = orc.estimate_scalings(r["seq"], model, k, ev)
        seqs.append(r["seq"]); evs.append(ev); scs.append((scale, shift))
    batch = synth.batch_from_reads(seqs, evs, scs)
    d = ctx.upload(batch)
    ctx.align_db_device(d)
    pairs, n_pairs, diag = ctx.download(d)
    o_pairs, o_n, o_diag = orc.align_batch(batch, model, k, n_threads=8)
    assert (n_pairs == o_n).all() and (o_n > 0).all()
    for i in range(len(o_n)):
        s = int(batch["pair_ptr"][i])
        assert (pairs[s:s + o_n[i]] == o_pairs[s:s + o_n[i]]).all()
    assert np.allclose(diag["sum_emission"], o_diag["sum_emission"], rtol=0, atol=1e-4)


@pytest.mark.gpu
def test_gpu_event_detection_bit_exact(ctx, orc, r9):
    """Row N2: raw ADC signal -> events + method-of-moments scalings on the device, bit-exact against the oracle's
    event detection (itself pinned to est_scalings.exp / adaptive.exp), on real reads and synthetic signals."""
    from f5c_amd import synth
    k, model = r9
    reads = list(_reads())
    sigs = [r["sig"] for r in reads]
    scal = np.array([[r["offset"], r["range"], r["digitisation"]] for r in reads], dtype=np.float32)
    seqs = [r["seq"] for r in reads]
    b = synth.make_batch(40, model, k, seed=91, law=1500, bad_frac=0.0)
    s_sigs, s_scal = synth.make_signals(b, seed=3)
    for i in range(40):
        s, L = int(b["read_ptr"][i]), int(b["read_len"][i])
        seqs.append(b["reads"][s:s + L].tobytes())
    sigs += s_sigs
    scal = np.concatenate([scal, s_scal])
    evs, ne, sc = ctx.detect_events_device(sigs, scal, seqs=seqs)
    for i, sig in enumerate(sigs):
        o_ev, _ = orc.getevents(sig, scal[i, 0], scal[i, 1], scal[i, 2])
        assert ne[i] == len(o_ev), (i, ne[i], len(o_ev))
        for f in ("start", "length", "mean", "stdv"):
            assert (evs[i][f] == o_ev[f]).all(), (i, f)
        scale, shift = orc.estimate_scalings(seqs[i], model, k, o_ev)
        assert sc["scale"][i] == np.float32(scale) and sc["shift"][i] == np.float32(shift)
    # the printed est_scalings.exp lines of the real reads come out of the device path too
    for i, r in enumerate(reads):
        assert "%.2f %.2f" % (sc["shift"][i], sc["scale"][i]) == r["est"]
    assert 5 < np.mean([len(s) / max(1, n) for s, n in zip(sigs, ne)]) < 15     # ~10 samples per event


@pytest.mark.gpu
def test_gpu_event_detection_adversarial_signals(ctx, orc):
    """The segment-parallel detector must equal the sequential one on signals where speculative segments do not
    re-synchronise quickly (plateaus, pure noise, slow waves: these take the per-read sequential fallback) and on
    degenerate lengths."""
    r = np.random.default_rng(5)
    sigs = [np.full(5000, 500, np.int16), np.full(7, 500, np.int16), np.full(1, 500, np.int16),
            np.full(513, 480, np.int16), np.arange(400, 1424, dtype=np.int16),
            r.integers(300, 700, 100000).astype(np.int16),
            np.repeat(r.integers(300, 700, 400), 250).astype(np.int16),
            (np.repeat(r.integers(300, 700, 30000), 3) + r.integers(-2, 3, 90000)).astype(np.int16),
            (500 + 100 * np.sin(np.arange(200000) / 50.0)).astype(np.int16),
            (500 + 60 * np.sin(np.arange(70000) / 700.0) + r.normal(0, 0.6, 70000)).astype(np.int16)]
    scal = np.tile(np.array([10.0, 1467.61, 8192.0], dtype=np.float32), (len(sigs), 1))
    # a channel offset that puts some samples within 1e-3 pA of zero: the prefix sums are then not provably exact in
    # every order and that read must take the sequential sums kernel
    sigs.append(r.integers(495, 700, 60000).astype(np.int16))
    scal = np.concatenate([scal, np.array([[-499.999, 1467.61, 8192.0]], dtype=np.float32)])
    evs, ne, _ = ctx.detect_events_device(sigs, scal, cap_div=1)
    for i, sig in enumerate(sigs):
        o_ev, _ = orc.getevents(sig, scal[i, 0], scal[i, 1], scal[i, 2])
        assert ne[i] == len(o_ev), (i, ne[i], len(o_ev))
        for f in ("start", "length", "mean", "stdv"):
            assert (evs[i][f] == o_ev[f]).all(), (i, f)


@pytest.mark.gpu
def test_gpu_raw_signal_to_recalibrated_scalings(ctx, orc, r9):
    """Whole device chain on real reads: raw signal -> events -> scalings -> ABEA -> scaling_single; the printed
    recalib_scalings.exp / adaptive.exp values of the reference come out of the GPU."""
    from f5c_amd import synth
    k, model = r9
    reads = list(_reads())
    # the event tables stay in HBM between the two calls (only n_events and the estimated scalings come to the host)
    d = ctx.signals_to_device_batch([r["sig"] for r in reads],
                                    np.array([[r["offset"], r["range"], r["digitisation"]] for r in reads]),
                                    [r["seq"] for r in reads])
    ctx.align_db_device(d, scaling=True)
    _, n_pairs, diag = ctx.download(d)
    _, rsc, _, flags, _ = ctx.download_scaling(d)
    for i, r in enumerate(reads):
        gsum, gn = r["ada"].split()
        assert n_pairs[i] == int(gn)
        assert abs(diag["sum_emission"][i] - float(gsum)) <= 1e-6 * abs(float(gsum)) + 1e-6
        assert "%.2f %.2f %.2f" % (rsc["shift"][i], rsc["scale"][i], rsc["var"][i]) == r["rec"]
        assert flags[i] == 0
