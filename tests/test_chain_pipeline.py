"""The raw-signal chunk pipeline (f5c_amd/csrc/abea_chain.cpp, round 5) behind abea_events_batch_host (event_db,
src/f5c.c:682-734) and abea_process_batch_host (process_db_rsq: event_db -> align_db -> scaling_db, src/resquiggle.c:283-315),
driven through the C ABI on synthetic raw signals and compared, read by read and bit for bit, with the oracle chain
getevents -> estimate_scalings -> align -> scaling_single:
  * many small chunks rotating through few slots (stage D of chunk c + 1 is issued before stage A of chunk c),
  * tables that overflow their first-guess capacity (ABEA_CHAIN_CAP_DIV) and are redone after the pipeline has drained from the
    int16 staging — with the float signal already rewritten to pA,
  * a read without signal in the middle of the batch, event_db alone, the pA conversion in place.
The real-read version of the same comparison (10 reads from a plain g++ caller) is tests/test_process_chain.py."""
import numpy as np
import pytest


def _oracle_chain(orc, model, k, seq, s16, sc3):
    o_ev, o_pa = orc.getevents(s16, *[float(x) for x in sc3])
    scale, shift = orc.estimate_scalings(seq, model, k, o_ev)
    o_pairs, _ = orc.align(seq, o_ev, model, k, scale, shift)
    rec = orc.scaling_single(o_pairs, seq, o_ev, model, k, scale, shift)
    return o_ev, o_pa, (scale, shift), o_pairs, rec


def _batch_and_signals(r9, n, seed, law):
    from f5c_amd import synth
    k, model = r9
    b = synth.make_batch(n, model, k, seed=seed, law=law, bad_frac=0.05, workers=4)
    sig, sp, ns, sc = synth.make_signals_flat(b, seed=seed + 1, threads=4)
    return b, sig, sp, ns, sc


def _compare(ctx, orc, model, k, b, v, sig0, sp, ns, sc, reads, want_pairs):
    n_aligned = 0
    for j in reads:
        rs, L = int(b["read_ptr"][j]), int(b["read_len"][j])
        seq = b["reads"][rs:rs + L].tobytes()
        if ns[j] == 0:                                                      # f5c.c:727-731, 826-828, 786-794
            assert v["ev_pp"][j] == 0 and v["n_events"][j] == 0 and v["n_pairs"][j] == 0 and v["map_pp"][j] == 0
            assert v["read_stat_flag"][j] & 2 and v["events_per_base"][j] == 0.0
            continue
        s16 = sig0[sp[j]:sp[j] + ns[j]].astype(np.int16)
        o_ev, _, (scale, shift), o_pairs, rec = _oracle_chain(orc, model, k, seq, s16, sc[j])
        g_ev = ctx.view_events(v, j)
        assert len(g_ev) == len(o_ev), j
        for f in ("start", "length", "mean", "stdv"):
            assert (g_ev[f] == o_ev[f]).all(), (j, f)
        est = v["scalings_estimated"][j]
        assert est["scale"] == np.float32(scale) and est["shift"] == np.float32(shift), j
        assert int(v["n_pairs"][j]) == len(o_pairs), j
        assert int(v["read_stat_flag"][j]) == rec["flag"] and int(v["n_event_alignment"][j]) == rec["n_alignment"], j
        assert float(v["events_per_base"][j]) == rec["events_per_base"], j
        if want_pairs:
            assert (ctx.view_pairs(v, j).view(np.int32).reshape(-1, 2) == o_pairs.view(np.int32).reshape(-1, 2)).all(), j
        m = ctx.view_map(v, j, L - k + 1)
        if len(o_pairs) > 0:
            n_aligned += 1
            assert m is not None and (m[:, 0] == rec["base_to_event_map"]["start"]).all() and (m[:, 1] == rec["base_to_event_map"]["stop"]).all(), j
            if not (rec["flag"] & 1):                                        # recalibrated: shift, scale, var, log_var (align.c:755-760)
                for f in ("shift", "scale", "var", "log_var"):
                    assert v["scalings"][f][j] == rec["scalings"][f], (j, f)
        else:
            assert m is None                                                 # f5c.c:787
    return n_aligned


@pytest.mark.gpu
@pytest.mark.parametrize("cap_div", [4, 48])
def test_process_chain_many_chunks_and_overflowed_tables(ctx, orc, r9, monkeypatch, cap_div):
    """120 synthetic reads + one without signal through abea_process_batch_host in ~10 chunks on 3 slots.  cap_div = 48 gives every
    table a first-guess capacity of n/48 + 16 events: most overflow and take the redo path (int16 staging -> exact capacity ->
    host entry) AFTER their float signal has been rewritten to pA; the outputs must not depend on it."""
    k, model = r9
    b, sig, sp, ns, sc = _batch_and_signals(r9, 120, 4242, 2500)
    ns = ns.copy(); ns[57] = 0                                              # a read without signal (nsample == 0)
    sig0 = sig.copy()
    monkeypatch.setenv("ABEA_CHAIN_SLOTS", "3")
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_SAMPLES", str(600_000))
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_READS", "8")
    monkeypatch.setenv("ABEA_CHAIN_CAP_DIV", str(cap_div))
    v = ctx.signal_view(sig, sp, ns, sc, batch=b, want_pairs=True, to_pa=True)
    ctx.process_view(v)
    st = ctx.stats()
    assert st["n_sub_batches"] >= 6 and st["gpu_busy_ms"] > 0
    n_aligned = _compare(ctx, orc, model, k, b, v, sig0, sp, ns, sc, range(120), True)
    assert n_aligned >= 100
    # the signal is left in pA (f5c.c:693-696: the same two float operations), untouched for the read without signal
    for j in (0, 31, 119):
        _, o_pa = orc.getevents(sig0[sp[j]:sp[j] + ns[j]].astype(np.int16), *[float(x) for x in sc[j]])
        assert (sig[sp[j]:sp[j] + ns[j]] == o_pa).all(), j
    if cap_div == 48:                                                        # the synthetic signals give one event per ~5 samples: all overflow n/48
        assert int((v["n_events"] > ns // 48 + 16).sum()) >= 100
    ctx.free_view(v)


@pytest.mark.gpu
def test_event_db_alone_and_repeatability(ctx, orc, r9, monkeypatch):
    """abea_events_batch_host through the same pipeline (no alignment stage): tables + method-of-moments scalings equal the
    oracle's, with and without the sequences; a second call gives the same tables; a fractional sample is refused."""
    from f5c_amd import abea
    k, model = r9
    b, sig, sp, ns, sc = _batch_and_signals(r9, 60, 99, 1800)
    monkeypatch.setenv("ABEA_CHAIN_SLOTS", "2")
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_SAMPLES", str(400_000))
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_READS", "4")
    v = ctx.signal_view(sig, sp, ns, sc, batch=b)
    ctx.events_view(v)
    first = [ctx.view_events(v, j) for j in range(60)]

    def same(a, b2):                                                         # field by field: event_t has 4 bytes of tail padding
        return len(a) == len(b2) and all((a[f] == b2[f]).all() for f in ("start", "length", "mean", "stdv"))
    for j in range(0, 60, 7):
        rs, L = int(b["read_ptr"][j]), int(b["read_len"][j])
        o_ev, _ = orc.getevents(sig[sp[j]:sp[j] + ns[j]].astype(np.int16), *[float(x) for x in sc[j]])
        assert len(first[j]) == len(o_ev) and all((first[j][f] == o_ev[f]).all() for f in ("start", "length", "mean", "stdv")), j
        scale, shift = orc.estimate_scalings(b["reads"][rs:rs + L].tobytes(), model, k, o_ev)
        assert v["scalings"]["scale"][j] == np.float32(scale) and v["scalings"]["shift"][j] == np.float32(shift), j
    ctx.free_view(v)
    ctx.events_view(v)
    assert all(same(ctx.view_events(v, j), first[j]) for j in range(60))
    ctx.free_view(v)
    v2 = ctx.signal_view(sig, sp, ns, sc)                                    # no sequences: tables only
    ctx.events_view(v2)
    assert all(same(ctx.view_events(v2, j), first[j]) for j in range(0, 60, 5))
    ctx.free_view(v2)
    bad = sig.copy(); bad[sp[3] + 10] += 0.5                                 # not an ADC count
    v3 = ctx.signal_view(bad, sp, ns, sc)
    with pytest.raises(abea.AbeaError, match="int16 ADC count"):
        ctx.events_view(v3)
    assert (v3["ev_pp"] == 0).all()                                          # nothing half-built is handed back


@pytest.mark.gpu
def test_process_chain_on_a_two_device_context(orc, r9, monkeypatch):
    """abea_process_batch_host on a multi-device context (one GPU listed twice, as the alignment's multi-device test does): the reads
    are LPT-split on the sample count, each device context runs the pipeline on its share from its own host thread; the outputs are
    those of the single-device call."""
    from f5c_amd import abea
    k, model = r9
    b, sig, sp, ns, sc = _batch_and_signals(r9, 50, 777, 2000)
    monkeypatch.setenv("ABEA_CHAIN_SLOTS", "2")
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_SAMPLES", str(500_000))
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_READS", "4")
    with abea.AbeaContext(model, k, device_ids=[0, 0], max_arena_bytes=3 << 30) as c2:
        assert c2.device_count() == 2
        v = c2.signal_view(sig, sp, ns, sc, batch=b, want_pairs=True)
        c2.process_view(v)
        st = c2.stats()
        assert st["n_devices"] == 2 and st["n_sub_batches"] >= 4
        assert _compare(c2, orc, model, k, b, v, sig, sp, ns, sc, range(0, 50, 3), True) >= 12
        c2.free_view(v)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,mover", [("full", "kernel"), ("packed", "kernel"), ("full", "engine"), ("packed", "engine")])
def test_tables_cross_pcie_in_either_form_by_either_mover(ctx, orc, r9, monkeypatch, fmt, mover):
    """Round 6: the event tables come down as 24-byte event_t or as 12-byte {start, mean, stdv} records from which the retire loop
    rebuilds (start, length) — the events of a read tile its samples, events.c:466-513 — and by abea_copy_out_kernel or by the copy
    engine on the slot's own stream.  Every combination gives the oracle's tables and the oracle's chain, through both entries."""
    k, model = r9
    b, sig, sp, ns, sc = _batch_and_signals(r9, 48, 31337, 1700)
    monkeypatch.setenv("ABEA_CHAIN_SLOTS", "3")
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_SAMPLES", str(300_000))
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_READS", "5")
    monkeypatch.setenv("ABEA_CHAIN_TABLE_FORMAT", fmt)
    monkeypatch.setenv("ABEA_CHAIN_TABLE_COPY", mover)
    v = ctx.signal_view(sig, sp, ns, sc, batch=b, want_pairs=True)
    ctx.process_view(v)
    assert ctx.stats()["n_sub_batches"] >= 5
    assert _compare(ctx, orc, model, k, b, v, sig, sp, ns, sc, range(48), True) >= 40
    d2h_process = ctx.stats()["d2h_bytes"]
    ctx.free_view(v)
    ve = ctx.signal_view(sig, sp, ns, sc)                                    # event_db alone, no sequences
    ctx.events_view(ve)
    n_ev = int(ve["n_events"].sum())
    for j in range(0, 48, 5):
        o_ev, _ = orc.getevents(sig[sp[j]:sp[j] + ns[j]].astype(np.int16), *[float(x) for x in sc[j]])
        g = ctx.view_events(ve, j)
        assert len(g) == len(o_ev) and all((g[f] == o_ev[f]).all() for f in ("start", "length", "mean", "stdv")), j
    d2h = ctx.stats()["d2h_bytes"]
    ctx.free_view(ve)
    per_event = 12 if fmt == "packed" else 24
    assert per_event * n_ev <= d2h <= per_event * n_ev + 48 * 64 * 48 and d2h_process > d2h     # the tables dominate what comes down


@pytest.mark.gpu
def test_arena_limit_closes_chunks_and_whole_waves_are_charged(orc, r9, monkeypatch):
    """Round-5 advisor finding: the detector lays its scratch out in 64-lane waves as long as their longest read, the chunk carving
    charged one lane per read; a chunk the ARENA LIMIT closed at a read count that is not a multiple of 64 then failed the whole batch
    with "internal: the detector needs ...".  A small arena, equal reads, chunk limits far away: the arena closes every chunk, the
    carving charges whole waves, and the call succeeds with the oracle's tables; a read whose single wave cannot fit is refused
    up front with the reason, not half-way with an internal error."""
    from f5c_amd import abea
    k, model = r9
    b, sig, sp, ns, sc = _batch_and_signals(r9, 150, 5150, 3000)
    monkeypatch.setenv("ABEA_CHAIN_SLOTS", "2")
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_SAMPLES", str(1 << 40))
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_READS", "100000")
    monkeypatch.setenv("ABEA_CHAIN_CHUNK_READS_MAX", "100000")
    n_max = int(ns.max())
    wave = (n_max + 1) * 64 * 24                                             # S, Q, two t-statistics of one wave
    # a slot's share holds two waves and a bit; the per-lane charge of round 5 would have packed ~2.6 x 64 reads into a chunk (three
    # waves of scratch) and failed in stage D
    with abea.AbeaContext(model, k, max_arena_bytes=int(2 * 2.6 * wave)) as c1:
        for mode in ("process", "events"):
            v = c1.signal_view(sig, sp, ns, sc, batch=b, want_pairs=(mode == "process"))
            (c1.process_view if mode == "process" else c1.events_view)(v)
            st = c1.stats()
            assert 2 <= st["n_sub_batches"] <= 4, st["n_sub_batches"]         # closed by the arena: the other limits are out of reach
            if mode == "process":
                assert _compare(c1, orc, model, k, b, v, sig, sp, ns, sc, range(0, 150, 7), True) >= 18
            else:
                for j in range(0, 150, 11):
                    o_ev, _ = orc.getevents(sig[sp[j]:sp[j] + ns[j]].astype(np.int16), *[float(x) for x in sc[j]])
                    g = c1.view_events(v, j)
                    assert len(g) == len(o_ev) and all((g[f] == o_ev[f]).all() for f in ("start", "length", "mean", "stdv")), j
            c1.free_view(v)
    with abea.AbeaContext(model, k, max_arena_bytes=int(2 * 0.8 * wave)) as c2:     # not even one wave of the longest read
        v = c2.signal_view(sig, sp, ns, sc, batch=b)
        with pytest.raises(abea.AbeaError, match="does not fit a .* share of the arena"):
            c2.events_view(v)
        assert (v["ev_pp"] == 0).all()


@pytest.mark.gpu
def test_link_probe_reports_both_directions(ctx):
    """abea_link_probe: the ceilings bench.py prices the raw-signal entries against; sane on any PCIe generation."""
    r = ctx.link_probe(64 << 20, 2)
    assert set(r) == {"h2d_copy", "d2h_copy", "d2h_kernel", "both_h2d_copy", "both_d2h_kernel", "both_copy_h2d", "both_copy_d2h",
                      "h2d_kernel", "both_h2d_kernel", "both_d2h_copy"}
    assert all(1.0 < v < 200.0 for v in r.values()), r
