"""The host-buffer entry (abea_align_batch_host = what align_db costs its caller, src/f5c.cu:647-1061): chunk pipeline,
both pair-return modes, fused scaling_single, the nsample guard, error exits, and the in-library multi-device dispatch.
GPU results are compared with the CPU oracle bit for bit."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_host(batch, plist, n_pairs, ora):
    o_pairs, o_n, _ = ora
    assert (n_pairs == o_n).all(), np.nonzero(n_pairs != o_n)[0][:10]
    for i in range(len(o_n)):
        s = int(batch["pair_ptr"][i])
        assert (plist[i] == o_pairs[s:s + o_n[i]]).all(), f"read {i}"


# ---------------------------------------------------------------- CPU: the splitter is host-only code
def test_lpt_split_is_the_sharding_rule(r9):
    """abea_lpt_split (the in-library multi-GPU split) = f5c_amd.synth.shard_batch (the bench's per-rank split):
    same bins, every read exactly once, bins balanced on the band count."""
    from f5c_amd import abea, synth
    k, model = r9
    lib = abea.load_library()
    rng = np.random.default_rng(3)
    L = np.exp(rng.uniform(np.log(1000.0), np.log(50000.0), 5000)).astype(np.int64)
    E = (2 * L + rng.integers(-50, 50, len(L))).astype(np.int64)
    w = np.ascontiguousarray(E + (L - k + 1) + 2)
    w[::97] = 0                                              # guard failures cost nothing
    for nb in (1, 2, 3, 8):
        bins = np.full(len(w), -1, dtype=np.int32)
        assert lib.abea_lpt_split(w.ctypes.data, len(w), nb, bins.ctypes.data) == 0
        assert bins.min() == 0 and bins.max() == nb - 1
        loads = np.bincount(bins, weights=w, minlength=nb)
        assert loads.max() - loads.min() <= w.max()          # LPT bound
        if nb > 1:
            assert loads.max() / loads.mean() < 1.01
    # same rule as the Python splitter the bench uses per rank
    fake = dict(read_len=L.astype(np.int32), n_events=(w - L).astype(np.int32))
    import heapq
    order = np.argsort(-w, kind="stable")
    heap = [(0, r) for r in range(4)]
    ref = np.zeros(len(w), dtype=np.int32)
    for i in order:
        load, r = heapq.heappop(heap)
        ref[i] = r
        heapq.heappush(heap, (load + int(w[i]), r))
    bins = np.zeros(len(w), dtype=np.int32)
    lib.abea_lpt_split(w.ctypes.data, len(w), 4, bins.ctypes.data)
    assert (bins == ref).all()
    assert lib.abea_lpt_split(None, 3, 2, bins.ctypes.data) != 0


def test_walk_code_expansion_matches_the_traceback(orc, r9):
    """abea_expand_walk_codes (what the host entry's un-flatten does with the 2-bit walk that crosses PCIe) rebuilds
    the oracle's pair list from the moves of that list, for real alignments and for hand-made corner cases."""
    import ctypes
    from f5c_amd import abea, synth
    from f5c_amd.types import PAIR_DT
    k, model = r9
    lib = abea.load_library()
    lib.abea_expand_walk_codes.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]

    def roundtrip(pairs):
        p = np.asarray(pairs).view(np.int32).reshape(-1, 2).astype(np.int64)
        n = len(p)
        # walk order = last pair first; code of step j = the move from pair n-1-j to pair n-2-j
        dk = p[1:, 0] - p[:-1, 0]; de = p[1:, 1] - p[:-1, 1]
        assert ((dk | de) == 1).all() and (dk >= 0).all() and (de >= 0).all()
        code = np.where((dk == 1) & (de == 1), 0, np.where(de == 1, 1, 2))[::-1]      # 0 diagonal, 1 up, 2 left
        code = np.concatenate([code, [3]])                     # the last step's code is never applied to a pair
        words = np.zeros((n + 15) // 16 + 1, dtype=np.uint32)
        for j, c in enumerate(code):
            words[j >> 4] |= np.uint32(int(c) << (2 * (j & 15)))
        out = np.zeros(n + 2, dtype=PAIR_DT)
        out["ref_pos"][-1] = -7                                # canary
        assert lib.abea_expand_walk_codes(words.ctypes.data, n, int(p[-1, 0]), int(p[-1, 1]), out.ctypes.data) == 0
        got = out.view(np.int32).reshape(-1, 2)
        assert (got[:n] == p).all() and got[n + 1, 0] == -7

    batch = synth.make_batch(12, model, k, seed=3, law=900, bad_frac=0.0)
    o_pairs, o_n, _ = orc.align_batch(batch, model, k, n_threads=2)
    done = 0
    for i in range(12):
        if o_n[i] > 0:
            s = int(batch["pair_ptr"][i])
            roundtrip(o_pairs[s:s + o_n[i]])
            done += 1
    assert done >= 8
    # the same walk expanded straight into base_to_event_map == postalign on the pair list (align.c:571-596)
    lib.abea_expand_walk_codes_to_map.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]

    def map_roundtrip(pairs, seq, ev, scale, shift):
        p = np.asarray(pairs).view(np.int32).reshape(-1, 2).astype(np.int64)
        n = len(p)
        dk = p[1:, 0] - p[:-1, 0]; de = p[1:, 1] - p[:-1, 1]
        code = np.concatenate([np.where((dk == 1) & (de == 1), 0, np.where(de == 1, 1, 2))[::-1], [3]])
        words = np.zeros((n + 15) // 16 + 1, dtype=np.uint32)
        for j, c in enumerate(code):
            words[j >> 4] |= np.uint32(int(c) << (2 * (j & 15)))
        K = int(p[-1, 0]) + 1
        out = np.full((K + 1, 2), -7, dtype=np.int32)
        assert lib.abea_expand_walk_codes_to_map(words.ctypes.data, n, K - 1, int(p[-1, 1]), out.ctypes.data) == 0
        want = orc.scaling_single(pairs, seq, ev, model, k, scale, shift)["base_to_event_map"]
        assert len(want) == K and (out[:K, 0] == want["start"]).all() and (out[:K, 1] == want["stop"]).all()
        assert (out[K] == -7).all()

    n_maps = 0
    for i in range(12):
        if o_n[i] > 0:
            s = int(batch["pair_ptr"][i]); rs, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
            es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
            map_roundtrip(o_pairs[s:s + o_n[i]], batch["reads"][rs:rs + L].tobytes(), batch["events"][es:es + E],
                          batch["scalings"]["scale"][i], batch["scalings"]["shift"][i])
            n_maps += 1
    assert n_maps >= 8
    # skips (k-mers whose only pair repeats the previous event keep {-1,-1}) and stays, by hand
    hand = np.array([[0, 0], [1, 0], [2, 0], [2, 1], [2, 2], [3, 3], [4, 3], [5, 4], [5, 5]], dtype=np.int32).view(PAIR_DT).ravel()
    p = hand.view(np.int32).reshape(-1, 2)
    dk = np.diff(p[:, 0]); de = np.diff(p[:, 1])
    code = np.concatenate([np.where((dk == 1) & (de == 1), 0, np.where(de == 1, 1, 2))[::-1], [0]])
    words = np.zeros(2, dtype=np.uint32)
    for j, c in enumerate(code):
        words[j >> 4] |= np.uint32(int(c) << (2 * (j & 15)))
    out = np.zeros((6, 2), dtype=np.int32)
    assert lib.abea_expand_walk_codes_to_map(words.ctypes.data, len(p), 5, 5, out.ctypes.data) == 0
    assert out.tolist() == [[0, 0], [-1, -1], [1, 2], [3, 3], [-1, -1], [4, 5]]
    assert lib.abea_expand_walk_codes_to_map(None, 3, 2, 2, out.ctypes.data) != 0

    def postalign_map(p):                                     # align.c:571-596 on an ascending pair list
        K = int(p[-1, 0]) + 1
        m = np.full((K, 2), -1, dtype=np.int32)
        prev = -1
        for kk, ee in p:
            if ee != prev:
                if m[kk, 0] == -1:
                    m[kk, 0] = ee
                m[kk, 1] = ee
            prev = ee
        return m

    rr = np.random.default_rng(77)
    for n in list(range(1, 40)) + [63, 64, 65, 95, 96, 97, 1000, 3001]:      # 32-code word boundaries of the per-k-mer scan
        for pw in ((0.6, 0.3, 0.1), (0.2, 0.1, 0.7), (0.1, 0.85, 0.05)):       # mostly diagonal / skips / long stays
            steps = rr.choice(3, size=n - 1, p=pw)
            p = np.zeros((n, 2), dtype=np.int64)
            for j, c in enumerate(steps):
                p[j + 1] = p[j] + [(1, 1), (0, 1), (1, 0)][c]
            code = np.concatenate([steps[::-1], [int(rr.integers(0, 4))]])     # the last code is never applied
            words = np.zeros((n + 15) // 16 + 2, dtype=np.uint32)
            for j, c in enumerate(code):
                words[j >> 4] |= np.uint32(int(c) << (2 * (j & 15)))
            K = int(p[-1, 0]) + 1
            out = np.full((K + 1, 2), -7, dtype=np.int32)
            assert lib.abea_expand_walk_codes_to_map(words.ctypes.data, n, K - 1, int(p[-1, 1]), out.ctypes.data) == 0
            assert (out[:K] == postalign_map(p)).all() and (out[K] == -7).all(), (n, pw)

    for n in (1, 2, 15, 16, 17, 31, 32, 33, 1024, 1025):       # word boundaries
        steps = np.random.default_rng(n).integers(0, 3, n - 1)
        p = np.zeros((n, 2), dtype=np.int32)
        for j, c in enumerate(steps):
            p[j + 1] = p[j] + [(1, 1), (0, 1), (1, 0)][c]
        roundtrip(p.copy().view(PAIR_DT).ravel())
    assert lib.abea_expand_walk_codes(None, 0, 0, 0, None) == 0 and lib.abea_expand_walk_codes(None, 3, 0, 0, None) != 0


def test_chunk_plan_of_the_host_entry(monkeypatch):
    """abea_host_plan_chunks = the carving abea_align_batch_host applies: every runnable read in exactly one chunk, size rules
    (>= 1024 reads and >= 24 M events, first two chunks a quarter / half, <= 16384 reads), the arena share respected, over-long
    reads alone, guard failures left out.  Launch order (round 5): an ascending ramp — every other read above 18 000 bands,
    shortest first — then everything else longest first; ABEA_HOST_ORDER=lpt = plain longest-first."""
    import ctypes
    from f5c_amd import abea, synth
    lib = abea.load_library()
    lib.abea_host_plan_chunks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint64,
                                          ctypes.c_void_p, ctypes.c_void_p]
    for var in ("ABEA_HOST_CHUNK_EVENTS", "ABEA_HOST_CHUNK_READS", "ABEA_HOST_CHUNK_READS_MAX", "ABEA_HOST_SLOTS", "ABEA_HOST_ORDER"):
        monkeypatch.delenv(var, raising=False)

    def plan(L, E, arena):
        L = np.ascontiguousarray(L, dtype=np.int32); E = np.ascontiguousarray(E, dtype=np.int32)
        out = np.full(len(L), -2, dtype=np.int32); n = ctypes.c_int32(-1)
        rc = lib.abea_host_plan_chunks(L.ctypes.data, E.ctypes.data, len(L), 6, arena, out.ctypes.data, ctypes.byref(n))
        return rc, out, n.value

    L = synth.batch_lengths(100_000, 20250003, "loguniform")            # BASELINE configs[2]
    E = 2 * L + (L % 7)
    E[::1000] = 20 * L[::1000]                                          # over-segmented: skipped by E/L >= 15
    skipped = np.zeros(len(L), bool); skipped[::1000] = True
    bands = E.astype(np.int64) + (L - 6 + 1) + 2
    for order in ("lpt", "ramp"):
        if order == "lpt":
            monkeypatch.setenv("ABEA_HOST_ORDER", "lpt")
        else:
            monkeypatch.delenv("ABEA_HOST_ORDER")
        rc, ch, n = plan(L, E, 150 << 30)
        assert rc == 0 and 40 <= n <= 80
        assert (ch[skipped] == -1).all() and (ch[~skipped] >= 0).all() and ch.max() == n - 1
        hi = np.array([bands[ch == c].max() for c in range(n)]); lo = np.array([bands[ch == c].min() for c in range(n)])
        if order == "lpt":
            assert (lo[:-1] >= hi[1:]).all()                            # longest first across chunks
            desc_from = 0
        else:
            # the ramp: chunks of ascending length holding every other read above 18 000 bands, then longest-first
            top = int(np.argmax(hi))
            assert 2 <= top <= n // 2 and hi[top] == bands[~skipped].max()
            assert (hi[:top - 1] <= lo[1:top]).all() and lo[0] >= 18000             # ascending up to the chunk holding the turn; nothing below the ramp's floor
            ramp_reads = np.isin(ch, np.arange(top + 1)) & ~skipped
            above = (bands >= 18000) & ~skipped
            assert abs(int(ramp_reads.sum()) - int(above.sum()) // 2) <= 8192       # ~half of the long reads (the chunk holding the turn is mixed)
            desc_from = top + 1
            assert (lo[desc_from:-1] >= hi[desc_from + 1:]).all()       # the LPT tail is untouched
            assert hi[-1] < 4000 and (ch == 0).sum() >= 256 and E[ch == 0].sum() < 20 << 20   # a cheap first chunk: the GPU starts after ~2 ms
        for c in range(n):
            m = ch == c
            cnt, ev = int(m.sum()), int(E[m].sum())
            ramp = 4 if c == 0 else 2 if c == 1 else 1
            if c < n - 1:
                assert cnt >= 1024 // ramp and (ev >= (24 << 20) // ramp or cnt == 16384)
            assert cnt <= 16384
            # closing rule (descending part): without its last (shortest) read the chunk would have been below one of the two thresholds
            last = np.nonzero(m)[0][np.argmin(bands[m])]
            if desc_from <= c < n - 1 and cnt < 16384:
                assert cnt - 1 < 1024 // ramp or ev - int(E[last]) < (24 << 20) // ramp
    # small batches keep plain longest-first: nothing to ramp with fewer than 8192 long reads
    rc, chs, ns = plan(L[:6000], 2 * L[:6000], 150 << 30)
    assert rc == 0
    b6 = 2 * L[:6000].astype(np.int64) + L[:6000]
    assert all(b6[chs == c].min() >= b6[chs == c + 1].max() for c in range(ns - 1))
    # a small arena: shares of 2 MiB; the long read runs alone, the others are packed under the share
    L2 = np.array([1500, 40000, 1200, 2500, 60000, 1800]); E2 = 2 * L2
    rc, ch2, n2 = plan(L2, E2, 16 << 20)
    assert rc == 0 and ch2[4] == 0 and ch2[1] == 1 and (ch2 == 0).sum() == 1 and (ch2 == 1).sum() == 1
    assert n2 >= 3 and set(ch2[[0, 2, 3, 5]]) <= set(range(2, n2))
    rc, _, _ = plan([250000], [500000], 16 << 20)
    assert rc != 0 and b"arena" in lib.abea_last_error()
    # the env knobs the GPU tests use
    monkeypatch.setenv("ABEA_HOST_CHUNK_READS", "4"); monkeypatch.setenv("ABEA_HOST_CHUNK_EVENTS", "20000")
    rc, ch3, n3 = plan(np.full(60, 1800), np.full(60, 3600), 4 << 30)
    assert rc == 0 and n3 >= 8 and (np.bincount(ch3)[2:-1] >= 4).all()


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["codes", "device"])
def test_both_pair_return_modes_many_chunks(ctx, orc, r9, monkeypatch, mode):
    """Pairs expanded on the host from the 2-bit walk (default) and pairs expanded + compacted on the device
    (ABEA_HOST_PAIRS=device) give the oracle's lists, over many small chunks and over one chunk."""
    from f5c_amd import synth
    k, model = r9
    batch = synth.make_batch(150, model, k, seed=91, law="loguniform", bad_frac=0.1,
                             lengths=np.exp(np.random.default_rng(1).uniform(np.log(300), np.log(9000), 150)).astype(int))
    ora = orc.align_batch(batch, model, k, n_threads=8)
    monkeypatch.setenv("ABEA_HOST_PAIRS", "device" if mode == "device" else "host")
    for chunk_reads, chunk_events in ((7, 30000), (1024, 24 << 20)):
        monkeypatch.setenv("ABEA_HOST_CHUNK_READS", str(chunk_reads))
        monkeypatch.setenv("ABEA_HOST_CHUNK_EVENTS", str(chunk_events))
        plist, n_pairs, diag = ctx.align_flat_host(batch)
        st = ctx.stats()
        assert st["n_sub_batches"] >= (10 if chunk_reads == 7 else 1)
        _check_host(batch, plist, n_pairs, ora)
        ran = (diag["flags"] & 3) == 0
        assert (diag["n_aligned"][ran] == ora[2]["n_aligned"][ran]).all()
        assert np.allclose(diag["sum_emission"][ran], ora[2]["sum_emission"][ran], rtol=0, atol=1e-4)
        assert st["sum_pairs"] == int(ora[1].sum())
        if mode == "codes":                                  # the walk is ~0.4 B per event on the wire, the pairs ~8
            assert st["d2h_bytes"] < 2 * int(batch["n_events"].sum()) + 200 * len(n_pairs) + (1 << 16) * st["n_sub_batches"]


@pytest.mark.gpu
@pytest.mark.parametrize("want_pairs", [True, False])
def test_fused_scaling_single_through_the_host_entry(ctx, orc, r9, monkeypatch, want_pairs):
    """align_db + scaling_db in one call (f5c.c:924-936): base_to_event_map, recalibrated scalings, events_per_base, flags
    and counts equal the oracle's scaling_single on the oracle's pairs; with pairs=NULL nothing but those comes back."""
    from f5c_amd import synth
    k, model = r9
    batch = synth.make_batch(60, model, k, seed=83, law="gamma8k", bad_frac=0.1)
    monkeypatch.setenv("ABEA_HOST_CHUNK_READS", "16")
    monkeypatch.setenv("ABEA_HOST_CHUNK_EVENTS", "200000")
    o_pairs, o_n, _ = orc.align_batch(batch, model, k, n_threads=8)
    flag_in = np.zeros(60, dtype=np.int32); flag_in[5] = 0x100          # unrelated bits survive
    plist, n_pairs, _, sc = ctx.align_flat_host(batch, scaling=True, want_pairs=want_pairs, read_stat_flag=flag_in)
    assert ctx.stats()["n_sub_batches"] >= 3
    assert (n_pairs == o_n).all()
    assert (plist is None) == (not want_pairs)
    n_cal = 0
    for i in range(60):
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        ps = int(batch["pair_ptr"][i])
        r = orc.scaling_single(o_pairs[ps:ps + o_n[i]], batch["reads"][s:s + L].tobytes(), batch["events"][es:es + E],
                               model, k, batch["scalings"]["scale"][i], batch["scalings"]["shift"][i])
        assert sc["read_stat_flag"][i] == (r["flag"] | flag_in[i]), i
        assert sc["n_event_alignment"][i] == r["n_alignment"] and sc["events_per_base"][i] == r["events_per_base"]
        m = sc["base_to_event_map"][i]
        if o_n[i] > 0:
            assert (m[:, 0] == r["base_to_event_map"]["start"]).all() and (m[:, 1] == r["base_to_event_map"]["stop"]).all()
            if not (r["flag"] & 1) or r["scalings"]["var"] != 0:
                assert sc["scalings"]["shift"][i] == r["scalings"]["shift"]
                assert sc["scalings"]["scale"][i] == r["scalings"]["scale"]
                assert sc["scalings"]["var"][i] == r["scalings"]["var"]
                # align.c:758-760 (CACHED_LOG): log_var = (float)log(double var) — round-4 advisor finding: it stayed log(1) = 0
                assert sc["scalings"]["log_var"][i] == r["scalings"]["log_var"] != 0
                n_cal += 1
            else:
                assert sc["scalings"]["log_var"][i] == batch["scalings"]["log_var"][i]
        else:
            assert (m == -1).all()                            # untouched: the reference leaves NULL there
            assert sc["scalings"]["scale"][i] == batch["scalings"]["scale"][i]
    assert n_cal >= 40


@pytest.mark.gpu
def test_nsample_zero_reads_are_skipped(ctx, orc, r9):
    """align_single's first guard (f5c.c:812,826-828): db->sig[i]->nsample == 0 -> n_event_align_pairs = 0, nothing
    written; the other reads are unaffected."""
    from f5c_amd import synth
    k, model = r9
    batch = synth.make_batch(30, model, k, seed=93, law=1500, bad_frac=0.0)
    ora = orc.align_batch(batch, model, k, n_threads=4)
    ns = np.full(30, 6000, dtype=np.int64); ns[[0, 7, 29]] = 0
    seqs, evs = [], []
    for i in range(30):
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        seqs.append(batch["reads"][s:s + L].tobytes()); evs.append(batch["events"][es:es + E])
    plist, n_pairs, diag = ctx.align_db_host(seqs, evs, batch["scalings"], n_samples=ns)
    assert (n_pairs[[0, 7, 29]] == 0).all() and (diag["flags"][[0, 7, 29]] & 1).all()
    keep = np.setdiff1d(np.arange(30), [0, 7, 29])
    assert (n_pairs[keep] == ora[1][keep]).all() and (ora[1][keep] > 0).all()
    for i in keep:
        s = int(batch["pair_ptr"][i])
        assert (plist[i] == ora[0][s:s + ora[1][i]]).all()


@pytest.mark.gpu
def test_error_exit_leaves_nothing_in_flight(orc, r9, monkeypatch):
    """A read that cannot fit the arena fails the call before anything is launched, and the same context then runs
    the next batch correctly (regression: stale chunk bookkeeping used to be un-flattened into the next batch)."""
    from f5c_amd import abea, synth
    k, model = r9
    monkeypatch.setenv("ABEA_HOST_CHUNK_READS", "4")
    monkeypatch.setenv("ABEA_HOST_CHUNK_EVENTS", "10000")
    good = synth.make_batch(40, model, k, seed=95, law=1200, bad_frac=0.05)
    ora = orc.align_batch(good, model, k, n_threads=4)
    bad = synth.make_batch(13, model, k, seed=96, lengths=[1200] * 12 + [250000], bad_frac=0.0)
    with abea.AbeaContext(model, k, max_arena_bytes=20 << 20) as c:
        for _ in range(2):
            with pytest.raises(abea.AbeaError, match="arena"):
                c.align_flat_host(bad)
            plist, n_pairs, _ = c.align_flat_host(good)
            _check_host(good, plist, n_pairs, ora)


@pytest.mark.gpu
def test_multi_device_context_two_contexts_on_one_gpu(orc, r9, monkeypatch):
    """abea_init_multi: one process, several device contexts; every batch is LPT-split inside the library and each
    share runs its own pipeline from its own host thread.  On a 1-GPU box the device is listed twice."""
    from f5c_amd import abea, synth
    k, model = r9
    batch = synth.make_batch(120, model, k, seed=97, bad_frac=0.08,
                             lengths=np.exp(np.random.default_rng(2).uniform(np.log(400), np.log(12000), 120)).astype(int))
    ora = orc.align_batch(batch, model, k, n_threads=8)
    monkeypatch.setenv("ABEA_HOST_CHUNK_READS", "8")
    monkeypatch.setenv("ABEA_HOST_CHUNK_EVENTS", "50000")
    with abea.AbeaContext(model, k, device_ids=[0, 0], max_arena_bytes=1 << 30) as c:
        assert c.device_count() == 2
        c.selftest()
        for scaling in (False, True):
            res = c.align_flat_host(batch, scaling=scaling)
            _check_host(batch, res[0], res[1], ora)
            st = c.stats()
            assert st["n_devices"] == 2 and st["n_reads_gpu"] + st["n_reads_skipped"] == 120
            per = [c.device_stats(d) for d in range(2)]
            assert all(p["n_reads_gpu"] > 0 for p in per)
            assert sum(p["sum_events"] for p in per) == st["sum_events"]
            bands = [p["sum_bands"] for p in per]
            assert max(bands) / (sum(bands) / 2) < 1.05       # LPT balance on the band count
        with pytest.raises(abea.AbeaError, match="single-device"):
            c.align_db_device(abea.AbeaContext.upload(batch))
    # the single-device result is identical
    with abea.AbeaContext(model, k, device_ids=[0], max_arena_bytes=1 << 30) as c1:
        assert c1.device_count() == 1
        plist, n_pairs, _ = c1.align_flat_host(batch)
        _check_host(batch, plist, n_pairs, ora)


def _run_shim(tmp_path, batch, model, k, env):
    import struct
    from f5c_amd import abea
    exe = str(tmp_path / "shim_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++11", "-O2", os.path.join(ROOT, "tests", "shim_driver.cpp"), "-o", exe,
                               "-L", os.path.dirname(abea.LIB_PATH), "-labea_hip",
                               "-Wl,-rpath," + os.path.dirname(os.path.abspath(abea.LIB_PATH))])
    n = len(batch["read_len"])
    blob = struct.pack("<4i", n, k, len(model), 0) + model.tobytes() + batch["read_len"].tobytes() + \
        batch["n_events"].tobytes() + batch["scalings"].tobytes()
    for i in range(n):
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        blob += batch["reads"][s:s + L].tobytes()
    blob += batch["events"].tobytes()
    (tmp_path / "batch.bin").write_bytes(blob)
    e = dict(os.environ); e.update(env)
    subprocess.check_call([exe, str(tmp_path / "batch.bin"), str(tmp_path / "out.txt")], env=e)
    lines = (tmp_path / "out.txt").read_text().splitlines()
    got = []
    for ln in lines:
        f = ln.split("\t")
        got.append([tuple(int(v) for v in t.strip("{}").split(",")) for t in f[1:] if t])
    return got


@pytest.mark.gpu
def test_cpp_caller_nsample_multi_device_and_fused(orc, r9, tmp_path):
    """The g++-built process_db-shaped caller through include/abea_f5c_shim.h with (a) bad reads (nsample == 0),
    (b) a device LIST (two contexts on device 0), (c) align_db + scaling_db fused (abea_f5c_align_scale)."""
    from f5c_amd import synth
    k, model = r9
    batch = synth.make_batch(24, model, k, seed=72, law=1600, bad_frac=0.1)
    o_pairs, o_n, _ = orc.align_batch(batch, model, k, n_threads=4)

    def expect(i):
        s = int(batch["pair_ptr"][i])
        return [tuple(int(v) for v in p) for p in o_pairs[s:s + o_n[i]]]

    got = _run_shim(tmp_path, batch, model, k, {"SHIM_NSAMPLE0": "1,5", "SHIM_DEVS": "0,0"})
    for i in range(24):
        assert got[i] == ([] if i in (1, 5) else expect(i)), i
    got = _run_shim(tmp_path, batch, model, k, {"SHIM_ASYNC": "1", "SHIM_NSAMPLE0": "3"})      # abea_f5c_align_submit / _wait
    for i in range(24):
        assert got[i] == ([] if i == 3 else expect(i)), i
    got = _run_shim(tmp_path, batch, model, k, {"SHIM_FUSED": "1"})
    for i in range(24):
        assert got[i] == expect(i), i
    for ln in (tmp_path / "out.txt.scale").read_text().splitlines():
        f = ln.split("\t")
        i = int(f[0])
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        ps = int(batch["pair_ptr"][i])
        r = orc.scaling_single(o_pairs[ps:ps + o_n[i]], batch["reads"][s:s + L].tobytes(), batch["events"][es:es + E],
                               model, k, batch["scalings"]["scale"][i], batch["scalings"]["shift"][i])
        assert int(f[1]) == r["flag"] and int(f[2]) == r["n_alignment"] and float.fromhex(f[3]) == r["events_per_base"]
        if o_n[i] > 0:
            m = np.array([[int(v) for v in t.split(",")] for t in f[7].split()], dtype=np.int32)
            assert (m[:, 0] == r["base_to_event_map"]["start"]).all() and (m[:, 1] == r["base_to_event_map"]["stop"]).all()
            if not (r["flag"] & 1) or r["scalings"]["var"] != 0:
                assert np.float32(float.fromhex(f[4])) == r["scalings"]["shift"]
                assert np.float32(float.fromhex(f[5])) == r["scalings"]["scale"]
                assert np.float32(float.fromhex(f[6])) == r["scalings"]["var"]
        else:
            assert f[7].strip() == "NULL"                      # f5c.c:787 base_to_event_map[i] = NULL


@pytest.mark.gpu
def test_bench_two_ranks_and_single_process_two_contexts(tmp_path):
    """bench.py's N>1 paths on the GPU box: two ranks under torch.distributed.run (strong scaling of ONE batch, each
    rank generating only its LPT shard; gloo + both ranks on device 0 because the box has one GPU), and one process
    driving two device contexts through abea_init_multi.  Both must account for every event of the whole batch."""
    import json
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "r9_10k_8kb", "--reads", "1500", "--steps", "2",
            "--warmup", "1", "--one-device", "--arena-gib", "8", "--no-cpu-baseline", "--no-small-batch", "--mode", "host"]
    one = subprocess.run(base + ["--gpus", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads(one.stdout.strip().splitlines()[-1])
    port = 29600 + os.getpid() % 300
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port)] + base[1:] +
                         ["--gpus", "2", "--backend", "gloo"], capture_output=True, text=True, env=env, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    j2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    sp = subprocess.run(base + ["--gpus", "2", "--single-process"], capture_output=True, text=True, env=env, timeout=600)
    assert sp.returncode == 0, sp.stderr[-2000:]
    j3 = json.loads(sp.stdout.strip().splitlines()[-1])
    for j in (j2, j3):
        assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
        assert j["config"]["events"] == ref["config"]["events"] and j["config"]["reads"] == 1500
        assert abs(j["qc_pass_frac"] - ref["qc_pass_frac"]) < 1e-9
    assert j2["config"]["events_rank0"] < 0.6 * ref["config"]["events"]          # rank 0 holds its shard only
    assert j3["host_to_host"]["devices"] == 2


@pytest.mark.gpu
def test_bench_under_torchrun_world_size_1_runs_rccl(tmp_path):
    """The scaling runs are the driver's (no multi-GPU box here), but their communication path must have executed at least
    once: bench.py under torch.distributed.run with ONE rank builds the `nccl` (= RCCL) process group and runs its
    statistics all_gather on `cuda`; the line carries the per-rank host breakdown and the host/gpu verdict."""
    import json
    import sys
    port = 29900 + os.getpid() % 90
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--config", "r9_10k_8kb", "--reads", "1500", "--steps", "2", "--warmup", "1",
                        "--arena-gib", "8", "--no-cpu-baseline", "--no-small-batch", "--backend", "nccl"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["collective"] == {"backend": "nccl", "world": 1, "gather_device": "cuda",
                               "note": "statistics all_gather only; no data-path collective (reads are independent)"}
    import bench                                                  # the literals are bench.py's: BOUND_VALUES, not a copy of them
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["bound"] in bench.BOUND_VALUES
    assert len(j["per_rank"]) == 1 and j["per_rank"][0]["rank"] == 0 and j["per_rank"][0]["host_threads"] >= 1
    assert j["per_rank"][0]["events"] == j["config"]["events"]
    assert set(j["per_rank"][0]["host_ms_per_step"]) == {"flatten", "unflatten", "wait_for_gpu"}
    assert j["roofline"]["target_frac"] == 0.40 and j["roofline"]["target_met"] is False
    vr = j["roofline"]["valu_roofline"]                          # 1500 reads fill a third of the wave slots: far below the ceiling
    assert vr["class_floor"]["right_move_bands"] > 0 and vr["measured_ms"] > 0
    assert 0.0 < vr["class_floor"]["frac"] < 1.0 and vr["frac"] == vr["class_floor"]["frac"]


def test_kmer_count_expansion_matches_postalign(orc, r9):
    """abea_expand_kmer_counts_to_map (what the host entry's un-flatten does with the one-byte-per-k-mer form in which the fused
    scaling_single phase sends base_to_event_map over PCIe): the entries of the map tile the path's events in k order, so the
    event count of every entry determines the map.  Against the oracle's postalign on aligned reads, plus the escape value."""
    import ctypes
    from f5c_amd import abea, synth
    k, model = r9
    lib = abea.load_library()
    lib.abea_expand_kmer_counts_to_map.restype = ctypes.c_int
    lib.abea_expand_kmer_counts_to_map.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    batch = synth.make_batch(40, model, k, seed=812, law=1100, bad_frac=0.0)
    n_ok = 0
    for i in range(40):
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        seq, ev = batch["reads"][s:s + L].tobytes(), batch["events"][es:es + E]
        if i % 5 == 0:                                               # long stays: k-mers with many events
            ev = np.concatenate([ev[:50], np.repeat(ev[50:51], 40), ev[50:]])
        sc = batch["scalings"][i]
        pairs, d = orc.align(seq, ev, model, k, sc["scale"], sc["shift"])
        if len(pairs) == 0:
            continue
        m = orc.scaling_single(pairs, seq, ev, model, k, sc["scale"], sc["shift"])["base_to_event_map"]
        K = L - k + 1
        cnt = np.where(m["start"] >= 0, m["stop"] - m["start"] + 1, 0)
        assert cnt.max() < 255 and cnt.sum() == pairs["read_pos"][-1] - pairs["read_pos"][0] + 1     # the entries tile the events
        c8 = cnt.astype(np.uint8)
        out = np.zeros((K, 2), dtype=np.int32)
        assert lib.abea_expand_kmer_counts_to_map(c8.ctypes.data, K, int(d["best_event"]), out.ctypes.data) == 0
        assert (out[:, 0] == m["start"]).all() and (out[:, 1] == m["stop"]).all()
        c8[K // 2] = 255                                             # "255 or more": refused, the caller falls back to the walk
        assert lib.abea_expand_kmer_counts_to_map(c8.ctypes.data, K, int(d["best_event"]), out.ctypes.data) != 0
        n_ok += 1
    assert n_ok >= 30


@pytest.mark.gpu
def test_fused_map_with_kmers_of_255_and_more_events(ctx, orc, r9):
    """The fused call hands base_to_event_map to the host as one event-count byte per k-mer; 255 means "255 or more" and sends
    that read's map back to the 2-bit walk.  Reads with a stretch of 240..600 repeated events on one k-mer (they align: a long
    stay) sit on both sides of the escape value; every map equals the oracle's postalign."""
    from f5c_amd import synth
    k, model = r9
    base = synth.make_batch(8, model, k, seed=4242, law=1500, bad_frac=0.0)
    seqs, evs, scal, reps = [], [], [], [0, 240, 250, 253, 256, 300, 600, 252]
    for i in range(8):
        s, L = int(base["read_ptr"][i]), int(base["read_len"][i])
        es, E = int(base["event_ptr"][i]), int(base["n_events"][i])
        seq, ev = base["reads"][s:s + L].tobytes(), base["events"][es:es + E]
        if reps[i]:
            ev = np.concatenate([ev[:1000], np.repeat(ev[1000:1001], reps[i]), ev[1000:]])
        seqs.append(seq); evs.append(np.ascontiguousarray(ev)); scal.append(orc.estimate_scalings(seq, model, k, ev))
    batch = synth.batch_from_reads(seqs, evs, scal)
    plist, n_pairs, _, sc = ctx.align_flat_host(batch, scaling=True, want_pairs=False)
    most = []
    for i in range(8):
        o_pairs, _ = orc.align(seqs[i], evs[i], model, k, scal[i][0], scal[i][1])
        assert len(o_pairs) == n_pairs[i] > 0
        r = orc.scaling_single(o_pairs, seqs[i], evs[i], model, k, scal[i][0], scal[i][1])
        m = sc["base_to_event_map"][i]
        assert (m[:, 0] == r["base_to_event_map"]["start"]).all() and (m[:, 1] == r["base_to_event_map"]["stop"]).all(), i
        assert sc["read_stat_flag"][i] == r["flag"] and sc["events_per_base"][i] == r["events_per_base"]
        assert sc["scalings"]["shift"][i] == r["scalings"]["shift"] and sc["scalings"]["var"][i] == r["scalings"]["var"]
        assert sc["scalings"]["log_var"][i] == r["scalings"]["log_var"]
        bm = r["base_to_event_map"]
        most.append(int(np.where(bm["start"] >= 0, bm["stop"] - bm["start"] + 1, 0).max()))
    assert max(most) >= 500 and sum(c >= 255 for c in most) >= 2 and sum(200 < c < 255 for c in most) >= 1, most


def test_the_walk_code_rules_of_phase_3_are_postalign(orc, r9):
    """Phase 3 of abea_align_kernel writes base_to_event_map straight from the walk (no pair lists in HBM for the fused call): in
    walk order code[t] is the move from pair t to its predecessor in the list (1 = same k-mer, 2 = same event), and a pair
        opens its k-mer's run  iff code[t] != 1 (or it is the last step),   closes it  iff code[t-1] != 1 (or t = 0),
        is a new event         iff code[t] != 2 (or it is the last step),   and the successor's event is e + [code[t-1] != 2].
    The same rules in numpy, on oracle alignments, against the oracle's postalign (align.c:571-596)."""
    from f5c_amd import synth
    k, model = r9
    batch = synth.make_batch(30, model, k, seed=577, law=1300, bad_frac=0.0)
    n_ok = 0
    for i in range(30):
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        seq, ev = batch["reads"][s:s + L].tobytes(), batch["events"][es:es + E]
        if i % 4 == 0:
            ev = np.concatenate([ev[:300], np.repeat(ev[300:301], 30), ev[300:]])       # a long stay
        if i % 4 == 1:
            ev = np.concatenate([ev[:200], ev[260:]])                                    # dropped events: skips
        sc = orc.estimate_scalings(seq, model, k, ev)
        pairs, _ = orc.align(seq, ev, model, k, sc[0], sc[1])
        if len(pairs) == 0:
            continue
        want = orc.scaling_single(pairs, seq, ev, model, k, sc[0], sc[1])["base_to_event_map"]
        W = pairs.view(np.int32).reshape(-1, 2)[::-1]                                    # walk order: t = 0 at the end cell
        n, K = len(W), L - k + 1
        dk, de = W[:-1, 0] - W[1:, 0], W[:-1, 1] - W[1:, 1]
        code = np.concatenate([np.where((dk == 1) & (de == 1), 0, np.where(dk == 0, 1, 2)), [3]])   # the last step's code is never used
        t = np.arange(n)
        cprev = np.concatenate([[0], code[:-1]])
        run_start = (t == n - 1) | (code != 1)
        run_end = (t == 0) | (cprev != 1)
        is_new = (t == n - 1) | (code != 2)
        next_e = W[:, 1] + ((t > 0) & (cprev != 2))
        start = np.full(K, -1); stop = np.full(K, -1)
        start[W[run_start, 0]] = np.where(is_new, W[:, 1], np.where(~run_end, next_e, -1))[run_start]
        stop[W[run_end, 0]] = np.where(~run_start | is_new, W[:, 1], -1)[run_end]
        assert (start == want["start"]).all() and (stop == want["stop"]).all(), i
        n_ok += 1
    assert n_ok >= 24


def test_walk_code_expansion_every_length_and_alignment():
    """abea_expand_walk_codes (AVX2 + BMI2 blocks of 16 steps between a scalar head that aligns the block stores to 32 bytes and
    a scalar tail; plain loop on a CPU without them or with ABEA_HOST_SCALAR_EXPAND set): every list length from 1 step, every
    placement of the output buffer, against a step-by-step reference; nothing is written outside the list."""
    import ctypes
    from f5c_amd import abea
    lib = abea.load_library()
    lib.abea_expand_walk_codes.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    r = np.random.default_rng(5)
    for n in list(range(1, 100)) + [255, 256, 257, 1023, 1024, 4097]:
        steps = r.choice([0, 0, 0, 1, 1, 2], size=n).astype(np.uint32)
        words = np.zeros((n + 15) // 16 + 1, dtype=np.uint32)
        for j in range(16):
            sl = steps[j::16]
            words[:len(sl)] |= sl << (2 * j)
        dk, de = (steps != 1).astype(np.int64), (steps != 2).astype(np.int64)
        k0, e0 = 70000 + n, 90000 + n
        want = np.stack([k0 - np.concatenate([[0], np.cumsum(dk)[:-1]]), e0 - np.concatenate([[0], np.cumsum(de)[:-1]])], axis=1)[::-1]
        for off in range(4):
            buf = np.full((n + off + 6, 2), -7, dtype=np.int32)
            out = buf[off:off + n]
            assert lib.abea_expand_walk_codes(words.ctypes.data, n, k0, e0, out.ctypes.data) == 0
            assert (out == want).all(), (n, off)
            assert (buf[:off] == -7).all() and (buf[off + n:] == -7).all(), (n, off)


def test_kmer_count_expansion_every_length_and_alignment():
    """abea_expand_kmer_counts_to_map (AVX2 blocks of eight entries between scalar head and tail; plain loop otherwise): every
    table length, every placement of the output, counts up to 254, the escape value anywhere -> refused; nothing written outside."""
    import ctypes
    from f5c_amd import abea
    lib = abea.load_library()
    lib.abea_expand_kmer_counts_to_map.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    r = np.random.default_rng(9)
    for K in list(range(1, 70)) + [255, 256, 1000, 4099]:
        for off in range(4):
            c = np.where(r.random(K) < 0.1, 0, r.integers(1, 12, K)).astype(np.uint8)
            if K > 20:
                c[K // 3] = 254
            end = int(c.sum()) + 1000
            stop = end - (np.cumsum(c[::-1].astype(np.int64))[::-1] - c)
            want = np.stack([np.where(c > 0, stop - c + 1, -1), np.where(c > 0, stop, -1)], axis=1)
            buf = np.full((K + off + 6, 2), -7, dtype=np.int32)
            out = buf[off:off + K]
            assert lib.abea_expand_kmer_counts_to_map(c.ctypes.data, K, end, out.ctypes.data) == 0
            assert (out == want).all(), (K, off)
            assert (buf[:off] == -7).all() and (buf[off + K:] == -7).all(), (K, off)
            c[r.integers(0, K)] = 255
            assert lib.abea_expand_kmer_counts_to_map(c.ctypes.data, K, end, out.ctypes.data) != 0


def test_expansion_sweeps_through_the_scalar_fallback():
    """The expansion path (AVX2 / plain loop) is latched once per process, so on an AVX2 host the in-process sweeps above only ever
    run the vector code (round-4 advisor finding): re-run both in a child with ABEA_HOST_SCALAR_EXPAND=1."""
    import sys
    env = dict(os.environ, ABEA_HOST_SCALAR_EXPAND="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k",
                        "every_length_and_alignment or kmer_count_expansion_matches_postalign or walk_code_expansion_matches"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_flatten_loop_every_length_prefetch_and_hint():
    """abea_flatten_event_means = the flatten loop of the host entry: means[e] = events[e].mean for every table length, with and
    without the software prefetch (distances beyond the table included), every hint; nothing written past the table."""
    import ctypes
    from f5c_amd import abea
    from f5c_amd.types import EVENT_DT
    lib = abea.load_library()
    lib.abea_flatten_event_means.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    r = np.random.default_rng(11)
    for E in list(range(0, 40)) + [255, 256, 1000, 4099]:
        ev = np.zeros(E, dtype=EVENT_DT)
        ev["mean"] = r.normal(90, 12, E).astype(np.float32); ev["start"] = np.arange(E); ev["stdv"] = 1.5; ev["length"] = 9
        out = np.zeros(E + 12, dtype=np.float32)                       # numpy data is 16-byte aligned
        for pf in (0, 64, 1536, 65536):
            for hint in (0, 1, 2):
                out[:] = -1
                assert lib.abea_flatten_event_means(ev.ctypes.data, E, out.ctypes.data, pf, hint) == 0
                assert (out[:E] == ev["mean"]).all() and (out[E:] == -1).all(), (E, pf, hint)
    assert lib.abea_flatten_event_means(ev.ctypes.data, 4, out.ctypes.data + 4, 0, 0) != 0   # misaligned destination: refused
