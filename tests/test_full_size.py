"""Full-size parity for BASELINE.json configs[2] (100k R9.4.1 reads, 1-50 kb) and configs[4] (50k reads, 9-mer model),
through the host entry the bench times: the whole batch on the GPU and EVERY read of it compared with the CPU oracle —
pair count, a 64-bit position-dependent hash of the pair list (tests/pairhash.py), traceback length, end event and the
emission sum — through the per-read goldens minted from the oracle in the build container
(tests/golden/config_goldens_*.npz, tests/golden/make_config_goldens.py; round 4: no read of a BASELINE config is
property-checked only).  Kept beside it: a live oracle run on a random sample of 2000 reads (pair lists compared element by
element — a cross-check of the hash route), the size-independent properties on every read, and a bit-identical second run."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def check_against_config_goldens(config, batch, pairs, n_pairs, diag):
    """Every read of a synthetic BASELINE config against the oracle's committed per-read record."""
    from pairhash import hash_pair_lists_fast
    g = np.load(os.path.join(ROOT, "tests", "golden", f"config_goldens_{config}.npz"))
    n = len(g["n_pairs"])
    assert len(n_pairs) == n
    assert (batch["n_events"] == g["n_events"]).all() and (batch["read_len"] == g["read_len"]).all()   # the same batch
    bad = np.nonzero(n_pairs != g["n_pairs"])[0]
    assert len(bad) == 0, ("n_pairs", bad[:10])
    h = hash_pair_lists_fast(pairs, batch["pair_ptr"], n_pairs)       # tests/pairhash.c; == the numpy definition (test_config_goldens.py)
    bad = np.nonzero(h != g["pair_hash"])[0]
    assert len(bad) == 0, ("pair list hash", bad[:10])
    if diag is not None:
        ran = (diag["flags"] & 3) == 0
        assert (diag["n_aligned"][ran] == g["n_aligned"][ran]).all()
        assert (diag["best_event"][ran] == g["best_event"][ran]).all()
        assert np.allclose(diag["sum_emission"][ran], g["sum_emission"][ran], rtol=0, atol=1e-4)       # north_star tolerance
        assert (diag["sum_emission"][ran] == g["sum_emission"][ran]).mean() > 0.999                    # observed: equal
    return int((g["n_pairs"] > 0).sum())


def _full_size(config, sample_reads, orc):
    from f5c_amd import abea, synth, load_model_f32, synthetic_model
    cfg = synth.CONFIGS[config]
    k = cfg["k"]
    model = (load_model_f32(os.path.join(ROOT, "tests", "golden", "r9.4_450bps.6mer.f32"))[1] if k == 6
             else synthetic_model(k, seed=9))
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    batch = synth.make_batch(cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=workers)
    n = len(batch["read_len"])
    ctx = abea.AbeaContext(model, k, max_arena_bytes=150 << 30)
    try:
        view = ctx.host_view(batch, want_diag=True)
        ctx.align_view(view)
        st = ctx.stats()
        assert st["n_reads_gpu"] + st["n_reads_skipped"] == n and st["n_sub_batches"] >= 8
        n_pairs = view["n_pairs"].copy()
        pairs = view["pairs"].view(np.int32).reshape(-1, 2)
        assert (n_pairs > 0).mean() > 0.95                           # QC-pass fraction
        assert st["sum_pairs"] == int(n_pairs.sum())
        # ---- every read: ascending, spanning, bounded ----
        K = batch["read_len"].astype(np.int64) - k + 1
        E = batch["n_events"].astype(np.int64)
        assert (n_pairs <= E + K).all()
        digest = np.zeros(2, dtype=np.uint64)
        for i in range(n):
            m = int(n_pairs[i])
            if m == 0:
                continue
            s = int(batch["pair_ptr"][i])
            seg = pairs[s:s + m]
            assert seg[0, 0] == 0 and seg[-1, 0] == K[i] - 1, i      # spanned: first k-mer to last k-mer (align.c:529)
            assert seg[0, 1] >= 0 and seg[-1, 1] < E[i], i
            if m > 1:
                step = np.diff(seg, axis=0)
                assert step.min() >= 0 and step.max() <= 1 and (step.sum(axis=1) > 0).all(), i   # one DP move per pair
            digest[0] += np.uint64(int(seg[:, 0].sum(dtype=np.int64)) & 0xFFFFFFFFFFFF)
            digest[1] ^= np.uint64(int(seg[:, 1].sum(dtype=np.int64)) * (i + 1) & 0xFFFFFFFFFFFF)
        # ---- EVERY read against the oracle's per-read goldens ----
        n_ok = check_against_config_goldens(config, batch, view["pairs"], n_pairs, view["diag"])
        assert n_ok == int((n_pairs > 0).sum())
        # ---- the oracle, live, on a random sample (element-wise: cross-checks the hash route) ----
        idx = np.sort(np.random.default_rng(2024).choice(n, sample_reads, replace=False))
        sub = synth.take_reads(batch, idx)
        o_pairs, o_n, o_diag = orc.align_batch(sub, model, k, n_threads=workers)
        assert (n_pairs[idx] == o_n).all(), np.nonzero(n_pairs[idx] != o_n)[0][:10]
        for j, i in enumerate(idx):
            a, b, m = int(batch["pair_ptr"][i]), int(sub["pair_ptr"][j]), int(o_n[j])
            assert (view["pairs"][a:a + m] == o_pairs[b:b + m]).all(), f"read {i} pair list differs"
        dg = view["diag"][idx]
        ran = (dg["flags"] & 3) == 0
        assert (dg["n_aligned"][ran] == o_diag["n_aligned"][ran]).all()
        assert (dg["best_event"][ran] == o_diag["best_event"][ran]).all()
        assert np.allclose(dg["sum_emission"][ran], o_diag["sum_emission"][ran], rtol=0, atol=1e-4)   # north_star tolerance
        assert (o_n > 0).sum() >= 0.9 * sample_reads
        # ---- a second run into zeroed buffers is identical (no stale scratch, no order dependence between chunks) ----
        view["pairs"].fill(0)
        view["n_pairs"].fill(-1)
        ctx.align_view(view)
        assert (view["n_pairs"] == n_pairs).all()
        digest2 = np.zeros(2, dtype=np.uint64)
        for i in range(n):
            m = int(n_pairs[i])
            if m == 0:
                continue
            s = int(batch["pair_ptr"][i])
            seg = pairs[s:s + m]
            digest2[0] += np.uint64(int(seg[:, 0].sum(dtype=np.int64)) & 0xFFFFFFFFFFFF)
            digest2[1] ^= np.uint64(int(seg[:, 1].sum(dtype=np.int64)) * (i + 1) & 0xFFFFFFFFFFFF)
        assert (digest == digest2).all()
    finally:
        ctx.close()


def test_baseline_config3_100k_mixed_full_size(orc):
    """BASELINE.json configs[2]: 100 000 reads, 1-50 kb log-uniform, seed 20250003 (~2.5 G events)."""
    _full_size("r9_100k_mixed", 2000, orc)


def test_baseline_config5_r10_9mer_full_size(orc):
    """BASELINE.json configs[4]: 50 000 reads, mean 10 kb, 9-mer model of 262 144 entries (synthetic table)."""
    _full_size("r10_50k_10kb", 2000, orc)
