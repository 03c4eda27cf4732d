"""CPU: the per-read goldens of the synthetic BASELINE configs (tests/golden/config_goldens_*.npz, minted by
tests/golden/make_config_goldens.py from the oracle over EVERY read) are what the oracle says today on a slice of each
config, the batch they describe is the one the seeded generator builds, and the pair-list hash notices what it must."""
import os
import numpy as np
import pytest

from pairhash import hash_pair_lists, hash_pair_lists_fast

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,lo", [("r9_10k_8kb", 7000), ("r9_100k_mixed", 61000), ("r10_50k_10kb", 33000)])
def test_goldens_are_the_oracle_on_a_slice(orc, r9, config, lo):
    from f5c_amd import synth, synthetic_model
    path = os.path.join(ROOT, "tests", "golden", f"config_goldens_{config}.npz")
    if not os.path.exists(path):
        pytest.skip("not minted yet")
    g = np.load(path)
    cfg = synth.CONFIGS[config]
    k = cfg["k"]
    model = r9[1] if k == 6 else synthetic_model(k, seed=9)
    assert len(g["n_pairs"]) == cfg["n_reads"]
    L = synth.batch_lengths(cfg["n_reads"], cfg["seed"], cfg["law"])
    assert (g["read_len"] == L).all()                                  # the whole batch's length draw
    idx = np.arange(lo, lo + 40)
    b = synth.make_batch(cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], subset=idx)
    assert (b["n_events"] == g["n_events"][idx]).all()
    pairs, n_pairs, diag = orc.align_batch(b, model, k, n_threads=4)
    assert (n_pairs == g["n_pairs"][idx]).all()
    assert (diag["n_aligned"] == g["n_aligned"][idx]).all() and (diag["best_event"] == g["best_event"][idx]).all()
    assert (diag["sum_emission"] == g["sum_emission"][idx]).all()
    h = hash_pair_lists(pairs, b["pair_ptr"], n_pairs)
    assert (h == g["pair_hash"][idx]).all()
    assert (hash_pair_lists(pairs, b["pair_ptr"], n_pairs, block=5000) == h).all()      # independent of the blocking
    assert (hash_pair_lists_fast(pairs, b["pair_ptr"], n_pairs, threads=3) == h).all()   # the C twin the full-size tests use
    assert (g["n_pairs"] > 0).mean() > 0.95 and (g["pair_hash"][g["n_pairs"] == 0] == 0).all()


def test_hash_notices_swaps_shifts_and_truncation():
    r = np.random.default_rng(3)
    n = np.array([5, 0, 7, 1], dtype=np.int32)
    ptr = np.array([0, 9, 12, 30], dtype=np.int64)
    pairs = np.zeros((40, 2), dtype=np.int32)
    for i in range(4):
        pairs[ptr[i]:ptr[i] + n[i]] = np.cumsum(1 + r.integers(0, 2, (n[i], 2)), axis=0) + [0, 3 * i + 1]
    h = hash_pair_lists(pairs, ptr, n)
    assert h[1] == 0 and len(set(h.tolist())) == 4
    p = pairs.copy(); p[[0, 1]] = p[[1, 0]]
    assert pairs[0].tolist() != pairs[1].tolist() and hash_pair_lists(p, ptr, n)[0] != h[0]
    p = pairs.copy(); p[14, 1] += 1
    d = hash_pair_lists(p, ptr, n) != h
    assert d.tolist() == [False, False, True, False]
    p = pairs.copy(); p[:, [0, 1]] = p[:, [1, 0]]                        # ref_pos and read_pos exchanged
    assert (hash_pair_lists(p, ptr, n)[[0, 2]] != h[[0, 2]]).all()
    n2 = n.copy(); n2[2] = 6
    assert hash_pair_lists(pairs, ptr, n2)[2] != h[2]
    assert (hash_pair_lists(pairs.view(np.dtype([("a", "<i4"), ("b", "<i4")])).reshape(-1), ptr, n) == h).all()
