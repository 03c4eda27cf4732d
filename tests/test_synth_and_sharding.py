"""CPU: synthetic batch layout invariants, determinism, LPT sharding, and the world_size-2 gloo path
of the multi-GPU driver logic (reads shard with no data-path collective; only a final gather)."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_layout_and_determinism(r9):
    from f5c_amd import synth
    k, model = r9
    a = synth.make_batch(40, model, k, seed=5, law="loguniform", chunk_reads=16)
    b = synth.make_batch(40, model, k, seed=5, law="loguniform", chunk_reads=16, workers=2)
    for key in ("reads", "events", "n_events", "read_len", "scalings", "read_ptr", "event_ptr", "pair_ptr"):
        assert (a[key] == b[key]).all(), key
    n = 40
    assert a["read_ptr"][0] == 0 and a["event_ptr"][0] == 0 and a["pair_ptr"][0] == 0
    assert (np.diff(a["read_ptr"]) == a["read_len"][:-1] + 1).all()
    assert (np.diff(a["event_ptr"]) == a["n_events"][:-1]).all()
    assert (np.diff(a["pair_ptr"]) == (a["n_events"][:-1].astype(np.int64) + a["read_len"][:-1])).all()
    for i in range(n):
        s, L = int(a["read_ptr"][i]), int(a["read_len"][i])
        assert a["reads"][s + L] == 0 and set(a["reads"][s:s + L].tobytes()) <= set(b"ACGT")
    epb = a["n_events"].sum() / a["read_len"].sum()
    assert 1.8 < epb < 2.2
    assert a["read_len"].min() >= 1000 and a["read_len"].max() <= 50000


def test_lpt_sharding_partitions_and_balances(r9):
    from f5c_amd import synth
    k, model = r9
    full = synth.make_batch(64, model, k, seed=6, law="loguniform")
    seen = []
    loads = []
    for r in range(4):
        sub, idx = synth.shard_batch(full, r, 4)
        seen.extend(idx.tolist())
        loads.append(int((sub["n_events"].astype(np.int64) + sub["read_len"]).sum()))
        for j, i in enumerate(idx):
            s, L = int(full["read_ptr"][i]), int(full["read_len"][i])
            t = int(sub["read_ptr"][j])
            assert (sub["reads"][t:t + L] == full["reads"][s:s + L]).all()
    assert sorted(seen) == list(range(64))
    assert max(loads) / (sum(loads) / 4) < 1.15        # LPT keeps the bins within 15 % of the mean


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from f5c_amd import synth, load_model_f32, dist_util
    from oracle import orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, model = load_model_f32(os.path.join(ROOT, "tests", "golden", "r9.4_450bps.6mer.f32"))
    # as bench.py does for config 4: split on the read lengths BEFORE generating, then build only this rank's reads
    L_all = synth.batch_lengths(24, 8, "loguniform")
    idx = np.nonzero(synth.lpt_bins(3 * L_all, world) == rank)[0]
    sub = synth.make_batch(24, model, k, seed=8, law="loguniform", bad_frac=0.1, subset=idx)
    # the CPU checker stands in for the per-rank GPU call here (tests/test_host_pipeline.py runs the real bench with
    # 2 ranks on the GPU box); the sharding / gather logic is what is tested
    _, n_pairs, _ = orc.align_batch(sub, model, k, n_threads=2, want_diag=False)
    stats = dist_util.gather_stats(dict(elapsed=0.01 * (rank + 1), events=float(sub["n_events"].sum()),
                                        reads=float(len(idx)), pairs=float(n_pairs.sum())), device="cpu")
    allp = dist_util.gather_per_read(idx, n_pairs, len(L_all), device="cpu")
    if rank == 0:
        np.save(out, np.concatenate([[stats["t_max"], stats["events"], stats["reads"], stats["pairs"]], allp]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather(r9, orc, tmp_path):
    import torch.multiprocessing as mp
    from f5c_amd import synth
    k, model = r9
    out = str(tmp_path / "res.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    full = synth.make_batch(24, model, k, seed=8, law="loguniform", bad_frac=0.1)     # shards built alone == slices of the whole
    _, n_pairs, _ = orc.align_batch(full, model, k, n_threads=4, want_diag=False)
    assert res[0] == pytest.approx(0.02)                   # MAX over ranks
    assert res[1] == full["n_events"].sum() and res[2] == 24 and res[3] == n_pairs.sum()
    assert (res[4:].astype(np.int64) == n_pairs).all()     # per-read gather reassembles the unsharded result


def test_lpt_split_of_the_100k_batch_over_8_gpus_is_balanced():
    """BASELINE configs[3]: the 100k-read batch over 8 GPUs.  The library's split (abea_lpt_split, what
    abea_align_batch_host applies on a multi-device context and bench.py per rank) on the real band counts E + K + 2 of
    configs[2] (n_events / read_len of every read are part of the committed goldens): every bin within 2 % of the mean."""
    import ctypes as C
    from f5c_amd import abea, synth
    path = os.path.join(ROOT, "tests", "golden", "config_goldens_r9_100k_mixed.npz")
    if not os.path.exists(path):
        pytest.skip("config goldens not minted")
    g = np.load(path)
    w = g["n_events"].astype(np.int64) + (g["read_len"].astype(np.int64) - 6 + 1) + 2
    lib = abea.load_library()
    for bins in (2, 4, 8):
        out = np.zeros(len(w), dtype=np.int32)
        assert lib.abea_lpt_split(w.ctypes.data_as(C.c_void_p), len(w), bins, out.ctypes.data_as(C.c_void_p)) == 0
        assert (out == synth.lpt_bins(w, bins)).all()                 # bench.py's per-rank rule is the same rule
        load = np.bincount(out, weights=w, minlength=bins)
        assert load.min() > 0 and (np.abs(load / load.mean() - 1.0) < 0.02).all(), load / load.mean()
        # and on what bench.py actually shards on before the batch exists: 3 x read length (E ~ 2.04 L for this generator)
        pre = synth.lpt_bins(3 * g["read_len"].astype(np.int64), bins)
        load = np.bincount(pre, weights=w, minlength=bins)
        assert (np.abs(load / load.mean() - 1.0) < 0.02).all(), load / load.mean()
