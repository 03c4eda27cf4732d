#!/usr/bin/env python3
"""Mint per-read goldens for EVERY read of the synthetic BASELINE configs (configs[1], [2], [4]) from the CPU oracle:

    tests/golden/config_goldens_<config>.npz :  n_pairs int32 [n], pair_hash uint64 [n] (tests/pairhash.py),
                                                sum_emission float64 [n], n_aligned int32 [n], best_event int32 [n],
                                                n_events int32 [n], read_len int32 [n]   (the last two pin the generator)

Run in the build container (≈25 min on 8 cores for all three):   python tests/golden/make_config_goldens.py [config ...]
The oracle (oracle/abea_oracle.c) is the pinned line-by-line restatement of src/align.c:180-559; the batch is
f5c_amd.synth's seeded generator (every read has its own stream, so the batch is built and aligned in slices of a few
thousand reads and no 60 GB table is ever held).  tests/test_full_size.py and tests/test_gpu_parity.py compare the GPU's
output of every read with these on the GPU box, where neither the oracle's full run nor /root/reference is affordable.
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from f5c_amd import synth, load_model_f32, synthetic_model   # noqa: E402
from oracle import orc                                        # noqa: E402
from pairhash import hash_pair_lists                          # noqa: E402

SLICE = 4000


def config_model(k):
    return (load_model_f32(os.path.join(ROOT, "tests", "golden", "r9.4_450bps.6mer.f32"))[1] if k == 6
            else synthetic_model(k, seed=9))


def mint(config, workers):
    cfg = synth.CONFIGS[config]
    k, n = cfg["k"], cfg["n_reads"]
    model = config_model(k)
    out = dict(n_pairs=np.zeros(n, np.int32), pair_hash=np.zeros(n, np.uint64), sum_emission=np.zeros(n, np.float64),
               n_aligned=np.zeros(n, np.int32), best_event=np.zeros(n, np.int32), n_events=np.zeros(n, np.int32),
               read_len=np.zeros(n, np.int32))
    t0 = time.time()
    for a in range(0, n, SLICE):
        idx = np.arange(a, min(n, a + SLICE))
        b = synth.make_batch(n, model, k, seed=cfg["seed"], law=cfg["law"], workers=workers, subset=idx)
        pairs, n_pairs, diag = orc.align_batch(b, model, k, n_threads=workers)
        out["n_pairs"][idx] = n_pairs
        out["pair_hash"][idx] = hash_pair_lists(pairs, b["pair_ptr"], n_pairs)
        out["sum_emission"][idx] = diag["sum_emission"]
        out["n_aligned"][idx] = diag["n_aligned"]
        out["best_event"][idx] = diag["best_event"]
        out["n_events"][idx] = b["n_events"]
        out["read_len"][idx] = b["read_len"]
        print(f"{config}: {idx[-1] + 1}/{n} reads, {time.time() - t0:.0f} s", flush=True)
    path = os.path.join(ROOT, "tests", "golden", f"config_goldens_{config}.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", int((out["n_pairs"] > 0).sum()), "reads with pairs,",
          int(out["n_pairs"].sum()), "pairs")


if __name__ == "__main__":
    w = max(1, len(os.sched_getaffinity(0)))
    for c in (sys.argv[1:] or ["r9_10k_8kb", "r10_50k_10kb", "r9_100k_mixed"]):
        mint(c, w)
