"""Deterministic synthetic read/event batches in the flattened layout of the reference's device
arrays (src/f5c.cu:672-690): the workloads BASELINE.json names.

  reads      uint8  Σ(L+1)   sequences, each NUL-terminated
  read_ptr   int64  [n]      offset of read i in `reads`
  read_len   int32  [n]
  events     event_t Σ E     AoS {start u64, length f32, mean f32, stdv f32}
  event_ptr  int64  [n]
  n_events   int32  [n]
  pair_ptr   int64  [n]      offset of read i's output pairs; capacity n_events+read_len (f5c.c:724)
  scalings   scalings_t [n]  method-of-moments estimate (align.c:58-106 restated with numpy)

Read synthesis: bases i.i.d. ACGT; each k-mer emits d events, d = 0 with p=0.04 else 1+Poisson(1.08)
(≈2 events/base, a few skips); event.mean = scale_t*model.mean + shift_t + N(0,(1.3*model.stdv)^2);
≈1 % of reads get pure-noise signal (exercise the QC-fail path).  Every read draws from its own generator seeded
[seed, read index], so any box regenerates the identical batch and any subset of it can be built on its own.
"""
import numpy as np
from .types import EVENT_DT, SCAL_DT

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)

CONFIGS = {
    # name: (n_reads, seed, length law, k)   BASELINE.json configs[1..4]
    "r9_10k_8kb": dict(n_reads=10_000, seed=20250002, law="gamma8k", k=6),
    "r9_100k_mixed": dict(n_reads=100_000, seed=20250003, law="loguniform", k=6),
    "r10_50k_10kb": dict(n_reads=50_000, seed=20250005, law="gamma10k", k=9),
    # not a BASELINE config: fixed-length reads, isolates kernel throughput from the longest-read tail
    "r9_uniform_8kb": dict(n_reads=10_000, seed=20250099, law=8000, k=6),
}


def read_lengths(law, n, rng):
    if law == "gamma8k":
        L = rng.gamma(4.0, 8000.0 / 4.0, n)
        return np.clip(L, 1000, 30000).astype(np.int64)
    if law == "gamma10k":
        L = rng.gamma(4.0, 10000.0 / 4.0, n)
        return np.clip(L, 1000, 40000).astype(np.int64)
    if law == "loguniform":
        return np.exp(rng.uniform(np.log(1000.0), np.log(50000.0), n)).astype(np.int64)
    if isinstance(law, (int, np.integer)):
        return np.full(n, int(law), dtype=np.int64)
    raise ValueError(law)


def _kmer_ranks(codes, k):
    """rank of the k-mer starting at every position of `codes` (valid for positions <= len-k)."""
    n = len(codes) - k + 1
    r = np.zeros(n, dtype=np.int64)
    for j in range(k):
        r = (r << 2) | codes[j:j + n]
    return r


def _chunk(idx, lengths, model, k, seed, bad_frac):
    """Reads `idx` (global read indices) of the batch; every read draws from its own generator seeded [seed, index],
    so any subset of a batch can be generated on its own (bench.py: each rank builds only its shard)."""
    n = len(lengths)
    K = lengths - k + 1
    codes_l, d_l, noise_l, len_l, sd_l, bad_l, badmean_l = [], [], [], [], [], [], []
    scale_t = np.zeros(n); shift_t = np.zeros(n)
    for j in range(n):
        rng = np.random.default_rng([seed, int(idx[j]) + 1])
        L, Kj = int(lengths[j]), int(K[j])
        codes_l.append(rng.integers(0, 4, L, dtype=np.int64))
        d = np.where(rng.random(Kj, dtype=np.float32) < 0.04, 0, 1 + rng.poisson(1.08, Kj)).astype(np.int64)
        d[0] = max(d[0], 1)                                  # every read needs at least one event
        d_l.append(d)
        E = int(d.sum())
        scale_t[j] = rng.normal(1.0, 0.04); shift_t[j] = rng.normal(0.0, 6.0)
        noise_l.append(rng.standard_normal(E, dtype=np.float32))
        len_l.append((1 + rng.poisson(8.0, E)).astype(np.float32))
        sd_l.append(0.5 + 2.5 * rng.random(E, dtype=np.float32))
        bad = rng.random() < bad_frac
        bad_l.append(bad)
        badmean_l.append(90.0 + 12.0 * rng.standard_normal(E, dtype=np.float32) if bad else None)
    tot = int(lengths.sum())
    codes = np.concatenate(codes_l)
    starts = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    # k-mer start positions (exclude k-mers that would straddle two reads)
    kpos = np.concatenate([np.arange(s, s + kk) for s, kk in zip(starts, K)])
    kread = np.repeat(np.arange(n), K)
    ranks = _kmer_ranks(np.concatenate([codes, np.zeros(k, dtype=np.int64)]), k)[kpos]
    d = np.concatenate(d_l)
    E = np.bincount(kread, weights=d, minlength=n).astype(np.int64)
    ev_rank = np.repeat(ranks, d)
    ev_read = np.repeat(kread, d)
    mu = model["level_mean"][ev_rank]
    sd = model["level_stdv"][ev_rank]
    noise = np.concatenate(noise_l)
    mean = scale_t.astype(np.float32)[ev_read] * mu + shift_t.astype(np.float32)[ev_read] + noise * (1.3 * sd)
    bad = np.array(bad_l, dtype=bool)
    estart = np.concatenate([[0], np.cumsum(E)[:-1]])
    for j in np.nonzero(bad)[0]:
        mean[estart[j]:estart[j] + E[j]] = badmean_l[j]
    length = np.concatenate(len_l)
    ev = np.zeros(len(ev_rank), dtype=EVENT_DT)
    ev["mean"] = mean.astype(np.float32)
    ev["length"] = length
    ev["stdv"] = np.concatenate(sd_l)
    cs = np.cumsum(length.astype(np.int64))
    run = cs - length.astype(np.int64)
    ev["start"] = (run - np.repeat(run[estart], E)).astype(np.uint64)
    # method-of-moments scalings (align.c:58-106), float64 numpy reductions
    m32 = ev["mean"].astype(np.float64)
    ev_sum = np.bincount(ev_read, weights=m32, minlength=n)
    km = model["level_mean"].astype(np.float64)[ranks]
    km_sum = np.bincount(kread, weights=km, minlength=n)
    km_sq = np.bincount(kread, weights=km * km, minlength=n)
    shift = ev_sum / E - km_sum / K
    ev_sq = np.bincount(ev_read, weights=(m32 - shift[ev_read]) ** 2, minlength=n)
    scale = (ev_sq / E) / (km_sq / K)
    sc = np.zeros(n, dtype=SCAL_DT)
    sc["scale"] = scale.astype(np.float32)
    sc["shift"] = shift.astype(np.float32)
    sc["var"] = 1.0
    # sequences with NUL terminators
    seq = np.zeros(tot + n, dtype=np.uint8)
    dst = np.arange(tot) + np.repeat(np.arange(n), lengths)
    seq[dst] = _BASES[codes]
    return seq, ev, E.astype(np.int32), sc, bad


_G = {}


def _pool_init(args):
    _G["args"] = args


def _chunk_job(c):
    idx, L, model, k, seed, bad_frac, chunk_reads = _G["args"]
    sl = slice(c * chunk_reads, (c + 1) * chunk_reads)
    return _chunk(idx[sl], L[sl], model, k, seed, bad_frac)


_TOOL_ENV = ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIB")


def _under_profiler():
    import os
    return any(("rocprof" in os.environ.get(v, "").lower()) for v in _TOOL_ENV) or \
        any(v.startswith("ROCPROF") for v in os.environ)


def batch_lengths(n_reads, seed, law):
    """Read lengths of a whole batch (cheap: needs no generation); bench.py shards on them before generating."""
    return read_lengths(law, n_reads, np.random.default_rng([seed, 0x5EED]))


def make_batch(n_reads, model, k, seed, law="gamma8k", bad_frac=0.01, chunk_reads=256, lengths=None,
               workers=1, subset=None):
    """Build a flattened batch. `lengths` overrides the length law (array of read lengths).
    `workers` > 1 generates chunks in worker processes (same result: every read has its own generator): forked
    normally; under rocprofv3 the workers are SPAWNED with the profiler's preload variables removed (forking a
    process that carries the profiler's tool threads hung the 100k-read WRITE_SIZE pass of round 2).
    `subset`: indices of the reads to build (in that order) instead of the whole batch."""
    L = np.asarray(lengths, dtype=np.int64) if lengths is not None else batch_lengths(n_reads, seed, law)
    idx = np.arange(len(L), dtype=np.int64)
    if subset is not None:
        idx = np.asarray(subset, dtype=np.int64)
        L = L[idx]
    n_reads = len(L)
    n_chunks = (n_reads + chunk_reads - 1) // chunk_reads
    _G["args"] = (idx, L, model, k, seed, bad_frac, chunk_reads)
    if workers > 1 and n_chunks > 1:
        import multiprocessing as mp
        import os
        if _under_profiler() or os.environ.get("ABEA_SYNTH_SPAWN"):
            saved = {v: os.environ.pop(v) for v in list(os.environ) if v in _TOOL_ENV or v.startswith("ROCPROF")}
            try:
                with mp.get_context("spawn").Pool(min(workers, n_chunks), initializer=_pool_init,
                                                  initargs=(_G["args"],)) as pool:
                    parts = pool.map(_chunk_job, range(n_chunks), chunksize=1)
            finally:
                os.environ.update(saved)
        else:
            with mp.get_context("fork").Pool(min(workers, n_chunks)) as pool:
                parts = pool.map(_chunk_job, range(n_chunks), chunksize=1)
    else:
        parts = [_chunk_job(c) for c in range(n_chunks)]
    seqs = [p[0] for p in parts]; evs = [p[1] for p in parts]; Es = [p[2] for p in parts]
    scs = [p[3] for p in parts]; bads = [p[4] for p in parts]
    E = np.concatenate(Es) if Es else np.zeros(0, np.int32)
    batch = {
        "reads": np.concatenate(seqs) if seqs else np.zeros(0, np.uint8),
        "read_len": L.astype(np.int32),
        "read_ptr": np.concatenate([[0], np.cumsum(L + 1)[:-1]]).astype(np.int64) if n_reads else np.zeros(0, np.int64),
        "events": np.concatenate(evs) if evs else np.zeros(0, EVENT_DT),
        "n_events": E,
        "event_ptr": np.concatenate([[0], np.cumsum(E.astype(np.int64))[:-1]]).astype(np.int64) if n_reads else np.zeros(0, np.int64),
        "scalings": np.concatenate(scs) if scs else np.zeros(0, SCAL_DT),
        "bad": np.concatenate(bads) if bads else np.zeros(0, bool),
    }
    cap = E.astype(np.int64) + L
    batch["pair_ptr"] = (np.concatenate([[0], np.cumsum(cap)[:-1]]).astype(np.int64)
                         if n_reads else np.zeros(0, np.int64))
    batch["pair_cap"] = int(cap.sum())
    return batch


def lpt_bins(weight, world):
    """Longest-processing-time-first split, the rule of the library's abea_lpt_split: items in descending
    weight go to the currently lightest bin, ties to the lowest bin.  Returns the bin of every item."""
    import heapq
    weight = np.asarray(weight, dtype=np.int64)
    order = np.argsort(-weight, kind="stable")
    heap = [(0, r) for r in range(world)]
    bins = np.zeros(len(weight), dtype=np.int32)
    for i in order:
        load, r = heapq.heappop(heap)
        bins[i] = r
        heapq.heappush(heap, (load + max(int(weight[i]), 0), r))
    return bins


def batch_from_reads(seqs, events_list, scalings):
    """Flatten explicit reads: seqs = list of bytes, events_list = list of EVENT_DT arrays,
    scalings = list of (scale, shift)."""
    n = len(seqs)
    L = np.array([len(s) for s in seqs], dtype=np.int64)
    E = np.array([len(e) for e in events_list], dtype=np.int64)
    reads = np.zeros(int(L.sum()) + n, dtype=np.uint8)
    rp = np.concatenate([[0], np.cumsum(L + 1)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
    for i, s in enumerate(seqs):
        reads[rp[i]:rp[i] + L[i]] = np.frombuffer(s, dtype=np.uint8)
    sc = np.zeros(n, dtype=SCAL_DT)
    for i, (a, b) in enumerate(scalings):
        sc["scale"][i] = a
        sc["shift"][i] = b
        sc["var"][i] = 1.0
    cap = E + L
    return {
        "reads": reads, "read_len": L.astype(np.int32), "read_ptr": rp,
        "events": (np.concatenate(events_list) if n and E.sum() else np.zeros(0, EVENT_DT)),
        "n_events": E.astype(np.int32),
        "event_ptr": np.concatenate([[0], np.cumsum(E)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64),
        "scalings": sc,
        "pair_ptr": np.concatenate([[0], np.cumsum(cap)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64),
        "pair_cap": int(cap.sum()),
        "bad": np.zeros(n, bool),
    }


def shard_batch(batch, rank, world):
    """LPT shard of a batch over `world` GPUs: reads sorted by band count E+K
    descending are dealt greedily to the lightest bin; returns the sub-batch of `rank` plus the
    original read indices it holds."""
    w = batch["n_events"].astype(np.int64) + batch["read_len"].astype(np.int64)
    idx = np.nonzero(lpt_bins(w, world) == rank)[0].astype(np.int64)
    return take_reads(batch, idx), idx


def take_reads(batch, idx):
    """Sub-batch holding reads `idx` (in that order), re-flattened."""
    idx = np.asarray(idx, dtype=np.int64)
    L = batch["read_len"][idx].astype(np.int64)
    E = batch["n_events"][idx].astype(np.int64)
    n = len(idx)
    rp = np.concatenate([[0], np.cumsum(L + 1)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
    ep = np.concatenate([[0], np.cumsum(E)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
    reads = np.zeros(int((L + 1).sum()), dtype=np.uint8)
    events = np.zeros(int(E.sum()), dtype=EVENT_DT)
    for j, i in enumerate(idx):
        s = batch["read_ptr"][i]
        reads[rp[j]:rp[j] + L[j]] = batch["reads"][s:s + L[j]]
        s = batch["event_ptr"][i]
        events[ep[j]:ep[j] + E[j]] = batch["events"][s:s + E[j]]
    cap = E + L
    return {
        "reads": reads, "read_len": L.astype(np.int32), "read_ptr": rp, "events": events,
        "n_events": E.astype(np.int32), "event_ptr": ep, "scalings": batch["scalings"][idx].copy(),
        "pair_ptr": np.concatenate([[0], np.cumsum(cap)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64),
        "pair_cap": int(cap.sum()), "bad": batch["bad"][idx].copy(),
    }


def make_signals(batch, seed=0, offset=10.0, rng_pa=1467.61, digitisation=8192.0):
    """Raw int16 ADC signals for the reads of a batch: each event contributes `length` samples drawn around its
    mean (so event detection has something real to find). Returns (list of int16 arrays, float32 [n,3] scaling)."""
    r = np.random.default_rng([seed, 0x516])
    raw_unit = np.float32(rng_pa) / np.float32(digitisation)
    sigs = []
    for i in range(len(batch["read_len"])):
        s, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        ev = batch["events"][s:s + E]
        ln = ev["length"].astype(np.int64)
        pa = np.repeat(ev["mean"].astype(np.float64), ln) + r.normal(0.0, 1.2, int(ln.sum()))
        sigs.append(np.clip(np.rint(pa / raw_unit - offset), -32768, 32767).astype(np.int16))
    sc = np.tile(np.array([offset, rng_pa, digitisation], dtype=np.float32), (len(sigs), 1))
    return sigs, sc


def make_signals_flat(batch, idx=None, seed=0, offset=10.0, rng_pa=1467.61, digitisation=8192.0, threads=8):
    """The same signals as make_signals() in law (each event contributes `length` samples drawn around its mean, sigma 1.2 pA),
    for whole batches: one flat FLOAT32 array of ADC counts — the form f5c holds them in (signal_t.rawptr, f5c.h:276-286) —
    built by `threads` numpy threads, every read with its own generator.  Returns (signal float32, sig_ptr int64[n] in
    samples, n_samples int64[n], scaling float32 [n,3])."""
    import threading
    idx = np.arange(len(batch["read_len"])) if idx is None else np.asarray(idx)
    n = len(idx)
    raw_unit = np.float32(rng_pa) / np.float32(digitisation)
    ep = batch["event_ptr"].astype(np.int64)
    ne = batch["n_events"].astype(np.int64)
    ns = np.zeros(n, dtype=np.int64)
    lens = []
    for j, i in enumerate(idx):
        ln = batch["events"]["length"][ep[i]:ep[i] + ne[i]].astype(np.int64)
        lens.append(ln)
        ns[j] = int(ln.sum())
    pad = (ns + 7) // 8 * 8
    sig_ptr = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
    out = np.zeros(int(pad.sum()), dtype=np.float32)

    def work(t):
        for j in range(t, n, threads):
            i = int(idx[j])
            r = np.random.default_rng([seed, 0x516, i])
            mean = batch["events"]["mean"][ep[i]:ep[i] + ne[i]]
            pa = np.repeat(mean, lens[j]) + np.float32(1.2) * r.standard_normal(int(ns[j]), dtype=np.float32)
            np.clip(np.rint(pa / raw_unit - np.float32(offset)), -32768, 32767, out=out[sig_ptr[j]:sig_ptr[j] + ns[j]])
    th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    sc = np.tile(np.array([offset, rng_pa, digitisation], dtype=np.float32), (n, 1))
    return out, sig_ptr, ns, sc
