"""Row N4 on the device: batches of profile_hmm_score() calls (src/hmm.c:689-735, call site meth.c:473) through
abea_hmm_score_batch_host against the CPU restatement, bit for bit.  The oracle is pinned to the reference's
printed single_read/meth.exp scores (tests/test_hmm_pin.py, which also runs those 90 jobs on the GPU) and checked against
an independently written float32 twin in tests/test_hmm_oracle.py."""
import numpy as np
import pytest

from test_hmm_oracle import _cpg_model, _methylate, _rc_meth

pytestmark = pytest.mark.gpu


def _jobs(r, model, k, n, nk_lo, nk_hi):
    from f5c_amd.types import EVENT_DT
    jobs = []
    for _ in range(n):
        n_k = int(r.integers(nk_lo, nk_hi + 1))
        seq = bytes(r.choice(list(b"ACGT"), n_k + k - 1).astype(np.uint8))
        pos = int(r.integers(0, max(1, len(seq) - 2)))
        seq = seq[:pos] + b"CG" + seq[pos + 2:]                # at least one CpG
        n_ev = int(r.integers(max(1, n_k // 2), 3 * n_k + 2))
        pad = int(r.integers(0, 30))
        ev = np.zeros(n_ev + pad + 7, dtype=EVENT_DT)
        ev["mean"] = r.normal(90, 12, len(ev)).astype(np.float32)
        rc = bool(r.integers(0, 2))
        var = float(np.float32(r.uniform(0.8, 1.6)))
        scal = (float(np.float32(r.normal(1.0, 0.05))), float(np.float32(r.normal(0, 4))), var,
                float(np.float32(np.log(np.float32(var)))))
        for mseq in (seq, _methylate(seq)):                     # meth.c:473-474: both variants of every group
            jobs.append(dict(m_seq=mseq, m_rc_seq=_rc_meth(mseq), events=ev, scaling=scal,
                             e_start=pad + (n_ev - 1 if rc else 0), e_stop=pad + (0 if rc else n_ev - 1),
                             stride=-1 if rc else 1, rc=rc, events_per_base=float(r.uniform(1.3, 3.5)),
                             flags=int(r.integers(0, 4))))
    return jobs


def _oracle(orc, jobs, model, k):
    return np.array([orc.profile_hmm_score(j["m_seq"], j["m_rc_seq"], j["events"], j["scaling"], model, k, j["e_start"],
                                           j["e_stop"], j["stride"], j["rc"], j["events_per_base"], j["flags"])
                     for j in jobs], dtype=np.float32)


@pytest.mark.parametrize("k", [6, 5])
def test_profile_hmm_scores_bit_exact(ctx, orc, k):
    model = _cpg_model(k, 7 + k)
    r = np.random.default_rng(40 + k)
    jobs = (_jobs(r, model, k, 150, 1, 16)          # four to a wavefront
            + _jobs(r, model, k, 60, 17, 64)        # one per wavefront
            + _jobs(r, model, k, 12, 65, 200))      # tiled: more k-mers than lanes
    order = r.permutation(len(jobs))
    jobs = [jobs[i] for i in order]
    got = ctx.hmm_score_batch(jobs, model, k)
    want = _oracle(orc, jobs, model, k)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    assert len(bad) == 0, [(int(i), len(jobs[i]["m_seq"]) - k + 1, jobs[i]["rc"], jobs[i]["flags"], got[i], want[i]) for i in bad[:8]]
    assert np.isfinite(want).mean() > 0.9
    assert ctx.stats()["hmm_ms"] > 0
    # a second call, different job count, same context
    got2 = ctx.hmm_score_batch(jobs[:37], model, k)
    assert (got2.view(np.uint32) == want[:37].view(np.uint32)).all()


def test_profile_hmm_scores_from_device_resident_event_tables(ctx, orc):
    """abea_hmm_score_batch_device: the reads' event tables are in HBM (as the chain leaves them) and every job names its
    read's table by device address; the windows are gathered on the device.  Same bits as the host entry and the oracle."""
    import torch
    from f5c_amd.types import EVENT_DT
    k = 6
    model = _cpg_model(k, 21)
    r = np.random.default_rng(77)
    jobs = _jobs(r, model, k, 120, 1, 16) + _jobs(r, model, k, 40, 17, 64) + _jobs(r, model, k, 8, 65, 150)
    want = _oracle(orc, jobs, model, k)
    # one flat device buffer holding every job's table back to back, like `events` + event_ptr[] of a device batch
    off, flat = [], []
    at = 0
    for j in jobs:
        off.append(at); flat.append(np.ascontiguousarray(j["events"], dtype=EVENT_DT)); at += len(j["events"])
    d_ev = torch.from_numpy(np.concatenate(flat).view(np.uint8)).cuda()
    torch.cuda.synchronize()
    djobs = [dict(j, events=d_ev.data_ptr() + 24 * o) for j, o in zip(jobs, off)]
    got = ctx.hmm_score_batch(djobs, model, k, device_events=True)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert (ctx.hmm_score_batch(jobs, model, k).view(np.uint32) == want.view(np.uint32)).all()
    assert ctx.stats()["hmm_ms"] > 0


def test_methylation_signal_and_call_shape(ctx, orc):
    """The use meth.c makes of the scores: events drawn from the unmethylated levels favour the unmethylated sequence
    (log-likelihood ratio < 0), from the methylated levels the methylated one; GPU == CPU on every score."""
    from f5c_amd.types import EVENT_DT
    k = 6
    model = _cpg_model(k, 11)
    r = np.random.default_rng(9)

    def rank(kmer):
        v = 0
        for c in kmer:
            v = v * 5 + b"ACGMT".index(c)
        return v
    jobs, truth = [], []
    for trial in range(40):
        seq = bytes(r.choice(list(b"ACGT"), 22).astype(np.uint8))
        seq = seq[:10] + b"CG" + seq[12:]
        meth = bool(trial & 1)
        src = _methylate(seq) if meth else seq
        n_k = len(seq) - k + 1
        ev = np.zeros(2 * n_k + 10, dtype=EVENT_DT)
        lv = np.repeat([model["level_mean"][rank(src[i:i + k])] for i in range(n_k)], 2)
        ev["mean"][5:5 + 2 * n_k] = lv + r.normal(0, 1.0, 2 * n_k).astype(np.float32)
        for mseq in (seq, _methylate(seq)):
            jobs.append(dict(m_seq=mseq, m_rc_seq=_rc_meth(mseq), events=ev, scaling=(1.0, 0.0, 1.0, 0.0), e_start=5,
                             e_stop=5 + 2 * n_k - 1, stride=1, rc=False, events_per_base=2.0, flags=3))
        truth.append(meth)
    got = ctx.hmm_score_batch(jobs, model, k)
    want = _oracle(orc, jobs, model, k)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    llr = got[1::2] - got[0::2]                                  # methylated - unmethylated (meth.c:482)
    assert ((llr > 0) == np.array(truth)).mean() > 0.9
