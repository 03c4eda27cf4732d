"""Row N4, oracle only (no device code yet): the restated profile-HMM forward score (oracle/abea_oracle.c,
src/hmm.c:314-735) against an independently written float32 twin in Python, plus properties of the score.  The pin against
the reference's own printed scores (single_read/meth_input.exp + meth.exp) is tests/test_hmm_pin.py."""
import numpy as np
import pytest

from f5c_amd.types import EVENT_DT, MODEL_DT

F = np.float32
NINF = F(-np.inf)


def _cpg_model(k, seed):
    r = np.random.default_rng(seed)
    m = np.zeros(5 ** k, dtype=MODEL_DT)
    m["level_mean"] = np.clip(r.normal(90, 12, 5 ** k), 50, 140).astype(np.float32)
    m["level_stdv"] = r.uniform(1.2, 3.5, 5 ** k).astype(np.float32)
    from f5c_amd.model import log_stdv
    m["level_log_stdv"] = log_stdv(m["level_stdv"])                                         # model.c:179 caches logf(stdv)
    return m


def _rank(kmer, k):
    r = 0
    for c in kmer[:k]:
        r = r * 5 + b"ACGMT".index(c)
    return r


def _methylate(seq):                       # meth.c:338-361: every CG -> MG
    return seq.replace(b"CG", b"MG")


def _rc_meth(seq):
    """meth.c:366-400 for complete sites: reverse complement in which a methylated CpG ("MG") stays "MG" — the site is
    its own reverse complement and the mark moves to the other strand's C."""
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    out = bytearray()
    s, i = bytes(seq), 0
    while i < len(s):
        if s[i:i + 2] == b"MG":
            out = bytearray(b"MG") + out
            i += 2
        else:
            out = bytearray([comp[s[i]]]) + out
            i += 1
    return bytes(out)


def _flogsum(tbl, a, b):
    mx, mn = (a, b) if a > b else (b, a)
    if mn == NINF or F(mx - mn) >= F(15.7):
        return mx
    return F(mx + tbl[int(F(F(mx - mn) * F(1000.0)))])


def _twin(tbl, m_seq, m_rc_seq, ev, scaling, model, k, e_start, e_stop, stride, rc, epb, flags):
    """Forward algorithm of hmm.c written from the HMM's definition (three states per k-mer: skip K, bad event B, match M)."""
    scale, shift, var, log_var = (F(x) for x in scaling)
    n_k = len(m_seq) - k + 1
    n_ev = abs(int(e_stop) - int(e_start)) + 1
    p_stay = F(1 - (1 / epb))
    p_skip, p_bad, p_skip_self = F(0.0025), F(0.001), F(0.3)
    from f5c_amd.model import _libm
    lg = lambda p: F(_libm.logf(float(p)))               # hmm.c is compiled as C++: log(float) is glibc logf
    lp_mk, lp_mb, lp_mm_self = lg(p_skip), lg(p_bad), lg(p_stay)
    lp_mm_next = lg(F(F(F(F(1.0) - p_stay) - p_skip) - p_bad))
    lp_bb = lg(p_bad)
    third = F(F(F(1.0) - p_bad) / F(3))
    lp_bk = lp_bm_next = lp_bm_self = lg(third)
    lp_kk, lp_km = lg(p_skip_self), lg(F(F(1.0) - p_skip_self))
    L = len(m_seq)
    ranks = [_rank(m_seq[i:i + k] if not rc else m_rc_seq[L - i - k:L - i], k) for i in range(n_k)]
    pre = [F(np.log(0.5)), F(np.log(0.5) + np.float64(F(-3.0)) + np.log(1 - 0.9))]
    for i in range(2, n_ev + 1):
        pre.append(F(np.log(0.9) + np.float64(F(-3.0)) + np.float64(pre[i - 1])))
    post = [F(0)] * n_ev
    post[n_ev - 1] = F(np.log(0.5))
    if n_ev > 1:
        post[n_ev - 2] = F(np.log(0.5) + np.float64(F(-3.0)) + np.log(1 - 0.9))
        for i in range(n_ev - 3, -1, -1):
            post[i] = F(np.log(0.9) + np.float64(F(-3.0)) + np.float64(post[i + 1]))
    prev = [[NINF] * 3 for _ in range(n_k + 1)]          # index 0 = start block; state order K, B, M
    end = NINF

    def lsum(xs):
        s = xs[0]
        for x in xs[1:]:
            s = _flogsum(tbl, s, x)
        return s
    for row in range(1, n_ev + 1):
        cur = [[NINF] * 3 for _ in range(n_k + 1)]
        e = int(e_start) + (row - 1) * stride
        x = F(ev["mean"][e])
        for b in range(1, n_k + 1):
            mo = model[ranks[b - 1]]
            gp_mean = F(F(scale * F(mo["level_mean"])) + shift)
            a = F(F(x - gp_mean) / F(F(mo["level_stdv"]) * var))
            lp = F(F(F(-0.918938) - F(F(mo["level_log_stdv"]) + log_var)) + F(F(F(-0.5) * a) * a))
            soft = F(F(0.0) + pre[row - 1]) if (b == 1 and (e == e_start or (flags & 1))) else NINF
            M = F(lsum([F(lp_mm_self + prev[b][2]), F(lp_mm_next + prev[b - 1][2]), F(lp_bm_self + prev[b][1]),
                        F(lp_bm_next + prev[b - 1][1]), F(lp_km + prev[b - 1][0]), soft]) + lp)
            B = F(lsum([F(lp_mb + prev[b][2]), NINF, F(lp_bb + prev[b][1]), NINF, NINF, NINF]) + F(0.0))
            cur[b][2], cur[b][1] = M, B
            Kk = F(lsum([NINF, F(lp_mk + cur[b - 1][2]), NINF, F(lp_bk + cur[b - 1][1]), F(lp_kk + cur[b - 1][0]), NINF]) + F(0.0))
            cur[b][0] = Kk
            if b == n_k and ((flags & 2) or row == n_ev):
                for st in (2, 1, 0):
                    end = _flogsum(tbl, end, F(F(F(0.0) + cur[b][st]) + post[row - 1]))
        prev = cur
    return float(end)


def _job(r, model, k, n_k, rc):
    seq = bytes(r.choice(list(b"ACGT"), n_k + k - 1).astype(np.uint8))
    seq = seq[:8] + b"CG" + seq[10:]                      # at least one CpG
    n_ev = int(r.integers(n_k, 3 * n_k))
    ev = np.zeros(n_ev + 40, dtype=EVENT_DT)
    ev["mean"] = r.normal(90, 12, len(ev)).astype(np.float32)
    return seq, ev, n_ev


@pytest.mark.parametrize("rc", [False, True])
def test_oracle_equals_independent_twin(orc, rc):
    k = 6
    model = _cpg_model(k, 7)
    tbl = orc.flogsum_table()
    assert tbl[0] == np.float32(np.log(2.0)) and abs(float(tbl[15999])) < 2e-7
    r = np.random.default_rng(3)
    for trial in range(6):
        seq, ev, n_ev = _job(r, model, k, int(r.integers(11, 24)), rc)
        for mseq in (seq, _methylate(seq)):
            rcs = _rc_meth(mseq)
            stride = -1 if rc else 1
            e_start = 20 + (n_ev - 1 if rc else 0)
            e_stop = 20 + (0 if rc else n_ev - 1)
            scal = (1.02, 1.5, 1.3, float(np.float32(np.log(np.float32(1.3)))))
            for flags in (3, 0):
                got = orc.profile_hmm_score(mseq, rcs, ev, scal, model, k, e_start, e_stop, stride, rc, 1.9, flags)
                want = _twin(tbl, mseq, rcs, ev, scal, model, k, e_start, e_stop, stride, rc, 1.9, flags)
                assert got == want, (trial, rc, flags, got, want)
                assert np.isfinite(got) and got < 0


def test_score_properties(orc):
    """Events generated from the unmethylated levels score higher under the unmethylated sequence; events outside
    e_start..e_stop do not matter; the rank function is base-5 over ACGMT."""
    k = 6
    model = _cpg_model(k, 11)
    assert orc.cpg_kmer_rank(b"AAAAAA", k) == 0 and orc.cpg_kmer_rank(b"TTTTTT", k) == 5 ** 6 - 1
    assert orc.cpg_kmer_rank(b"AAAAMG", k) == 3 * 5 + 2
    r = np.random.default_rng(5)
    wins = 0
    for trial in range(12):
        seq = bytes(r.choice(list(b"ACGT"), 22).astype(np.uint8))
        seq = seq[:10] + b"CG" + seq[12:]
        n_k = len(seq) - k + 1
        ev = np.zeros(2 * n_k + 10, dtype=EVENT_DT)
        lv = np.repeat([model["level_mean"][_rank(seq[i:i + k], k)] for i in range(n_k)], 2)
        ev["mean"][5:5 + 2 * n_k] = lv + r.normal(0, 1.0, 2 * n_k).astype(np.float32)
        scal = (1.0, 0.0, 1.0, 0.0)
        args = (ev, scal, model, k, 5, 5 + 2 * n_k - 1, 1, False, 2.0, 3)
        un = orc.profile_hmm_score(seq, b"", *args)
        me = orc.profile_hmm_score(_methylate(seq), b"", *args)
        wins += un > me
        ev2 = ev.copy(); ev2["mean"][:5] = 1e6; ev2["mean"][5 + 2 * n_k:] = -1e6
        assert orc.profile_hmm_score(seq, b"", ev2, *args[1:]) == un
    assert wins >= 10
