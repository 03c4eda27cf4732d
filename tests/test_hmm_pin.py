"""Row N4 PINNED: the reference prints, for read1 of test/ecoli_2kb_region/single_read, the arguments of its 90
profile_hmm_score() calls (meth_input.exp: m_seq, m_rc_seq, event_start/stop, stride, rc — call site meth.c:473-474)
and the resulting log_lik_methylated / log_lik_unmethylated of the same 45 CpG groups (single_read/meth.exp).  The
oracle chain (method-of-moments scalings -> align -> scaling_single, each pinned elsewhere) followed by the restated
profile_hmm_score (src/hmm.c:689-735; hmm_flags = HAF_ALLOW_PRE_CLIP|HAF_ALLOW_POST_CLIP, meth.c:559) must reproduce
the printed numbers.  Fixtures are data only (tests/golden/make_golden.py): single_read_meth.npz and the reference's
CpG 6-mer table r9.4_450bps.cpg.6mer.f32 (alphabet ACGMT).

Tolerance: meth.exp prints %.2f; 89 of the 90 scores equal the printed value at two decimals, one sits 0.005 from a
rounding boundary (-312.1950 vs -312.20), so the test allows 0.006 and counts the exact-at-%.2f matches."""
import os
import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cpg_model():
    from f5c_amd.model import _finish
    tab = np.fromfile(os.path.join(GOLDEN, "r9.4_450bps.cpg.6mer.f32"), dtype=np.float32).reshape(-1, 2)
    assert len(tab) == 5 ** 6
    return 6, _finish(tab[:, 0], tab[:, 1])


def single_read_meth_jobs(orc, r9, single_read):
    """The 90 jobs with read1's recalibrated scalings, as calculate_methylation_for_read passes them (meth.c:430-474)."""
    k, model = r9
    seq, ev = single_read["seq"], single_read["events"]
    scale, shift = orc.estimate_scalings(seq, model, k, ev)
    pairs, _ = orc.align(seq, ev, model, k, scale, shift)
    rec = orc.scaling_single(pairs, seq, ev, model, k, scale, shift)
    assert rec["flag"] == 0
    sc = rec["scalings"]
    scaling = (float(sc["scale"]), float(sc["shift"]), float(sc["var"]), float(sc["log_var"]))
    g = np.load(os.path.join(GOLDEN, "single_read_meth.npz"))
    jobs = []
    for s, rcs, a in zip(g["m_seq"], g["m_rc_seq"], g["args"]):
        jobs.append(dict(m_seq=str(s).encode(), m_rc_seq=str(rcs).encode(), events=ev, scaling=scaling,
                         e_start=int(a[0]), e_stop=int(a[1]), stride=int(a[2]), rc=bool(a[3]),
                         events_per_base=rec["events_per_base"], flags=3))
    want = np.empty(len(jobs))
    want[0::2] = g["exp_log_lik_unmethylated"]          # job 2g: unmethylated sequence, 2g+1: methylated (meth.c:473-474)
    want[1::2] = g["exp_log_lik_methylated"]
    return jobs, want, g, rec


def check_against_printed(scores, want, g):
    scores = np.asarray(scores, dtype=np.float64)
    assert np.isfinite(scores).all()
    err = np.abs(scores - want)
    assert err.max() < 0.006, (int(err.argmax()), scores[err.argmax()], want[err.argmax()])
    same_print = sum(("%.2f" % s) == ("%.2f" % w) for s, w in zip(scores, want))
    assert same_print >= 89, same_print
    # the printed log_lik_ratio = methylated - unmethylated (meth.c:482), %.2f of float differences
    llr = scores[1::2].astype(np.float32) - scores[0::2].astype(np.float32)
    assert np.abs(llr - g["exp_log_lik_ratio"]).max() < 0.011


def test_recalibrated_scalings_of_read1(orc, r9, single_read):
    _, _, _, rec = single_read_meth_jobs(orc, r9, single_read)
    sc = rec["scalings"]
    assert ("%.2f %.2f %.2f" % (sc["shift"], sc["scale"], sc["var"])) == "3.18 0.98 1.40"
    assert abs(rec["events_per_base"] - 1.9595) < 1e-3


def test_oracle_profile_hmm_reproduces_meth_exp(orc, r9, single_read):
    kc, cpg = load_cpg_model()
    jobs, want, g, _ = single_read_meth_jobs(orc, r9, single_read)
    assert len(jobs) == 90
    scores = [orc.profile_hmm_score(j["m_seq"], j["m_rc_seq"], j["events"], j["scaling"], cpg, kc, j["e_start"],
                                    j["e_stop"], j["stride"], j["rc"], j["events_per_base"], j["flags"]) for j in jobs]
    check_against_printed(scores, want, g)


@pytest.mark.gpu
def test_gpu_profile_hmm_reproduces_meth_exp(ctx, orc, r9, single_read):
    """The same 90 jobs through abea_hmm_score_batch_host: bit-equal to the oracle AND within 0.006 of meth.exp."""
    kc, cpg = load_cpg_model()
    jobs, want, g, _ = single_read_meth_jobs(orc, r9, single_read)
    got = ctx.hmm_score_batch(jobs, cpg, kc)
    ref = np.array([orc.profile_hmm_score(j["m_seq"], j["m_rc_seq"], j["events"], j["scaling"], cpg, kc, j["e_start"],
                                          j["e_stop"], j["stride"], j["rc"], j["events_per_base"], j["flags"])
                    for j in jobs], dtype=np.float32)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    check_against_printed(got, want, g)
