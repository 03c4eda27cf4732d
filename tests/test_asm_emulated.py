"""The hand-written gfx950 statements of the alignment kernel (abea_fill.inc: band fill, interior and border variant;
abea_walk.inc: scalar traceback walk), EXECUTED on the CPU by tools/gfx950_emu.py and compared with the oracle.

This is the only place where the hot loop's arithmetic and control flow are checked without a GPU: the C++ around the
statements (abea_kernels.hip: the state after bands 0 and 1, the FIFO lanes 52..63 that hold the upcoming events and k-mers,
the re-entry loop, the expansion of the walk's codes) is mirrored below line by line, the statements themselves are taken from the generated files as they are compiled."""
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gfx950_emu import Wave, M32, M64   # noqa: E402
import asm_lint                          # noqa: E402
from f5c_amd import load_model_f32, EVENT_DT   # noqa: E402
from oracle import orc                   # noqa: E402

CSRC = os.path.join(ROOT, "f5c_amd", "csrc")
NINF32 = 0xFF800000
NINF64 = 0xFFF0000000000000
KPAR_DT = np.dtype([("gpm", "<f4"), ("ck", "<f4"), ("istd", "<f8")])
FILL = asm_lint.statement(os.path.join(CSRC, "abea_fill.inc"), "ABEA_FILL_ASM")
WALK = asm_lint.statement(os.path.join(CSRC, "abea_walk.inc"), "ABEA_WALK_ASM")


def f32bits(x):
    return int(np.float32(x).view(np.uint32))


def f64bits(x):
    return int(np.float64(x).view(np.uint64))


def emulate_align(seq: bytes, means: np.ndarray, model, k: int, scale, shift, fill=None, walk=None):
    """abea_pre_kernel + abea_align_kernel for one read; returns (pairs [n, 2] int32 in ascending order, info)."""
    L, E = len(seq), len(means)
    K = L - k + 1
    lane = np.arange(64)
    # ---- abea_pre_kernel (abea_kernels.hip): per-k-mer parameters
    code = np.array([{65: 0, 67: 1, 71: 2, 84: 3}.get(c, 0) for c in seq], dtype=np.int64)
    rank = np.zeros(K, dtype=np.int64)
    for j in range(k):
        rank = (rank << 2) | code[j:j + K]
    m = model[rank]
    kpar = np.zeros(K, dtype=KPAR_DT)
    kpar["gpm"] = (np.float32(scale) * m["level_mean"]).astype(np.float32) + np.float32(shift)
    kpar["ck"] = np.float32(-0.918938) - m["level_log_stdv"]
    kpar["istd"] = 1.0 / m["level_stdv"].astype(np.float64)
    evm = np.ascontiguousarray(means, dtype=np.float32)
    # ---- plan_desc (abea_capi.cpp)
    n_bands = E + K + 2
    n_groups = (n_bands + 31) // 32
    nb_pad = n_groups * 32
    p_stay = 1 - (1 / (E / K + 1))
    lp_skip = math.log(1e-10)
    lp_stay = math.log(p_stay)
    lp_step = math.log(1.0 - math.exp(lp_skip) - math.exp(lp_stay))
    lp_trim = math.log(0.01)

    w = Wave()
    a_evm = w.alloc(evm)
    a_kpar = w.alloc(kpar)
    a_trace = w.alloc(np.zeros(n_groups * 64 * 4, dtype=np.uint32))
    a_codes = w.alloc(np.zeros((E + K) // 16 + 8, dtype=np.uint32))
    def ev(i):
        return evm[np.minimum(i, E - 1)]

    def kp(i):
        return kpar[np.minimum(i, K - 1)]

    # ---- phase 1: the state after bands 0 and 1
    ll_e, ll_k = 50, -51
    Pf0 = np.full(64, NINF32, dtype=np.uint32); Pf1 = Pf0.copy()
    Pf0[25] = f32bits(np.float32(lp_trim))
    U0 = np.full(64, NINF64, dtype=np.uint64); U1 = U0.copy(); L0 = U0.copy(); L1 = U0.copy()
    U0[25] = 0; L1[25] = 0
    o0, o1 = 2 * lane, 2 * lane + 1

    def ev_or0(e):
        return np.where((e >= 0) & (e < E), evm[np.clip(e, 0, E - 1)], np.float32(0)).astype(np.float32)

    def kp_or0(kk):
        out = kpar[np.clip(kk, 0, K - 1)].copy()
        bad = ~((kk >= 0) & (kk < K))
        out["gpm"][bad] = 0; out["ck"][bad] = 0; out["istd"][bad] = 0
        return out
    x0, x1 = ev_or0(ll_e - o0), ev_or0(ll_e - o1)
    p0, p1 = kp_or0(ll_k + o0), kp_or0(ll_k + o1)
    acc = np.full(64, 0xFF, dtype=np.uint32); acc[25] = 0xFE

    def bits(a):
        return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

    def bits64(a):
        return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    # lanes 52..63 of the event registers hold the next 24 events (lane 63's cell 1 first); the pending registers hold what the
    # next refill, 24 moves of a kind later, will put into those lanes (abea_kernels.hip, phase 1 set-up)
    e_in = ll_e + 1
    q = 2 * (63 - lane)
    fl = lane >= 52
    x1 = np.where(fl, ev(e_in + q), x1).astype(np.float32); x0 = np.where(fl, ev(e_in + q + 1), x0).astype(np.float32)
    ka, kb = kp(np.maximum(ll_k + 24 + 2 * lane, 0)), kp(np.maximum(ll_k + 24 + 2 * lane + 1, 0))
    for name, val in (("px1", bits(ev(e_in + 24 + q))), ("px0", bits(ev(e_in + 24 + q + 1))), ("kag", bits(ka["gpm"])),
                      ("kac", bits(ka["ck"])), ("kbg", bits(kb["gpm"])), ("kbc", bits(kb["ck"]))):
        w.bind_v(name, val)
    w.bind_v("kai", bits64(ka["istd"]), wide=True); w.bind_v("kbi", bits64(kb["istd"]), wide=True)
    w.sym.update(e_cnt=24, k_cnt=24)
    for name, val in (("Pf0", Pf0), ("Pf1", Pf1), ("x0", bits(x0)), ("x1", bits(x1)), ("g0", bits(p0["gpm"])), ("c0", bits(p0["ck"])),
                      ("g1", bits(p1["gpm"])), ("c1", bits(p1["ck"])),
                      ("a1", np.zeros(64, np.uint32)), ("a2", np.zeros(64, np.uint32)), ("a3", np.zeros(64, np.uint32)),
                      ("acc", acc), ("toff", np.zeros(64, np.uint32)), ("lane", lane.astype(np.uint32))):
        w.bind_v(name, val)
    for name, val in (("i0", bits64(p0["istd"])), ("i1", bits64(p1["istd"])), ("L0", L0), ("L1", L1), ("U0", U0), ("U1", U1)):
        w.bind_v(name, val, wide=True)
    S = w.sym
    S.update(lp_step=f64bits(lp_step), lp_stay=f64bits(lp_stay), lp_skip=f64bits(lp_skip), lp_trim=f64bits(lp_trim),
             Km1=K - 1, Em1=E - 1, m50=1 << 50, evm=a_evm, kpar=a_kpar, trace=a_trace,
             ninf=NINF32, mvacc=0, mvprev=0, best=NINF32, best_e=0, best_llk=0)
    for t in ("t0", "t1", "t2", "t3", "t4", "cnt", "per", "cm0a", "cm0b", "cm1a", "cm1b", "cv0", "cv1"):
        S[t] = 0xDEADBEEF                                        # write-only operands: garbage at entry
    b = 2
    entries = 0
    while b < nb_pad:
        run = min(E - 2 - ll_e, K - 102 - ll_k, nb_pad - b)
        interior = ll_k >= 0 and ll_e >= 99 and run > 0
        past_edge = (E - 2 - ll_e <= 0) or (K - 102 - ll_k <= 0)
        b_end = b + run if interior else (nb_pad if past_edge else min(nb_pad, b + max(max(-ll_k, 99 - ll_e), 256)))
        w.bind_v("toff", (lane * 16 + (b >> 5) * 1024).astype(np.uint32))
        S.update(ll_e=ll_e & M32, ll_k=ll_k & M32, b=b, b_end=b_end, mode=0 if interior else 1)
        w.run(fill or FILL)
        ll_e = S["ll_e"] - (1 << 32) if S["ll_e"] & 0x80000000 else S["ll_e"]
        ll_k = S["ll_k"] - (1 << 32) if S["ll_k"] & 0x80000000 else S["ll_k"]
        assert S["b"] > b, "the statement made no progress"
        b = S["b"]
        entries += 1
    best = np.uint32(S["best"]).view(np.float32)
    info = dict(instructions=w.n_exec, entries=entries, best=float(best), best_e=S["best_e"], bands=n_bands)
    if S["best"] == NINF32:
        return np.zeros((0, 2), np.int32), info

    # ---- phase 2: the walk
    w.bind_v("o_cv", np.zeros(64, np.uint32))
    llk0 = S["best_llk"]
    S.update(k0=K - 1, e0=S["best_e"], llk0=llk0, codes=a_codes)
    for t in ("last_k", "o_cwd", "o_sh2", "o_nfl", "o_maxgap", "o_reloads"):
        S[t] = 0xDEADBEEF
    n0 = w.n_exec
    w.run(walk or WALK)
    n = S["o_nfl"] * 16 + S["o_sh2"] // 2
    cv = w.get_v("o_cv")
    if n & 15:
        cv[(n >> 4) & 63] = S["o_cwd"]
    codes = w.view(a_codes, np.uint32, (E + K) // 16 + 8)
    if n & 1023:
        last = ((n - 1) >> 4) & 63
        codes[(n >> 10) * 64: (n >> 10) * 64 + last + 1] = cv[:last + 1]
    # ---- phase 3: expansion (ascending order)
    pairs = np.zeros((n, 2), dtype=np.int32)
    kk, ee = K - 1, S["best_e"]
    for j in range(n):
        pairs[n - 1 - j] = (kk, ee)
        cd = (int(codes[j >> 4]) >> (2 * (j & 15))) & 3
        kk -= cd != 1
        ee -= cd != 2
    info.update(walk_instructions=w.n_exec - n0, n=n, last_k=S["last_k"], max_gap=S["o_maxgap"], reloads=S["o_reloads"],
                mix=dict(w.count))
    return pairs, info


def synthetic_read(rng, model, k, n_bases, events_per_base=2.0, noise=1.0):
    seq = bytes(rng.choice(list(b"ACGT"), size=n_bases).tolist())
    K = n_bases - k + 1
    rank = np.array([orc.kmer_rank(seq[i:i + k], k) for i in range(K)])
    reps = np.maximum(1, rng.poisson(events_per_base, size=K))
    reps[rng.random(K) < 0.04] = 0                               # skipped k-mers
    idx = np.repeat(np.arange(K), reps)
    means = (model["level_mean"][rank[idx]] + rng.normal(0, noise, size=len(idx)) * model["level_stdv"][rank[idx]]).astype(np.float32)
    ev = np.zeros(len(means), dtype=EVENT_DT)
    ev["mean"] = means
    return seq, ev


def check_against_oracle(seq, ev, model, k, fill=None, walk=None):
    scale, shift = orc.estimate_scalings(seq, model, k, ev)
    o_pairs, o_diag = orc.align(seq, ev, model, k, scale, shift)
    pairs, info = emulate_align(seq, ev["mean"], model, k, scale, shift, fill, walk)
    assert np.float32(info["best"]) == np.float32(o_diag["max_score"])           # bit for bit: the end-point scan's maximum
    if info["best"] > -np.inf:
        assert info["best_e"] == int(o_diag["best_event"])
        assert info["n"] == int(o_diag["n_aligned"]) and info["max_gap"] == int(o_diag["max_gap"])
        assert (info["last_k"] == 0) == bool(o_diag["spanned"])
    if len(o_pairs):                                             # passed QC: the lists must be the same, pair for pair
        assert (pairs == o_pairs.view(np.int32).reshape(-1, 2)).all()
    info["qc_pass"] = len(o_pairs) > 0
    return info


MODEL = load_model_f32(os.path.join(ROOT, "tests", "golden", "r9.4_450bps.6mer.f32"))
CASES = [  # (seed, bases, events per base): 3 k-mers / border variant only / interior stretches with ring refills and trace stores
    (7, 8, 2.0), (1, 60, 2.0), (3, 260, 1.6), (4, 420, 2.2),
]


@pytest.mark.parametrize("seed,n_bases,epb", CASES)
def test_emulated_statements_reproduce_the_oracle(seed, n_bases, epb):
    k, model = MODEL
    seq, ev = synthetic_read(np.random.default_rng(seed), model, k, n_bases, epb)
    info = check_against_oracle(seq, ev, model, k)
    assert info["qc_pass"]
    mix = info["mix"]
    if n_bases >= 400:     # what the run went through: the interior variant (only it packs trace bits with v_alignbit), both FIFO
        assert mix["v_alignbit_b32"] >= 4 * 500            # refills (every 24 moves of a kind), group stores, the border variant's selects
        assert mix["global_load_dwordx4"] >= 2 * 10 and mix["global_load_dword"] >= 2 * 20 and mix["global_store_dwordx4"] >= 40
        assert not any(op.startswith("ds_") for op in mix)  # the fill loop does not touch LDS
        assert mix["v_cndmask_b32"] > 1000 and info["entries"] >= 3


def test_emulated_statements_on_hostile_reads():
    """Signal unrelated to the sequence (the path wanders: the walk leaves its prefetch window and re-loads whole trace
    groups; QC fails), a stuck stretch of events, a doubled prefix: counts, end point, score and gaps still agree."""
    k, model = MODEL
    reloads = 0
    for seed, n_bases, kind in ((11, 150, "noise"), (12, 230, "noise"), (13, 150, "stuck"), (14, 110, "doubled")):
        rng = np.random.default_rng(seed)
        seq, ev = synthetic_read(rng, model, k, n_bases, 2.0)
        if kind == "noise":
            ev["mean"] = rng.normal(90, 12, size=len(ev)).astype(np.float32)
        elif kind == "stuck":
            ev["mean"][len(ev) // 3: len(ev) // 2] = ev["mean"][len(ev) // 3]
        else:
            ev = np.concatenate([ev[:20], ev[:20], ev[20:]])
        info = check_against_oracle(seq, ev, model, k)
        reloads += info.get("reloads", 0)
    assert reloads > 0


def test_a_planted_fault_in_the_statement_is_noticed():
    """The check has teeth: one changed instruction in the interior loop (the tie-break difference of cell 1 taken the other
    way round; the up conversion of the shifted score dropped) changes the result."""
    k, model = MODEL
    seq, ev = synthetic_read(np.random.default_rng(4), model, k, 420, 2.2)
    import re
    swapped = [re.sub(r"^v_sub_f32 v113, (v\d+), (v\d+)$", r"v_sub_f32 v113, \2, \1", ln) for ln in FILL]
    assert swapped != FILL
    with pytest.raises(AssertionError):
        check_against_oracle(seq, ev, model, k, fill=swapped)
    skewed = [ln.replace("v_fma_f32 v114, -0.5, v114", "v_fma_f32 v114, 0.5, v114") for ln in FILL]
    assert skewed != FILL
    with pytest.raises(AssertionError):
        check_against_oracle(seq, ev, model, k, fill=skewed)


def test_the_shipped_statements_are_the_round_4_design():
    """No LDS in the fill loop (the upcoming events and k-mers wait in lanes 52..63, rotated in with wave_ror), the walk's step
    loop exists per trace-group half; more reads through both, among them a hostile one."""
    assert not any(ln.startswith("ds_") for ln in FILL) and any("wave_ror:1" in ln for ln in FILL)
    assert any(ln.startswith("step_hi") for ln in WALK) and any(ln.startswith("step_lo") for ln in WALK)
    k, model = MODEL
    for seed, n_bases, epb, kind in ((9, 330, 3.0, ""), (12, 230, 2.0, "noise"), (21, 500, 1.1, "")):
        rng = np.random.default_rng(seed)
        seq, ev = synthetic_read(rng, model, k, n_bases, epb)
        if kind == "noise":
            ev["mean"] = rng.normal(90, 12, size=len(ev)).astype(np.float32)
        check_against_oracle(seq, ev, model, k)


def test_emulated_statements_on_a_real_read_against_the_reference_golden():
    """A real nanopore read of the reference's test set (tests/golden/ecoli_events.npz, the shortest: 1234 bases, 2047 events):
    the emulated statements give the oracle's pair list and the n_aligned_events the REFERENCE printed in adaptive.exp."""
    k, model = MODEL
    z = np.load(os.path.join(ROOT, "tests", "golden", "ecoli_events.npz"))
    i = int(np.argmin(np.diff(z["ev_ptr"])))
    seq = bytes(z["seq"][z["seq_ptr"][i]:z["seq_ptr"][i + 1]])
    ev = np.zeros(int(z["ev_ptr"][i + 1] - z["ev_ptr"][i]), dtype=EVENT_DT)
    ev["mean"] = z["mean"][z["ev_ptr"][i]:z["ev_ptr"][i + 1]]
    scale, shift = float(z["scale"][i]), float(z["shift"][i])
    o_pairs, o_diag = orc.align(seq, ev, model, k, scale, shift)
    pairs, info = emulate_align(seq, ev["mean"], model, k, scale, shift)
    assert info["n"] == int(z["printed_n"][i]) == int(o_diag["n_aligned"])
    assert len(o_pairs) and (pairs == o_pairs.view(np.int32).reshape(-1, 2)).all()
    assert np.float32(info["best"]) == np.float32(o_diag["max_score"])
