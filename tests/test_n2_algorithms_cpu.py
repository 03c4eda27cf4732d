"""CPU evidence for the two exactness arguments the device event detection (row N2) rests on.  No GPU, no product
code: (1) the segment-parallel form of the peak automaton, as prototyped in tools/proto/spec_detect.c, reproduces
the sequential automaton on real and adversarial signals; (2) when the exponent-range test of abea_ev_pscan_kernel
passes, fp64 sums of the samples are the same in every order."""
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pa(raw, sc):
    off, rng, dig = (np.float32(x) for x in sc)
    return ((raw.astype(np.float32) + off) * (rng / dig)).astype(np.float32)


def test_segment_parallel_automaton_prototype(tmp_path):
    exe = tmp_path / "spec_detect"
    subprocess.check_call(["gcc", "-O2", "-w", "-o", str(exe), os.path.join(ROOT, "tools/proto/spec_detect.c"), "-lm", "-lpthread"])
    g = np.load(os.path.join(ROOT, "tests/golden/ecoli_reads.npz"), allow_pickle=True)
    r = np.random.default_rng(5)
    sc0 = (10.0, 1467.61, 8192.0)
    sigs = [_pa(g[f"sig{i}"], g["scaling"][i]) for i in range(int(g["n"]))]
    sigs += [_pa(np.full(5000, 500, np.int16), sc0), _pa(np.full(7, 500, np.int16), sc0), _pa(np.full(1, 500, np.int16), sc0),
             _pa(r.integers(300, 700, 100000).astype(np.int16), sc0),
             _pa(np.repeat(r.integers(300, 700, 400), 250).astype(np.int16), sc0),
             _pa((np.repeat(r.integers(300, 700, 30000), 3) + r.integers(-2, 3, 90000)).astype(np.int16), sc0),
             _pa((500 + 100 * np.sin(np.arange(200000) / 50.0)).astype(np.int16), sc0)]
    path = tmp_path / "sigs.bin"
    with open(path, "wb") as f:
        for pa in sigs:
            f.write(np.int64(len(pa)).tobytes()); f.write(np.ascontiguousarray(pa).tobytes())
    for G, F in ((512, 64), (128, 64), (1024, 128)):          # the device uses 512 / 64
        out = subprocess.check_output([str(exe), str(path), str(G), str(F)], text=True)
        head = dict(kv.split("=") for kv in out.splitlines()[0].split())
        assert int(head["mismatches"]) == 0, out                # exit status is non-zero on a mismatch as well
        assert int(head["reads"]) == len(sigs)
        assert float(head["mean_sync"]) < 8.0                   # segments meet their replay within a few samples
        assert int(head["fallback_reads"]) <= 3                 # plateaus / slow waves: sequential fallback, by design


def _exact_by_rule(x):
    """The decision of abea_ev_pscan_kernel for one array of floats (all entries, not squares)."""
    ax = np.abs(x).view(np.uint32)
    nz = ax[ax != 0]
    if len(nz) == 0:
        return True
    if nz.max() >= 0x7f800000:
        return False
    emin, emax = max(int(nz.min() >> 23), 1), max(int(nz.max() >> 23), 1)
    nbits = int(len(x)).bit_length()
    return (emax - emin) + nbits + 24 <= 53


def test_exponent_range_rule_implies_order_independent_sums():
    r = np.random.default_rng(11)
    n_pass = n_fail_caught = 0
    for trial in range(300):
        n = int(r.integers(2, 200000))
        kind = trial % 4
        if kind == 0:   x = r.uniform(40, 200, n)
        elif kind == 1: x = r.uniform(30, 300, n) * r.choice([-1.0, 1.0], n)
        elif kind == 2: x = np.concatenate([r.uniform(60, 130, n - 1), [r.uniform(1e-6, 1e-3)]])
        else:           x = r.uniform(1, 2, n) * 2.0 ** r.integers(-20, 20, n)
        x = x.astype(np.float32)
        seq = 0.0
        for v in x[:2000]:                                      # python loop only for a prefix; numpy cumsum is sequential too
            seq += float(v)
        assert seq == float(np.cumsum(x[:2000].astype(np.float64))[-1])
        total_seq = float(np.cumsum(x.astype(np.float64))[-1])
        perm = r.permutation(n)
        total_perm = float(np.cumsum(x[perm].astype(np.float64))[-1])
        pair = x.astype(np.float64)
        while len(pair) > 1:                                    # pairwise tree, a third association
            if len(pair) & 1: pair = np.concatenate([pair, [0.0]])
            pair = pair[0::2] + pair[1::2]
        if _exact_by_rule(x):
            n_pass += 1
            assert total_seq == total_perm == float(pair[0]), (trial, n)
        elif total_seq != total_perm or total_seq != float(pair[0]):
            n_fail_caught += 1
    assert n_pass >= 100 and n_fail_caught >= 5                 # the rule accepts ordinary signals and rejects ones that do round


def test_float_sums_in_double_are_order_free_under_the_exponent_test():
    """The criterion abea_ev_scalings_kernel uses (round 6) before it lets 64 lanes add strided subsets of a read's event means /
    k-mer levels and finish with a butterfly: all terms are floats, so multiples of the smallest term's ulp 2^(emin-150), and any
    partial sum is below n * 2^(emax-126); when ceil(log2 n) + emax - emin <= 29 no double addition can round, in ANY order —
    the strided / butterfly sum is the sequential sum of estimate_scalings_using_mom (align.c:66-81) bit for bit.  Checked here on
    random inputs on both sides of the bound: inside it every order agrees exactly; outside it a counter-example exists."""
    import numpy as np
    rng = np.random.default_rng(99)

    def expo(x):
        e = (x.view(np.uint32) >> 23) & 0xFF
        nz = x != 0
        return (int(np.maximum(e[nz], 1).min()), int(e[nz].max())) if nz.any() else (255, 0)

    def clog2(n):
        return int(n - 1).bit_length() if n > 1 else 0

    def seq(x):
        s = 0.0
        for v in x.astype(np.float64):
            s += v
        return s

    def lanes(x):                                         # 64 strided partial sums, then a butterfly
        p = [seq(x[i::64]) for i in range(64)]
        off = 32
        while off:
            p = [p[i] + p[i ^ off] for i in range(64)]
            off >>= 1
        return p[0]

    n_exact = 0
    for trial in range(60):
        n = int(rng.integers(1, 5000))
        kind = trial % 3
        if kind == 0:       x = rng.normal(90.0, 15.0, n)                          # event means in pA
        elif kind == 1:     x = rng.uniform(-1.0, 1.0, n) * 10.0 ** rng.integers(-3, 6)
        else:               x = np.concatenate([rng.normal(100.0, 10.0, n), [0.0, -0.0]])
        x = x.astype(np.float32)
        lo, hi = expo(x)
        if hi < 255 and clog2(len(x)) + hi - lo <= 29:
            n_exact += 1
            assert seq(x) == lanes(x) == seq(x[::-1]) == float(np.sum(x.astype(np.float64)))
    assert n_exact >= 30
    # outside the bound the orders can differ: one tiny term next to large ones
    x = np.array([1e-9] + [100.0] * 4095, dtype=np.float32)
    lo, hi = expo(x)
    assert clog2(len(x)) + hi - lo > 29
    y = np.array([2.0 ** -30, 2.0 ** 30, -2.0 ** 30] * 64, dtype=np.float32)     # the classic: cancellation order matters
    lo, hi = expo(y)
    assert clog2(len(y)) + hi - lo > 29 and seq(y) != lanes(y)


def test_sliding_window_and_event_sums_equal_the_prefix_sum_differences():
    """What the common path of the device detector rests on (abea_ev_spec2_kernel / abea_ev_create3_kernel, round 6): where the
    exponent-range rule holds for a read — for the samples AND for their float squares — the reference's sums[b] - sums[a]
    (events.c:303-313, 343-366, 497-513: sequential fp64 prefix sums, then a difference) is, bit for bit, (1) the window sum kept by
    sliding (+ entering sample, - leaving sample) and (2) the plain sum of the samples of [a, b) — so neither the prefix-sum array
    nor the t-statistic array has to exist.  Real reads, synthetic steps and noise; DNA and RNA window lengths."""
    g = np.load(os.path.join(ROOT, "tests/golden/ecoli_reads.npz"), allow_pickle=True)
    r = np.random.default_rng(23)
    sigs = [(g[f"sig{i}"], g["scaling"][i]) for i in range(min(4, int(g["n"])))]
    sigs.append((np.repeat(r.integers(350, 750, 3000), r.integers(1, 12, 3000))[:20000].astype(np.int16), (10.0, 1467.61, 8192.0)))
    sigs.append((r.integers(-32768, 32767, 20000).astype(np.int16), (3.0, 748.58, 2048.0)))
    checked = 0
    for raw, sc in sigs:
        x = _pa(raw, sc)
        y = (x * x).astype(np.float32)                            # the float product of events.c:312
        if not (_exact_by_rule(x) and _exact_by_rule(y)):
            continue
        for v in (x, y):
            d = v.astype(np.float64)
            S = np.concatenate([[0.0], np.cumsum(d)])                # sequential, like the reference
            n = len(v)
            for w in (3, 6, 7, 14):
                if n < 2 * w + 2:
                    continue
                ref_left = S[w:n - w + 1] - S[0:n - 2 * w + 1]         # sum over [p - w, p) for p = w .. n - w
                ref_right = S[2 * w:n + 1] - S[w:n - w + 1]            # sum over [p, p + w)
                left = d[0:w].sum(); right = d[w:2 * w].sum()
                got_l, got_r = [left], [right]
                for p in range(w, n - w):                              # p -> p + 1
                    left = (left + d[p]) - d[p - w]
                    right = (right + d[p + w]) - d[p]
                    got_l.append(left); got_r.append(right)
                assert np.array_equal(np.array(got_l).view(np.uint64), ref_left.view(np.uint64)), w
                assert np.array_equal(np.array(got_r).view(np.uint64), ref_right.view(np.uint64)), w
            cuts = np.sort(r.choice(np.arange(1, n), size=min(2000, n - 1), replace=False))
            bounds = np.concatenate([[0], cuts, [n]])
            for a, b in zip(bounds[:-1], bounds[1:]):                  # "events"
                s = 0.0
                for t in d[a:b]:
                    s += t
                assert s == S[b] - S[a]
            checked += 1
    assert checked >= 8


def test_division_by_the_window_length_in_three_instructions(tmp_path):
    """tools/proto/div_const_check.c, thinned to a test: q0 = x * rc; r = fma(-q0, c, x); q = fma(r, rc, q0) with rc = RN(1 / c) is
    the IEEE quotient x / c for c in {3, 6, 7, 14} on every 61st float bit pattern (the full 2^32 sweep: 73 s, 0 mismatches outside
    x = +-inf and quotients below 4 FLT_MIN — the cases the kernel sends to the IEEE division) and on 2^24 sampled doubles per
    divisor and exponent."""
    src = tmp_path / "divc.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
int main(void) {
    const float cs[4] = {3.f, 6.f, 7.f, 14.f};
    unsigned long long bad = 0, guarded = 0, n = 0;
    for (int k = 0; k < 4; ++k) {
        const float c = cs[k]; volatile float one = 1.0f; const float rc = one / c;
        for (uint64_t i = 0; i < (1ull << 32); i += 61) {
            uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
            if (x != x) continue;
            const float ax = fabsf(x);
            if (!(ax >= 0x1p-118f && ax < INFINITY)) { ++guarded; continue; }      /* abea_tstat_fast's guard */
            const float ref = x / c, q0 = x * rc, r = fmaf(-q0, c, x), q = fmaf(r, rc, q0);
            if (memcmp(&q, &ref, 4)) ++bad;
            ++n;
        }
        const double cd = c; volatile double oned = 1.0; const double rcd = oned / cd;
        static const int ex[5] = {1023 - 149, 1023 - 20, 1023, 1023 + 20, 1023 + 132};
        for (uint64_t i = 0; i < (1ull << 24); ++i) {
            uint64_t z = i * 0x9E3779B97F4A7C15ull + k;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
            for (int e = 0; e < 5; ++e) {
                uint64_t u = ((uint64_t)ex[e] << 52) | (z & 0xFFFFFFFFFFFFFull) | ((z >> 63) << 63);
                double x; memcpy(&x, &u, 8);
                const double ref = x / cd, q0 = x * rcd, r = fma(-q0, cd, x), q = fma(r, rcd, q0);
                if (memcmp(&q, &ref, 8)) ++bad;
                ++n;
            }
        }
    }
    printf("%llu %llu %llu\n", bad, guarded, n);
    return 0;
}
''')
    exe = tmp_path / "divc"
    hw_fma = " fma " in open("/proc/cpuinfo").read().replace("\n", " ")       # without it libm's fma() is exact too, only slower
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off"] + (["-mfma"] if hw_fma else []) + ["-o", str(exe), str(src), "-lm"])
    bad, guarded, n = (int(t) for t in subprocess.check_output([str(exe)]).split())
    assert bad == 0 and n > 500_000_000 and guarded > 1_000_000
