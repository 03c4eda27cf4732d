#!/usr/bin/env python3
"""Throughput of the profile-HMM entry (row N4): synthetic CpG windows shaped like call-methylation's (meth.c:402-474:
a group of CpG sites plus 5 flanking bases either side, i.e. 11-40 k-mers, ~2.2 events per k-mer, two scores per
group).  Prints jobs/s, matrix cells/s and the kernel time; numbers quoted in DESIGN.md §7."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from f5c_amd import abea, load_model_f32
from f5c_amd.types import EVENT_DT
from test_hmm_oracle import _cpg_model, _methylate, _rc_meth
k = 6
model = _cpg_model(k, 7)
_, r9 = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
r = np.random.default_rng(1)
ev = np.zeros(20000, dtype=EVENT_DT); ev["mean"] = r.normal(90, 12, len(ev)).astype(np.float32)
jobs, cells = [], 0
for g in range(n_groups):
    n_k = int(min(64, 6 + r.geometric(0.12)))
    seq = bytes(r.choice(list(b"ACGT"), n_k + k - 1).astype(np.uint8)); seq = seq[:5] + b"CG" + seq[7:]
    n_ev = max(2, int(2.2 * n_k + r.integers(-3, 4)))
    s = int(r.integers(0, len(ev) - n_ev))
    rc = bool(g & 1)
    for mseq in (seq, _methylate(seq)):
        jobs.append(dict(m_seq=mseq, m_rc_seq=_rc_meth(mseq), events=ev, scaling=(1.0, 0.5, 1.2, float(np.log(np.float32(1.2)))),
                         e_start=s + (n_ev - 1 if rc else 0), e_stop=s + (0 if rc else n_ev - 1), stride=-1 if rc else 1,
                         rc=rc, events_per_base=2.1, flags=3))
        cells += 3 * n_k * n_ev
ctx = abea.AbeaContext(r9, 6, max_arena_bytes=4 << 30)
ctx.hmm_score_batch(jobs[:1000], model, k)
for rep in range(3):
    t0 = time.perf_counter(); out = ctx.hmm_score_batch(jobs, model, k); t = time.perf_counter() - t0
    st = ctx.stats()
    print(f"{len(jobs)} jobs ({cells/1e6:.1f} M state cells): kernel {st['hmm_ms']:.2f} ms = {len(jobs)/st['hmm_ms']/1e3:.2f} M jobs/s, "
          f"{cells/st['hmm_ms']/1e6:.2f} G cells/s; library call {st['total_ms']:.1f} ms; with the Python job marshalling {t*1e3:.0f} ms")
print("finite scores:", float(np.isfinite(out).mean()))
