#!/usr/bin/env python3
"""CPU analysis for the traceback's trace read-back ("reading the trace back through a window"): how far does
the path move across lane pairs between the moment a 32-band trace group is prefetched (one group ahead) and the moment
the walk is done with it?  Replays the oracle's alignments of synthetic reads with the band lower-left positions exported by
the oracle's analysis hook and simulates the windowed prefetch for several radii: fraction of groups that would need a
whole-group reload, and trace bytes fetched per band (64-byte sectors, window + the move lane's sector).
usage: python tools/walk_drift.py [n_reads] [read_len]"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5c_amd import synth, load_model_f32
from oracle import orc

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
b = synth.make_batch(n_reads, model, k, seed=5, law=read_len, bad_frac=0.0)
L = orc.lib()
L.orc_debug_band_llk.restype = None
L.orc_debug_band_llk.argtypes = [C.c_void_p, C.c_size_t]
RADII = (2, 4, 6, 8, 12, 16, 26)
stats = {r: dict(groups=0, reloads=0, sectors=0) for r in RADII}
bands_total = 0
for i in range(n_reads):
    s, Ln = int(b["read_ptr"][i]), int(b["read_len"][i])
    es, E = int(b["event_ptr"][i]), int(b["n_events"][i])
    nb = E + (Ln - k + 1) + 2
    llk = np.zeros(nb, dtype=np.int32)
    L.orc_debug_band_llk(llk.ctypes.data, nb)
    pairs, d = orc.align(b["reads"][s:s + Ln].tobytes(), b["events"][es:es + E], model, k,
                         b["scalings"]["scale"][i], b["scalings"]["shift"][i])
    L.orc_debug_band_llk(None, 0)
    if len(pairs) == 0:
        continue
    kk = pairs["ref_pos"].astype(np.int64)[::-1]; ee = pairs["read_pos"].astype(np.int64)[::-1]      # walk order
    band = kk + ee + 2
    lp = (kk - llk[band]) >> 1                                  # lane pair of the path cell in its band
    grp = band >> 5
    bands_total += int(band[0] - band[-1] + 1)
    # first step index of every group along the walk
    first = np.nonzero(np.concatenate([[True], grp[1:] != grp[:-1]]))[0]
    for r in RADII:
        st = stats[r]
        for gi in range(len(first)):
            lo_i = first[gi]; hi_i = first[gi + 1] if gi + 1 < len(first) else len(grp)
            st["groups"] += 1
            if gi == 0:
                st["sectors"] += 16                              # the first group is loaded whole
                continue
            c = lp[first[gi - 1]]                                # window centre: the path's lane pair when the prefetch was issued
            lo = max(c - r, 0) & ~3; hi = min(c + r, 51) | 3
            sect = (hi - lo + 1) // 4 + (0 if lo <= 48 <= hi else 1)     # + the move lane's sector (lanes 48..51)
            seg = lp[lo_i:hi_i]
            if (seg < lo).any() or (seg > hi).any():
                st["reloads"] += 1; sect += 16
            st["sectors"] += sect
print(f"{n_reads} reads of {read_len} bases, {bands_total} bands walked")
print("radius (lane pairs) | groups needing a whole-group reload | trace bytes fetched per band (whole groups: 32.0)")
for r in RADII:
    st = stats[r]
    print(f"  {r:3d}               | {100.0 * st['reloads'] / max(1, st['groups']):6.2f} %                            | {st['sectors'] * 64 / max(1, bands_total):5.1f}")
