import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5c_amd import synth, load_model_f32
from oracle import orc
k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
b = synth.make_batch(1024, model, k, seed=20250002, law="gamma8k", workers=32)
ev = int(b["n_events"].sum())
for t in (1, 8, 16, 32, 64, 128, 256):
    sub = synth.take_reads(b, np.arange(min(1024, max(8, 4 * t))))
    t0 = time.perf_counter(); orc.align_batch(sub, model, k, n_threads=t, want_diag=False); dt = time.perf_counter() - t0
    print(t, "threads", round(sub["n_events"].sum() / dt / 1e6, 3), "Mev/s", round(dt, 2), "s", flush=True)
