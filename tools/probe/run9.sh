python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from f5c_amd import abea, synth, load_model_f32
k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
cfg = synth.CONFIGS["r9_100k_mixed"]
b = synth.make_batch(cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=16)
ev = int(b["n_events"].sum())
ctx = abea.AbeaContext(model, k, max_arena_bytes=150 << 30)
v = ctx.host_view(b)
ctx.align_view(v); ctx.align_view(v)
os.environ["ABEA_HOST_TRACE"] = "1"
t0 = time.perf_counter(); ctx.align_view(v); t = time.perf_counter() - t0
print("traced call", t * 1e3, "ms", ev / t / 1e6, "Mev/s", flush=True)
PY
