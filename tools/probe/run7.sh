export MODES=pairs DEVICE=0 ARENA_GIB=150
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
def run(label, reps=3, **env):
    for kk, vv in env.items(): os.environ[kk] = str(vv)
    best = None
    for r in range(reps):
        t0 = time.perf_counter(); ctx.align_view(v); t = time.perf_counter() - t0
        st = ctx.stats()
        if r: best = min(best or t, t)
    print(f"{label:40s} {ev/best/1e6:8.1f} Mevents/s wall {best*1e3:7.1f} | flatten {st['flatten_ms']:6.1f} unflatten {st['unflatten_ms']:6.1f} wait {st['wait_ms']:6.1f} chunks {st['n_sub_batches']} thr {st['host_threads']} | kernels pre {st['pre_ms']:.0f} align {st['fill_ms']:.0f} scal {st['trace_ms']:.0f} d2h {st['d2h_bytes']/1e9:.1f} GB", flush=True)
    for kk in env: os.environ.pop(kk)
run("default")
run("slots 6", ABEA_HOST_SLOTS=6)
run("slots 8", ABEA_HOST_SLOTS=8)
run("slots 8, 64M", ABEA_HOST_SLOTS=8, ABEA_HOST_CHUNK_EVENTS=64 << 20)
run("slots 6, 96M", ABEA_HOST_SLOTS=6, ABEA_HOST_CHUNK_EVENTS=96 << 20)
run("slots 8, 1024 reads 32M", ABEA_HOST_SLOTS=8, ABEA_HOST_CHUNK_EVENTS=32 << 20, ABEA_HOST_CHUNK_READS=1024)
v2 = ctx.host_view(b, scaling=True, want_pairs=False)
v, v1 = v2, v
run("fused scaling, no pairs")
run("fused scaling, no pairs, slots 8", ABEA_HOST_SLOTS=8)
os.environ["ABEA_HOST_TRACE"]="1"; ctx.align_view(v)
PY
