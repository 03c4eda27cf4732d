#!/usr/bin/env python3
"""The fused align + scaling_single host call for a rocprofv3 kernel trace (which kernel of a chunk waits for what).
Two-step so that no generator pool ever runs under the profiler:
    python tools/fused_trace.py 20000 /tmp/ft        # generates the batch (16 workers) and saves it
    rocprofv3 --kernel-trace ... -- python tools/fused_trace.py 20000 /tmp/ft     # loads it, runs pairs x3 and fused x3
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5c_amd import abea, synth, load_model_f32
n = int(sys.argv[1]); cache = sys.argv[2] + f".{n}.npz"
k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
if not os.path.exists(cache):
    cfg = synth.CONFIGS["r9_100k_mixed"]
    b = synth.make_batch(n, model, k, seed=cfg["seed"], law=cfg["law"], workers=16)
    np.savez(cache, **b)
    print("saved", cache, int(b["n_events"].sum()), "events")
    sys.exit(0)
z = np.load(cache)
b = {key: z[key] for key in z.files}
b["pair_cap"] = int(b["pair_cap"])
ev = int(b["n_events"].sum())
ctx = abea.AbeaContext(model, k, max_arena_bytes=120 << 30)
for mode in ("pairs", "fused"):
    v = ctx.host_view(b, scaling=(mode == "fused"), want_pairs=(mode != "fused"))
    for rep in range(3):
        t0 = time.perf_counter(); ctx.align_view(v); t = time.perf_counter() - t0
        st = ctx.stats()
        print(f"{mode:6s} rep {rep}: {ev/t/1e6:8.1f} Mevents/s wall {t*1e3:7.1f} ms | flatten {st['flatten_ms']:6.1f} unflatten {st['unflatten_ms']:6.1f} wait {st['wait_ms']:6.1f} "
              f"| kernels (sum over chunks) pre {st['pre_ms']:.1f} align {st['fill_ms']:.1f} gpu busy {st['gpu_busy_ms']:.1f} | {st['n_sub_batches']} chunks", flush=True)
    del v
ctx.close()
