cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from f5c_amd import abea, synth, load_model_f32
k, model = load_model_f32("tests/golden/r9.4_450bps.6mer.f32")
cfg = synth.CONFIGS["r9_100k_mixed"]
b = synth.make_batch(cfg["n_reads"], model, k, seed=cfg["seed"], law=cfg["law"], workers=16)
ev = int(b["n_events"].sum())
for rnd in range(2):
    for hi in ("1", "0"):
        os.environ["ABEA_HOST_HI_STREAM"] = hi
        ctx = abea.AbeaContext(model, k, max_arena_bytes=150 << 30)
        for mode in ("pairs", "fused"):
            v = ctx.host_view(b, scaling=(mode == "fused"), want_pairs=(mode == "pairs"))
            for rep in range(3):
                t0 = time.perf_counter(); ctx.align_view(v); t = time.perf_counter() - t0
                st = ctx.stats()
                if rep:
                    print(f"hi_stream={hi} {mode:5s} rep {rep}: {ev/t/1e6:7.1f} Mevents/s wall {t*1e3:6.1f} | flatten {st['flatten_ms']:6.1f} unflatten {st['unflatten_ms']:6.1f} wait {st['wait_ms']:6.1f} | scaling {st['trace_ms']:6.1f}", flush=True)
            del v
        ctx.close()
PY
