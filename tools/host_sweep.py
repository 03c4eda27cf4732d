#!/usr/bin/env python3
"""One process, one generated batch, many settings of the host entry (abea_align_batch_host): the library re-reads its
ABEA_HOST_* switches on every call, so a sweep costs one batch generation instead of one per setting.

  python tools/host_sweep.py --out gpurun_out/r05a [--config r9_100k_mixed] [--reads N] [--steps 4]

Writes <out>/sweep.json (one record per setting: ms per step and the caller thread's split into setup / plan / flatten /
un-flatten / wait, cgroup throttling deltas), <out>/trace_<name>.log (ABEA_HOST_TRACE timeline incl. the chunks' kernel
intervals on the GPU clock) for the settings marked `trace`, and <out>/flatten_threads.txt (the flatten loop alone:
GB/s against threads, prefetch distance and hint, on first-touch and NUMA-interleaved tables).  Not part of the
product or the tests; run on the GPU box through gpurun.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_stat():
    out = {}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            out[k] = int(v)
    except Exception:
        pass
    return out


def flatten_threads(lib, np, out_path, numa_interleave):
    """The flatten loop alone (abea_flatten_event_means), T python threads (ctypes releases the GIL), each over its own slice."""
    from f5c_amd.types import EVENT_DT
    lib.abea_flatten_event_means.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    n = int(os.environ.get("SWEEP_PROBE_EVENTS", 256 << 20))       # 256 M events = 6 GiB of event_t
    lines = []
    for placement in ("first-touch", "interleave"):
        inter = numa_interleave(True) if placement == "interleave" else False
        ev = np.zeros(n, dtype=EVENT_DT)
        ev["mean"] = 1.0
        if inter:
            numa_interleave(False)
        out = np.zeros(n, dtype=np.float32)
        out[:] = 0
        lib.abea_flatten_event_means(ev.ctypes.data, min(n, 1 << 24), out.ctypes.data, 0, 0)   # warm
        for T in (1, 8, 14, 16, 24, 32):
            for pf, hint in ((0, 0), (512, 0), (1536, 0), (4096, 0), (1536, 1), (1536, 2), (4096, 2)):
                per = n // T // 64 * 64

                def work(t):
                    lo = t * per
                    for c0 in range(lo, lo + per, 1 << 24):       # calls of <= 16 M events (int32 count, GIL released inside)
                        m = min(1 << 24, lo + per - c0)
                        lib.abea_flatten_event_means(ev.ctypes.data + c0 * 24, m, out.ctypes.data + c0 * 4, pf, hint)
                th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
                t0 = time.perf_counter()
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                dt = time.perf_counter() - t0
                lines.append(f"{placement:12s} threads {T:2d} prefetch {pf:5d} hint {hint}: {per * T * 24 / dt / 1e9:7.1f} GB/s read "
                             f"({per * T / dt / 1e6:8.0f} Mevents/s)")
                print(lines[-1], flush=True)
        del ev, out
    open(out_path, "w").write("\n".join(lines) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sweep")
    ap.add_argument("--config", default="r9_100k_mixed")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--no-flatten-probe", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated setting names")
    ap.add_argument("--hi-stream-ab", action="store_true", help="instead of the sweep: two contexts on the one batch, the second created "
                    "with ABEA_HOST_HI_STREAM=1 (the copy-out of a chunk's result block on a high-priority stream; read at slot creation), "
                    "alignment-only and fused calls alternating")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from f5c_amd import abea, synth, load_model_f32, synthetic_model

    lib = abea.load_library()
    info = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "cpu_max": open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None,
            "effective_cpus": bench.effective_cpus(), "hw_threads": os.cpu_count(),
            "numa_nodes": open("/sys/devices/system/node/online").read().strip() if os.path.exists("/sys/devices/system/node/online") else None}
    print(json.dumps(info), flush=True)
    if not args.no_flatten_probe:
        flatten_threads(lib, np, os.path.join(args.out, "flatten_threads.txt"), bench.numa_interleave)

    cfg = synth.CONFIGS[args.config]
    k = cfg["k"]
    model = load_model_f32(os.path.join(ROOT, "tests", "golden", "r9.4_450bps.6mer.f32"))[1] if k == 6 else synthetic_model(k, seed=9)
    n_total = args.reads or cfg["n_reads"]
    inter = bench.numa_interleave(True)
    t0 = time.time()
    batch = synth.make_batch(n_total, model, k, seed=cfg["seed"], law=cfg["law"], workers=max(1, min(16, bench.effective_cpus())))
    if inter:
        bench.numa_interleave(False)
    print(f"batch: {len(batch['read_len'])} reads, {int(batch['n_events'].sum())} events, {time.time() - t0:.0f} s, interleaved={inter}", flush=True)
    ev_total = int(batch["n_events"].sum())

    torch.cuda.set_device(0)
    if args.hi_stream_ab:
        ctxs = {}
        for name, val in (("own_stream", None), ("hi_stream", "1")):
            os.environ.pop("ABEA_HOST_HI_STREAM", None)
            if val:
                os.environ["ABEA_HOST_HI_STREAM"] = val
            c = abea.AbeaContext(model, k, device_id=0, max_arena_bytes=int(100 * (1 << 30)))
            vs = {"pairs": c.host_view(batch), "fused": c.host_view(batch, scaling=True, want_pairs=False)}
            for v in vs.values():
                c.align_view(v); c.align_view(v)          # creates the slots under this setting, warms staging
            ctxs[name] = (c, vs)
        os.environ.pop("ABEA_HOST_HI_STREAM", None)
        rows = []
        for rep in range(3):
            for name, (c, vs) in ctxs.items():
                for vname, v in vs.items():
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        c.align_view(v)
                    dt = (time.perf_counter() - t0) / args.steps
                    st = c.stats()
                    row = {"copy_out": name, "view": vname, "rep": rep, "ms_per_step": round(dt * 1e3, 2), "wait_ms": round(st["wait_ms"], 1),
                           "gpu_busy_ms": round(st["gpu_busy_ms"], 1)}
                    rows.append(row); print(json.dumps(row), flush=True)
        json.dump({"info": info, "rows": rows}, open(os.path.join(args.out, "hi_stream_ab.json"), "w"), indent=1)
        return
    ctx = abea.AbeaContext(model, k, device_id=0, max_arena_bytes=int(150 * (1 << 30)))
    views = {"pairs": ctx.host_view(batch), "fused": ctx.host_view(batch, scaling=True, want_pairs=False)}

    base = {"ABEA_HOST_FLATTEN_PREFETCH": "1536", "ABEA_HOST_FLATTEN_HINT": "1"}
    c24 = {"ABEA_HOST_CHUNK_EVENTS": str(24 << 20), "ABEA_HOST_CHUNK_READS": "1024"}
    c96 = {"ABEA_HOST_CHUNK_EVENTS": str(96 << 20), "ABEA_HOST_CHUNK_READS": "4096"}
    r512 = {"ABEA_HOST_CHUNK_READS": "512"}
    s16 = {"ABEA_HOST_SLOTS": "16"}
    settings = [
        ("base", "pairs", {}, True),
        ("lpt", "pairs", {"ABEA_HOST_ORDER": "lpt"}, True),
        ("fused_lpt", "fused", {"ABEA_HOST_ORDER": "lpt"}, False),
        ("pf0", "pairs", {"ABEA_HOST_FLATTEN_PREFETCH": "0"}, False),
        ("pf_nta1536", "pairs", {"ABEA_HOST_FLATTEN_HINT": "0"}, False),
        ("pf_t2_4096", "pairs", {"ABEA_HOST_FLATTEN_PREFETCH": "4096", "ABEA_HOST_FLATTEN_HINT": "2"}, False),
        ("thr12", "pairs", {"ABEA_HOST_THREADS": "12"}, False),
        ("thr16", "pairs", {"ABEA_HOST_THREADS": "16"}, False),
        ("slots4", "pairs", {"ABEA_HOST_SLOTS": "4"}, False),
        ("slots12", "pairs", {"ABEA_HOST_SLOTS": "12"}, False),
        ("slots16", "pairs", s16, True),
        ("chunk24M", "pairs", c24, False),
        ("chunk24M_slots16", "pairs", {**c24, **s16}, False),
        ("chunk96M", "pairs", c96, False),
        ("chunk96M_slots16", "pairs", {**c96, **s16}, False),
        ("rmin512", "pairs", r512, False),
        ("rmin512_slots16", "pairs", {**r512, **s16}, True),
        ("fused_base", "fused", {}, True),
        ("fused_slots16", "fused", s16, False),
        ("fused_rmin512_slots16", "fused", {**r512, **s16}, False),
        ("base_again", "pairs", {}, False),
        ("spread", "pairs", {"ABEA_HOST_NUMA": "spread"}, False),
        ("base_3", "pairs", {}, False),
        ("spread_2", "pairs", {"ABEA_HOST_NUMA": "spread"}, False),
        ("fused_spread", "fused", {"ABEA_HOST_NUMA": "spread"}, False),
        ("fused_base_2", "fused", {}, False),
        ("base_4", "pairs", {}, False),
        ("spread_3", "pairs", {"ABEA_HOST_NUMA": "spread"}, False),
    ]
    extra = os.environ.get("SWEEP_EXTRA")            # JSON list of [name, view, env, trace] appended by the caller
    if extra:
        settings += [tuple(x) for x in json.loads(extra)]
    only = set(filter(None, args.only.split(",")))
    rows = []
    swept = set()
    for _, _, env, _ in settings:
        swept |= set(env)
    swept |= set(base)
    for name, vname, env, trace in settings:
        if only and name not in only:
            continue
        for key in swept:
            os.environ.pop(key, None)
        for key, val in {**base, **env}.items():
            os.environ[key] = val
        view = views[vname]
        ctx.align_view(view)                                  # warm: slots, pinned staging, pool of this setting
        ctx.align_view(view)
        torch.cuda.synchronize()
        c0 = cpu_stat()
        acc = {}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.align_view(view)
            st = ctx.stats()
            for key in ("setup_ms", "plan_ms", "flatten_ms", "unflatten_ms", "wait_ms", "pre_ms", "fill_ms", "total_ms", "gpu_busy_ms"):
                acc[key] = acc.get(key, 0.0) + st[key]
        dt = (time.perf_counter() - t0) / args.steps
        c1 = cpu_stat()
        row = {"name": name, "view": vname, "env": env, "ms_per_step": round(dt * 1e3, 2), "mevents_per_s": round(ev_total / dt / 1e6, 1),
               **{key: round(v / args.steps, 2) for key, v in acc.items()},
               "chunks": int(st["n_sub_batches"]), "host_threads": int(st["host_threads"]),
               "throttled": {key: c1.get(key, 0) - c0.get(key, 0) for key in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec")}}
        rows.append(row)
        print(json.dumps(row), flush=True)
        if trace:
            os.environ["ABEA_HOST_TRACE"] = "1"
            sys.stderr.flush()
            saved = os.dup(2)
            fd = os.open(os.path.join(args.out, f"trace_{name}.log"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            os.dup2(fd, 2)
            try:
                ctx.align_view(view)
            finally:
                os.dup2(saved, 2)
                os.close(fd)
                os.close(saved)
                os.environ.pop("ABEA_HOST_TRACE", None)
    json.dump({"info": info, "config": args.config, "events": ev_total, "rows": rows}, open(os.path.join(args.out, "sweep.json"), "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
