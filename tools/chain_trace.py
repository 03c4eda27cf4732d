#!/usr/bin/env python3
"""The raw-signal entries (abea_events_batch_host = event_db, abea_process_batch_host = event_db -> align_db -> scaling_db) on a
10 k-read sample of configs[2], for the question "what bounds the chain": link ceilings (abea_link_probe), every form / mover of
the tables' trip down in ONE process on ONE set of signals, the per-call host split and GPU-clock busy time, and — with
ABEA_HOST_TRACE — the chunk timeline on stderr.  Two steps so that no generator pool runs under rocprofv3:
    python tools/chain_trace.py 10000 /tmp/ct                       # generates the batch (16 workers) and saves it
    [rocprofv3 --kernel-trace --stats ... --] python tools/chain_trace.py 10000 /tmp/ct [mode ...]
mode = format:mover[:trace][:dN][:bN][:uk], e.g. packed:kernel full:engine:trace packed:engine:d2 packed:kernel:b32:uk (dN = ABEA_CHAIN_DEPTH,
chunks in flight; bN = ABEA_CHAIN_COPY_BLOCKS, workgroups of a copy kernel; uk = ABEA_CHAIN_UP=kernel, the signal comes up by kernel);
default = all four without a trace."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5c_amd import abea, synth, load_model_f32

n = int(sys.argv[1]); cache = sys.argv[2] + f".{n}.npz"
modes = sys.argv[3:] or ["full:kernel", "packed:kernel", "full:engine", "packed:engine"]
k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
if not os.path.exists(cache):
    cfg = synth.CONFIGS["r9_100k_mixed"]
    b = synth.make_batch(n, model, k, seed=cfg["seed"] + 1, law=cfg["law"], workers=16)
    np.savez(cache, **b)
    print("saved", cache, int(b["n_events"].sum()), "events")
    sys.exit(0)
z = np.load(cache)
b = {key: z[key] for key in z.files}
b["pair_cap"] = int(b["pair_cap"])
t0 = time.time()
sig, sp, ns, sc = synth.make_signals_flat(b, seed=5, threads=14)
n_smp = int(ns.sum())
print(f"{n} reads, {n_smp/1e6:.1f} Msamples, signals in {time.time()-t0:.1f} s", flush=True)
ctx = abea.AbeaContext(model, k, max_arena_bytes=150 << 30)
link = ctx.link_probe()
print("link GB/s:", link, flush=True)
for blocks in (8, 16, 64, 512):
    os.environ["ABEA_CHAIN_COPY_BLOCKS"] = str(blocks)
    lk = ctx.link_probe()
    print(f"link GB/s with {blocks:3d}-workgroup copy kernels:", {k_: v for k_, v in lk.items() if "kernel" in k_ or k_ == "both_d2h_copy" or k_ == "both_h2d_copy"}, flush=True)
os.environ.pop("ABEA_CHAIN_COPY_BLOCKS", None)


def digest(v):
    """order-independent fingerprint of every table the call handed back"""
    h = 0
    for j in range(0, len(ns), max(1, len(ns) // 200)):
        e = abea.AbeaContext.view_events(v, j)
        h ^= hash((len(e), e["start"].tobytes()[-64:], e["mean"].tobytes()[:64], e["length"].tobytes()[-64:], e["stdv"].tobytes()[:64]))
    return h


ref = None
for mode in modes:
    parts = mode.split(":")
    os.environ["ABEA_CHAIN_TABLE_FORMAT"], os.environ["ABEA_CHAIN_TABLE_COPY"] = parts[0], parts[1]
    trace = "trace" in parts[2:]
    for key, var in (("d", "ABEA_CHAIN_DEPTH"), ("b", "ABEA_CHAIN_COPY_BLOCKS")):      # dN = chunks in flight, bN = workgroups of a copy kernel
        val = [x[1:] for x in parts[2:] if x.startswith(key) and x[1:].isdigit()]
        if val: os.environ[var] = val[0]
        else: os.environ.pop(var, None)
    if "uk" in parts[2:]: os.environ["ABEA_CHAIN_UP"] = "kernel"                       # uk = the signal comes up by the copy kernel
    else: os.environ.pop("ABEA_CHAIN_UP", None)
    for entry in ("events", "process"):
        v = ctx.signal_view(sig, sp, ns, sc, batch=b)
        call = ctx.events_view if entry == "events" else ctx.process_view
        call(v); ctx.free_view(v)                                  # warm: slots, pinned staging
        for rep in range(3):
            if trace and rep == 2:
                os.environ["ABEA_HOST_TRACE"] = "1"
                print(f"---- trace: {mode} {entry}", file=sys.stderr, flush=True)
            t0 = time.perf_counter(); call(v); t = time.perf_counter() - t0
            os.environ.pop("ABEA_HOST_TRACE", None)
            st = ctx.stats()
            n_ev = int(v["n_events"].sum())
            if rep == 2:
                d = digest(v)
                if entry == "events":
                    ref = d if ref is None else ref
                    assert d == ref, "tables differ between modes"
            ctx.free_view(v)
            print(f"{mode:14s} {entry:7s} rep {rep}: {t*1e3:7.1f} ms = {n_smp/t/1e9:5.2f} Gsamples/s | flatten {st['flatten_ms']:6.1f} scatter {st['unflatten_ms']:6.1f} "
                  f"wait {st['wait_ms']:6.1f} plan {st['plan_ms']:5.1f} | kernels sum: detect {st['event_ms']:6.1f} pre {st['pre_ms']:5.1f} align {st['fill_ms']:6.1f} "
                  f"| gpu busy {st['gpu_busy_ms']:6.1f} | h2d {st['h2d_bytes']/1e9:5.2f} GB ({st['h2d_bytes']/t/1e9:5.1f} GB/s) d2h {st['d2h_bytes']/1e9:5.2f} GB "
                  f"({st['d2h_bytes']/t/1e9:5.1f} GB/s) | {st['n_sub_batches']} chunks, {n_ev/1e6:.1f} Mevents", flush=True)
ctx.close()
