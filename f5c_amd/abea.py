"""Host-side mirror of the f5c `align_db` GPU branch over the C ABI of libabea_hip.so.

`AbeaContext` ~ init_cuda/free_cuda (src/f5c.cu:23,204); `align_db_host` ~ align_cuda on a db_t
(src/f5c.cu:647); `align_db_device` is the same computation on a device-resident flattened batch
(torch tensors are used only as HBM buffers).  There is no CPU fallback: if the library or the GPU
is missing these raise.
"""
import ctypes as C
import os
import numpy as np
from .types import EVENT_DT, MODEL_DT, PAIR_DT, SCAL_DT, DIAG_DT

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABEA_LIB_PATH", os.path.join(_HERE, "libabea_hip.so"))   # override only for kernel A/B experiments

EXPORTS = ["abea_init", "abea_init_multi", "abea_device_count", "abea_free", "abea_last_error", "abea_align_batch_host",
           "abea_align_batch_device", "abea_detect_events_device", "abea_get_stats", "abea_get_device_stats",
           "abea_device_info", "abea_selftest", "abea_rsq_format", "abea_lpt_split", "abea_hmm_score_batch_host", "abea_expand_walk_codes", "abea_expand_walk_codes_to_map",
           "abea_host_plan_chunks", "abea_host_plan_threads", "abea_set_inflight", "abea_align_batch_host_submit",
           "abea_align_batch_host_wait", "abea_events_batch_host", "abea_process_batch_host", "abea_rsq_format_batch",
           "abea_hmm_score_batch_device", "abea_expand_kmer_counts_to_map", "abea_flatten_event_means", "abea_link_probe"]
SHIM_EXPORTS = ["abea_f5c_init", "abea_f5c_align", "abea_f5c_align_scale", "abea_f5c_free", "abea_f5c_align_submit",
                "abea_f5c_align_wait", "abea_f5c_event_db", "abea_f5c_process"]      # include/abea_f5c_shim.h


class AbeaError(RuntimeError):
    pass


def rsq_format(fmt, read_id, seq_len, kmer_size, base_to_event_map, events, n_samples, scale, shift, rna=False):
    """Row N3: the text `f5c resquiggle` prints for one read (abea_rsq_format: TSV fmt=0, PAF fmt=1).  Host-only.
    base_to_event_map: IDXPAIR-like int32 [K,2] array (a copy is passed, so the caller's array is not reversed)."""
    lib = load_library()
    lib.abea_rsq_format.restype = C.c_int64
    lib.abea_rsq_format.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_char_p, C.c_int32, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int]
    m = np.ascontiguousarray(np.asarray(base_to_event_map).view(np.int32).reshape(-1, 2).copy())
    ev = np.ascontiguousarray(events)
    rid = read_id.encode() if isinstance(read_id, str) else read_id
    buf = C.create_string_buffer(len(m) * (len(rid) + 64) + 2 * len(rid) + 512)     # bounds both formats
    got = lib.abea_rsq_format(buf, len(buf), fmt, rid, seq_len, kmer_size, m.ctypes.data, ev.ctypes.data, n_samples,
                              scale, shift, 1 if rna else 0)
    if got < 0:
        raise AbeaError(f"abea_rsq_format: inconsistent base_to_event_map ({got})")
    assert got < len(buf)
    return buf.value.decode()


class _Cfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("kmer_size", C.c_uint32), ("model", C.c_void_p),
                ("mem_frac", C.c_float), ("max_arena_bytes", C.c_uint64), ("verbosity", C.c_int32),
                ("reserved", C.c_int32)]


class _HostBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("read", C.c_void_p), ("read_len", C.c_void_p),
                ("events", C.c_void_p), ("n_events", C.c_void_p), ("scalings", C.c_void_p),
                ("n_samples", C.c_void_p), ("pairs", C.c_void_p), ("n_pairs", C.c_void_p),
                ("diag", C.c_void_p),
                ("base_to_event_map", C.c_void_p), ("scalings_out", C.c_void_p), ("events_per_base", C.c_void_p),
                ("read_stat_flag", C.c_void_p), ("n_event_alignment", C.c_void_p),
                ("min_num_events_to_rescale", C.c_int32), ("flags", C.c_int32)]


class _DevBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("read_ptr", C.c_void_p), ("read_len", C.c_void_p),
                ("event_ptr", C.c_void_p), ("n_events", C.c_void_p), ("pair_ptr", C.c_void_p),
                ("scalings", C.c_void_p), ("reads", C.c_void_p), ("events", C.c_void_p),
                ("pairs", C.c_void_p), ("n_pairs", C.c_void_p), ("diag", C.c_void_p),
                ("kmer_ptr", C.c_void_p), ("base_to_event_map", C.c_void_p), ("scalings_io", C.c_void_p),
                ("events_per_base", C.c_void_p), ("read_stat_flag", C.c_void_p), ("n_event_alignment", C.c_void_p),
                ("min_num_events_to_rescale", C.c_int32), ("reserved", C.c_int32)]


class _SigBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("sig_ptr", C.c_void_p), ("n_samples", C.c_void_p), ("scaling", C.c_void_p),
                ("event_ptr", C.c_void_p), ("event_cap", C.c_void_p), ("read_ptr", C.c_void_p), ("read_len", C.c_void_p),
                ("signal", C.c_void_p), ("reads", C.c_void_p), ("events", C.c_void_p), ("n_events", C.c_void_p),
                ("scalings", C.c_void_p), ("rna", C.c_int32), ("reserved", C.c_int32)]


class _EventsHostBatch(C.Structure):
    """abea_events_host_batch (include/abea.h)"""
    _fields_ = [("n_reads", C.c_int32), ("rawptr", C.c_void_p), ("n_samples", C.c_void_p), ("offset", C.c_void_p),
                ("range", C.c_void_p), ("digitisation", C.c_void_p), ("read", C.c_void_p), ("read_len", C.c_void_p),
                ("rna", C.c_int32), ("signal_to_pa_in_place", C.c_int32), ("events", C.c_void_p), ("n_events", C.c_void_p),
                ("scalings", C.c_void_p)]


class _ProcessBatch(C.Structure):
    """abea_process_batch (include/abea.h)"""
    _fields_ = [("n_reads", C.c_int32), ("rawptr", C.c_void_p), ("n_samples", C.c_void_p), ("offset", C.c_void_p),
                ("range", C.c_void_p), ("digitisation", C.c_void_p), ("read", C.c_void_p), ("read_len", C.c_void_p),
                ("rna", C.c_int32), ("signal_to_pa_in_place", C.c_int32), ("events", C.c_void_p), ("n_events", C.c_void_p),
                ("scalings", C.c_void_p), ("scalings_estimated", C.c_void_p), ("pairs", C.c_void_p), ("n_pairs", C.c_void_p),
                ("diag", C.c_void_p), ("base_to_event_map", C.c_void_p), ("events_per_base", C.c_void_p),
                ("read_stat_flag", C.c_void_p), ("n_event_alignment", C.c_void_p),
                ("min_num_events_to_rescale", C.c_int32), ("reserved", C.c_int32)]


class _Scal(C.Structure):
    _fields_ = [("scale", C.c_float), ("shift", C.c_float), ("var", C.c_float), ("log_var", C.c_float)]


class HmmJob(C.Structure):
    """abea_hmm_job_t: one profile_hmm_score() call (hmm.c:689-735)."""
    _fields_ = [("m_seq", C.c_char_p), ("m_rc_seq", C.c_char_p), ("events", C.c_void_p), ("scaling", _Scal),
                ("event_start_idx", C.c_uint32), ("event_stop_idx", C.c_uint32), ("event_stride", C.c_int8),
                ("rc", C.c_uint8), ("pad", C.c_uint16), ("hmm_flags", C.c_uint32), ("events_per_base", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("pre_ms", C.c_double), ("fill_ms", C.c_double), ("trace_ms", C.c_double),
                ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("host_ms", C.c_double), ("event_ms", C.c_double),
                ("hmm_ms", C.c_double),
                ("total_ms", C.c_double),
                ("n_reads_gpu", C.c_int64), ("n_reads_skipped", C.c_int64), ("n_sub_batches", C.c_int64),
                ("sum_events", C.c_int64), ("sum_bands", C.c_int64), ("sum_pairs", C.c_int64),
                ("fill_launches", C.c_int64),
                ("arena_bytes", C.c_uint64), ("bytes_ref", C.c_uint64), ("bytes_min", C.c_uint64),
                ("bytes_moved", C.c_uint64),
                ("flatten_ms", C.c_double), ("unflatten_ms", C.c_double), ("wait_ms", C.c_double),
                ("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("n_devices", C.c_int32),
                ("host_threads", C.c_int32), ("plan_ms", C.c_double), ("setup_ms", C.c_double),
                ("gpu_busy_ms", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_LIB = None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load_library():
    """Load libabea_hip.so (built by __graft_entry__.build()). Raises if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AbeaError(f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
        # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): let it load first so that both this
        # library and torch share ONE HIP runtime in the process (the other order leaves torch without a device)
        try:
            import torch
            torch.cuda.is_available()
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.abea_init.restype = C.c_int
        L.abea_init.argtypes = [C.POINTER(C.c_void_p), C.POINTER(_Cfg)]
        L.abea_init_multi.restype = C.c_int
        L.abea_init_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(_Cfg), C.c_void_p, C.c_int32]
        L.abea_device_count.restype = C.c_int32
        L.abea_device_count.argtypes = [C.c_void_p]
        L.abea_get_device_stats.restype = C.c_int
        L.abea_get_device_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Stats)]
        L.abea_lpt_split.restype = C.c_int
        L.abea_lpt_split.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.abea_free.restype = None
        L.abea_free.argtypes = [C.c_void_p]
        L.abea_last_error.restype = C.c_char_p
        L.abea_align_batch_host.restype = C.c_int
        L.abea_align_batch_host.argtypes = [C.c_void_p, C.POINTER(_HostBatch)]
        L.abea_align_batch_device.restype = C.c_int
        L.abea_align_batch_device.argtypes = [C.c_void_p, C.POINTER(_DevBatch)]
        L.abea_detect_events_device.restype = C.c_int
        L.abea_detect_events_device.argtypes = [C.c_void_p, C.POINTER(_SigBatch)]
        L.abea_get_stats.restype = C.c_int
        L.abea_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.abea_device_info.restype = C.c_int
        L.abea_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_uint64)]
        L.abea_hmm_score_batch_host.restype = C.c_int
        L.abea_hmm_score_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.abea_selftest.restype = C.c_int
        L.abea_selftest.argtypes = [C.c_void_p]
        L.abea_set_inflight.restype = C.c_int
        L.abea_set_inflight.argtypes = [C.c_void_p, C.c_int32]
        L.abea_align_batch_host_submit.restype = C.c_int
        L.abea_align_batch_host_submit.argtypes = [C.c_void_p, C.POINTER(_HostBatch), C.POINTER(C.c_int32)]
        L.abea_align_batch_host_wait.restype = C.c_int
        L.abea_align_batch_host_wait.argtypes = [C.c_void_p, C.c_int32]
        L.abea_host_plan_threads.restype = C.c_int
        L.abea_host_plan_threads.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_size_t]
        _LIB = L
    return _LIB


def plan_host_threads(usable_cpus, n_devices, device_numa_node=None, node_cpulists=(), allowed=None):
    """abea_host_plan_threads (host-only): worker threads per device context and the cpulist each pool binds to."""
    lib = load_library()
    thr = np.zeros(n_devices, dtype=np.int32)
    cap = 1024
    buf = C.create_string_buffer(cap * n_devices)
    nodes = np.ascontiguousarray(device_numa_node, dtype=np.int32) if device_numa_node is not None else None
    lists = (C.c_char_p * max(1, len(node_cpulists)))(*[x.encode() for x in node_cpulists])
    rc = lib.abea_host_plan_threads(usable_cpus, allowed.encode() if allowed else None, n_devices,
                                    _p(nodes) if nodes is not None else None, len(node_cpulists),
                                    C.cast(lists, C.c_void_p), _p(thr), C.cast(buf, C.c_void_p), cap)
    if rc != 0:
        raise AbeaError(f"abea_host_plan_threads failed ({rc}): {lib.abea_last_error().decode()}")
    return thr, [buf.raw[d * cap:(d + 1) * cap].split(b"\0", 1)[0].decode() for d in range(n_devices)]


class AbeaContext:
    """Device context: model copy + scratch arena + stream (init_cuda / free_cuda)."""

    def __init__(self, model, kmer_size, device_id=0, mem_frac=0.9, max_arena_bytes=0, verbosity=0, device_ids=None):
        """device_ids: list of devices for one multi-GPU context (abea_init_multi); overrides device_id."""
        assert model.dtype == MODEL_DT and len(model) == 4 ** kmer_size
        self._lib = load_library()
        self._model = np.ascontiguousarray(model)
        self.kmer_size = kmer_size
        cfg = _Cfg(device_id, kmer_size, self._model.ctypes.data, mem_frac, max_arena_bytes, verbosity, 0)
        h = C.c_void_p()
        if device_ids is not None:
            ids = np.ascontiguousarray(device_ids, dtype=np.int32)
            rc = self._lib.abea_init_multi(C.byref(h), C.byref(cfg), _p(ids), len(ids))
        else:
            rc = self._lib.abea_init(C.byref(h), C.byref(cfg))
        if rc != 0:
            raise AbeaError(f"abea_init failed ({rc}): {self._lib.abea_last_error().decode()}")
        self._h = h

    def device_count(self):
        return int(self._lib.abea_device_count(self._h))

    def device_stats(self, d):
        s = Stats()
        self._chk(self._lib.abea_get_device_stats(self._h, d, C.byref(s)), "abea_get_device_stats")
        return s.asdict()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.abea_free(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            raise AbeaError(f"{what} failed ({rc}): {self._lib.abea_last_error().decode()}")

    def selftest(self):
        self._chk(self._lib.abea_selftest(self._h), "abea_selftest")

    def device_info(self):
        arch = C.create_string_buffer(64)
        ncu = C.c_int32()
        arena = C.c_uint64()
        self._chk(self._lib.abea_device_info(self._h, arch, 64, C.byref(ncu), C.byref(arena)), "abea_device_info")
        return dict(arch=arch.value.decode(), n_cu=ncu.value, arena_bytes=arena.value)

    def link_probe(self, nbytes=0, reps=0):
        """abea_link_probe: GB/s of the host<->device link with pinned memory, one direction at a time and both at once."""
        out = (C.c_double * 10)()
        self._lib.abea_link_probe.restype = C.c_int
        self._lib.abea_link_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(C.c_double), C.c_int32]
        self._chk(self._lib.abea_link_probe(self._h, nbytes, reps, out, 10), "abea_link_probe")
        keys = ("h2d_copy", "d2h_copy", "d2h_kernel", "both_h2d_copy", "both_d2h_kernel", "both_copy_h2d", "both_copy_d2h",
                "h2d_kernel", "both_h2d_kernel", "both_d2h_copy")
        return {k_: round(float(v), 2) for k_, v in zip(keys, out)}

    def stats(self):
        s = Stats()
        self._chk(self._lib.abea_get_stats(self._h, C.byref(s)), "abea_get_stats")
        return s.asdict()

    # ---- db_t view: arrays of per-read pointers (align_cuda, f5c.cu:647) ----
    def align_db_host(self, seqs, events_list, scalings, n_samples=None, want_diag=True, scaling=False, want_pairs=True,
                      read_stat_flag=None):
        """seqs: list[bytes]; events_list: list[EVENT_DT array]; scalings: SCAL_DT array.
        Returns (list of PAIR_DT arrays, n_pairs int32[n], diag DIAG_DT[n] or None); with scaling=True a 4th item:
        dict(base_to_event_map=list of int32 [K,2] arrays, scalings, events_per_base, read_stat_flag, n_event_alignment)
        = what scaling_single leaves in db_t (f5c.c:736-807)."""
        n = len(seqs)
        seq_bufs = [C.create_string_buffer(s, len(s) + 1) for s in seqs]
        evs = [np.ascontiguousarray(e, dtype=EVENT_DT) for e in events_list]
        outs = [np.zeros(len(e) + len(s), dtype=PAIR_DT) for e, s in zip(evs, seqs)] if want_pairs else None
        read_pp = (C.c_void_p * n)(*[C.addressof(b) for b in seq_bufs])
        ev_pp = (C.c_void_p * n)(*[e.ctypes.data if len(e) else None for e in evs])
        out_pp = (C.c_void_p * n)(*[o.ctypes.data if len(o) else None for o in outs]) if want_pairs else None
        read_len = np.array([len(s) for s in seqs], dtype=np.int32)
        n_events = np.array([len(e) for e in evs], dtype=np.uint64)
        sc = np.ascontiguousarray(scalings, dtype=SCAL_DT)
        n_pairs = np.zeros(n, dtype=np.int32)
        diag = np.zeros(n, dtype=DIAG_DT) if want_diag else None
        ns = np.ascontiguousarray(n_samples, dtype=np.int64) if n_samples is not None else None
        extra = [None] * 5 + [0, 0]
        if scaling:
            K = np.maximum(read_len.astype(np.int64) - self.kmer_size + 1, 0)
            maps = [np.full((int(kk), 2), -1, dtype=np.int32) for kk in K]
            map_pp = (C.c_void_p * n)(*[m.ctypes.data if len(m) else None for m in maps])
            sc_out = sc.copy()
            epb = np.zeros(n, dtype=np.float64)
            flag = (np.zeros(n, dtype=np.int32) if read_stat_flag is None
                    else np.ascontiguousarray(read_stat_flag, dtype=np.int32).copy())
            nal = np.zeros(n, dtype=np.int32)
            extra = [C.cast(map_pp, C.c_void_p), _p(sc_out), _p(epb), _p(flag), _p(nal), 0, 0]
        hb = _HostBatch(n, C.cast(read_pp, C.c_void_p), _p(read_len), C.cast(ev_pp, C.c_void_p), _p(n_events),
                        _p(sc), _p(ns) if ns is not None else None,
                        C.cast(out_pp, C.c_void_p) if want_pairs else None, _p(n_pairs),
                        _p(diag) if want_diag else None, *extra)
        self._chk(self._lib.abea_align_batch_host(self._h, C.byref(hb)), "abea_align_batch_host")
        plist = [o[:k] for o, k in zip(outs, n_pairs)] if want_pairs else None
        if scaling:
            return plist, n_pairs, diag, dict(base_to_event_map=maps, scalings=sc_out, events_per_base=epb,
                                              read_stat_flag=flag, n_event_alignment=nal)
        return plist, n_pairs, diag

    def align_flat_host(self, batch, want_diag=True, **kw):
        """Convenience: run a flattened numpy batch (f5c_amd.synth layout) through the host entry point."""
        n = len(batch["read_len"])
        seqs, evs = [], []
        for i in range(n):
            s = int(batch["read_ptr"][i]); L = int(batch["read_len"][i])
            seqs.append(batch["reads"][s:s + L].tobytes())
            s = int(batch["event_ptr"][i]); E = int(batch["n_events"][i])
            evs.append(batch["events"][s:s + E])
        return self.align_db_host(seqs, evs, batch["scalings"], want_diag=want_diag, **kw)

    def host_view(self, batch, want_diag=False, scaling=False, want_pairs=True):
        """Per-read pointer arrays over a flattened numpy batch (what a db_t holds), built once so that repeated
        align_view() calls time only the library (bench.py).  Outputs live in the returned view: pairs
        PAIR_DT[pair_cap] (read i at pair_ptr[i]), n_pairs, diag, and with scaling=True b2e int32[sum K, 2]
        (read i at kmer_ptr[i]), scalings_out, events_per_base, read_stat_flag, n_event_alignment."""
        n = len(batch["read_len"])
        v = dict(n=n, batch=batch)
        v["reads"] = np.ascontiguousarray(batch["reads"])
        v["events"] = np.ascontiguousarray(batch["events"])
        v["read_len"] = np.ascontiguousarray(batch["read_len"], dtype=np.int32)
        v["n_events"] = np.ascontiguousarray(batch["n_events"]).astype(np.uint64)
        v["scalings"] = np.ascontiguousarray(batch["scalings"], dtype=SCAL_DT)
        v["read_pp"] = (v["reads"].ctypes.data + batch["read_ptr"].astype(np.int64)).astype(np.uint64)
        v["ev_pp"] = (v["events"].ctypes.data + batch["event_ptr"].astype(np.int64) * EVENT_DT.itemsize).astype(np.uint64)
        v["n_pairs"] = np.zeros(n, dtype=np.int32)
        v["diag"] = np.zeros(n, dtype=DIAG_DT) if want_diag else None
        extra = [None] * 5 + [0, 0]
        out_pp = None
        if want_pairs:
            v["pairs"] = np.zeros(max(1, int(batch["pair_cap"])), dtype=PAIR_DT)
            v["out_pp"] = (v["pairs"].ctypes.data + batch["pair_ptr"].astype(np.int64) * PAIR_DT.itemsize).astype(np.uint64)
            out_pp = _p(v["out_pp"])
        if scaling:
            K = np.maximum(v["read_len"].astype(np.int64) - self.kmer_size + 1, 0)
            v["kmer_ptr"] = np.concatenate([[0], np.cumsum(K)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
            v["b2e"] = np.full((max(1, int(K.sum())), 2), -1, dtype=np.int32)
            v["map_pp"] = (v["b2e"].ctypes.data + v["kmer_ptr"] * 8).astype(np.uint64)
            v["scalings_out"] = v["scalings"].copy()
            v["events_per_base"] = np.zeros(n, dtype=np.float64)
            v["read_stat_flag"] = np.zeros(n, dtype=np.int32)
            v["n_event_alignment"] = np.zeros(n, dtype=np.int32)
            extra = [_p(v["map_pp"]), _p(v["scalings_out"]), _p(v["events_per_base"]), _p(v["read_stat_flag"]),
                     _p(v["n_event_alignment"]), 0, 0]
        v["hb"] = _HostBatch(n, _p(v["read_pp"]), _p(v["read_len"]), _p(v["ev_pp"]), _p(v["n_events"]),
                             _p(v["scalings"]), None, out_pp, _p(v["n_pairs"]),
                             _p(v["diag"]) if want_diag else None, *extra)
        return v

    def align_view(self, view):
        self._chk(self._lib.abea_align_batch_host(self._h, C.byref(view["hb"])), "abea_align_batch_host")

    # ---- several host batches in flight (abea_align_batch_host_submit / _wait) ----
    def set_inflight(self, n_lanes):
        self._chk(self._lib.abea_set_inflight(self._h, n_lanes), "abea_set_inflight")

    def submit_view(self, view):
        """Start a host batch (a host_view) on a free lane; returns the ticket.  The view's arrays are the caller-owned
        inputs and outputs: keep the view alive and untouched until wait()."""
        t = C.c_int32(-1)
        self._chk(self._lib.abea_align_batch_host_submit(self._h, C.byref(view["hb"]), C.byref(t)),
                  "abea_align_batch_host_submit")
        return t.value

    def wait(self, ticket):
        self._chk(self._lib.abea_align_batch_host_wait(self._h, ticket), "abea_align_batch_host_wait")

    # ---- rows N2 / N3 on host buffers: event_db and event_db -> align_db -> scaling_db (abea_process.cpp) ----
    def signal_view(self, signal_f32, sig_ptr, n_samples, scaling, batch=None, want_pairs=False, rna=False, to_pa=False):
        """Per-read pointer arrays over FLOAT ADC signals (signal_t.rawptr, f5c.h:276-286) held in one flat float32 array
        (read i at sig_ptr[i], n_samples[i] samples; scaling float32 [n,3] = offset, range, digitisation) plus, with
        `batch`, the read sequences of a flattened synth batch.  The outputs the library malloc()s per read (event tables,
        pair lists, maps) come back as pointer arrays; take what is needed with view_events() / view_map() and release
        them with free_view()."""
        n = len(n_samples)
        v = dict(n=n, signal=signal_f32, rna=rna)
        v["n_samples"] = np.ascontiguousarray(n_samples, dtype=np.int64)
        sc = np.ascontiguousarray(scaling, dtype=np.float32).reshape(n, 3)
        v["offset"], v["range"], v["digitisation"] = (np.ascontiguousarray(sc[:, j]) for j in range(3))
        v["raw_pp"] = (signal_f32.ctypes.data + np.asarray(sig_ptr, dtype=np.int64) * 4).astype(np.uint64)
        v["ev_pp"] = np.zeros(n, dtype=np.uint64); v["n_events"] = np.zeros(n, dtype=np.uint64)
        v["scalings"] = np.zeros(n, dtype=SCAL_DT); v["scalings_estimated"] = np.zeros(n, dtype=SCAL_DT)
        read_pp = read_len = None
        if batch is not None:
            v["reads"] = np.ascontiguousarray(batch["reads"])
            v["read_len"] = np.ascontiguousarray(batch["read_len"], dtype=np.int32)
            v["read_pp"] = (v["reads"].ctypes.data + batch["read_ptr"].astype(np.int64)).astype(np.uint64)
            read_pp, read_len = _p(v["read_pp"]), _p(v["read_len"])
        v["eb"] = _EventsHostBatch(n, _p(v["raw_pp"]), _p(v["n_samples"]), _p(v["offset"]), _p(v["range"]), _p(v["digitisation"]),
                                   read_pp, read_len, 1 if rna else 0, 1 if to_pa else 0, _p(v["ev_pp"]), _p(v["n_events"]),
                                   _p(v["scalings"]) if batch is not None else None)
        if batch is not None:
            v["pairs_pp"] = np.zeros(n, dtype=np.uint64) if want_pairs else None
            v["n_pairs"] = np.zeros(n, dtype=np.int32); v["diag"] = np.zeros(n, dtype=DIAG_DT)
            v["map_pp"] = np.zeros(n, dtype=np.uint64); v["events_per_base"] = np.zeros(n, dtype=np.float64)
            v["read_stat_flag"] = np.zeros(n, dtype=np.int32); v["n_event_alignment"] = np.zeros(n, dtype=np.int32)
            v["pb"] = _ProcessBatch(n, _p(v["raw_pp"]), _p(v["n_samples"]), _p(v["offset"]), _p(v["range"]), _p(v["digitisation"]),
                                    read_pp, read_len, 1 if rna else 0, 1 if to_pa else 0, _p(v["ev_pp"]), _p(v["n_events"]),
                                    _p(v["scalings"]), _p(v["scalings_estimated"]), _p(v["pairs_pp"]) if want_pairs else None,
                                    _p(v["n_pairs"]), _p(v["diag"]), _p(v["map_pp"]), _p(v["events_per_base"]),
                                    _p(v["read_stat_flag"]), _p(v["n_event_alignment"]), 0, 0)
        return v

    def events_view(self, view):
        """abea_events_batch_host = event_db (f5c.c:682-734) on the view's signals."""
        self._lib.abea_events_batch_host.restype = C.c_int
        self._lib.abea_events_batch_host.argtypes = [C.c_void_p, C.POINTER(_EventsHostBatch)]
        self._chk(self._lib.abea_events_batch_host(self._h, C.byref(view["eb"])), "abea_events_batch_host")

    def process_view(self, view):
        """abea_process_batch_host = event_db -> align_db -> scaling_db (resquiggle.c:283-315) on the view's signals."""
        self._lib.abea_process_batch_host.restype = C.c_int
        self._lib.abea_process_batch_host.argtypes = [C.c_void_p, C.POINTER(_ProcessBatch)]
        view["read_stat_flag"][:] = 0
        self._chk(self._lib.abea_process_batch_host(self._h, C.byref(view["pb"])), "abea_process_batch_host")

    @staticmethod
    def view_events(view, i):
        """copy of read i's malloc()ed event table"""
        ne = int(view["n_events"][i])
        if ne == 0 or not view["ev_pp"][i]:
            return np.zeros(0, dtype=EVENT_DT)
        return np.ctypeslib.as_array(C.cast(int(view["ev_pp"][i]), C.POINTER(C.c_uint8)), shape=(ne * EVENT_DT.itemsize,)).view(EVENT_DT).copy()

    @staticmethod
    def view_map(view, i, n_kmers):
        """copy of read i's malloc()ed base_to_event_map as int32 [K,2], None when the library left it NULL"""
        if not view["map_pp"][i]:
            return None
        return np.ctypeslib.as_array(C.cast(int(view["map_pp"][i]), C.POINTER(C.c_int32)), shape=(n_kmers * 2,)).reshape(-1, 2).copy()

    @staticmethod
    def view_pairs(view, i):
        if view.get("pairs_pp") is None or not view["pairs_pp"][i] or view["n_pairs"][i] <= 0:
            return np.zeros(0, dtype=PAIR_DT)
        return np.ctypeslib.as_array(C.cast(int(view["pairs_pp"][i]), C.POINTER(C.c_uint8)),
                                     shape=(int(view["n_pairs"][i]) * PAIR_DT.itemsize,)).view(PAIR_DT).copy()

    @staticmethod
    def free_view(view):
        """free() every per-read buffer the library malloc()ed into the view (what free_db_tmp does in f5c)"""
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free.restype = None
        for key in ("ev_pp", "pairs_pp", "map_pp"):
            arr = view.get(key)
            if arr is None:
                continue
            for ptr in arr[arr != 0]:
                libc.free(int(ptr))
            arr[:] = 0

    # ---- row N4: profile-HMM forward scores ----
    def hmm_score_batch(self, jobs, cpgmodel, kmer_size, device_events=False):
        """jobs: list of dicts(m_seq, m_rc_seq: bytes; events: EVENT_DT array (the read's table); scaling: 4 floats
        (scale, shift, var, log_var); e_start, e_stop, stride, rc, events_per_base, flags) = the arguments of
        profile_hmm_score (hmm.c:689-703).  Returns float32[n] scores.  device_events=True: `events` is the DEVICE address
        (int) of the read's event_t table in HBM and the call is abea_hmm_score_batch_device."""
        n = len(jobs)
        arr = (HmmJob * max(1, n))()
        keep = []
        for j, jb in enumerate(jobs):
            a = arr[j]
            if device_events:
                a.events = int(jb["events"])
            else:
                ev = np.ascontiguousarray(jb["events"], dtype=EVENT_DT)
                keep.append(ev)
                a.events = ev.ctypes.data
            a.m_seq = jb["m_seq"]; a.m_rc_seq = jb["m_rc_seq"]
            a.scaling = _Scal(*[float(x) for x in jb["scaling"]])
            a.event_start_idx = int(jb["e_start"]); a.event_stop_idx = int(jb["e_stop"])
            a.event_stride = int(jb["stride"]); a.rc = int(jb["rc"]); a.hmm_flags = int(jb.get("flags", 0))
            a.events_per_base = float(jb["events_per_base"])
        m = np.ascontiguousarray(cpgmodel, dtype=MODEL_DT)
        assert len(m) == 5 ** kmer_size
        out = np.zeros(max(1, n), dtype=np.float32)
        fn = self._lib.abea_hmm_score_batch_device if device_events else self._lib.abea_hmm_score_batch_host
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
        self._chk(fn(self._h, C.cast(arr, C.c_void_p), n, _p(m), kmer_size, _p(out)),
                  "abea_hmm_score_batch_device" if device_events else "abea_hmm_score_batch_host")
        return out[:n]

    # ---- device-resident flattened batch ----
    @staticmethod
    def upload(batch, device=None):
        """Copy a flattened numpy batch into HBM (torch tensors as plain device buffers)."""
        import torch
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        n = len(batch["read_len"])
        d = dict(n_reads=n,
                 read_ptr=np.ascontiguousarray(batch["read_ptr"], dtype=np.int64),
                 read_len=np.ascontiguousarray(batch["read_len"], dtype=np.int32),
                 event_ptr=np.ascontiguousarray(batch["event_ptr"], dtype=np.int64),
                 n_events=np.ascontiguousarray(batch["n_events"], dtype=np.int32),
                 pair_ptr=np.ascontiguousarray(batch["pair_ptr"], dtype=np.int64),
                 scalings=np.ascontiguousarray(batch["scalings"], dtype=SCAL_DT))
        d["reads"] = torch.from_numpy(np.ascontiguousarray(batch["reads"]).view(np.uint8)).to(dev)
        d["events"] = torch.from_numpy(np.ascontiguousarray(batch["events"]).view(np.uint8)).to(dev)
        d["pairs"] = torch.zeros(max(1, batch["pair_cap"]) * 2, dtype=torch.int32, device=dev)
        d["n_pairs"] = torch.zeros(max(1, n), dtype=torch.int32, device=dev)
        d["diag"] = torch.zeros(max(1, n) * DIAG_DT.itemsize, dtype=torch.uint8, device=dev)
        return d

    def align_db_device(self, dbatch, want_diag=True, scaling=False):
        """Run the hot path on a batch already resident in HBM (what bench.py times). Synchronous.
        scaling=True also runs scaling_single (postalign + recalibrate_model, row N1) on the device."""
        import torch
        sc = [None] * 6 + [0, 0]
        if scaling:
            n = dbatch["n_reads"]
            K = (dbatch["read_len"].astype(np.int64) - self.kmer_size + 1).clip(min=0)
            dbatch["kmer_ptr"] = np.concatenate([[0], np.cumsum(K)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
            dev = dbatch["pairs"].device
            dbatch["b2e"] = torch.full((max(1, int(K.sum())) * 2,), -1, dtype=torch.int32, device=dev)
            dbatch["scalings_io"] = torch.from_numpy(dbatch["scalings"].copy().view(np.uint8)).to(dev)
            dbatch["events_per_base"] = torch.zeros(max(1, n), dtype=torch.float64, device=dev)
            dbatch["read_stat_flag"] = torch.zeros(max(1, n), dtype=torch.int32, device=dev)
            dbatch["n_event_alignment"] = torch.zeros(max(1, n), dtype=torch.int32, device=dev)
            sc = [_p(dbatch["kmer_ptr"]), dbatch["b2e"].data_ptr(), dbatch["scalings_io"].data_ptr(),
                  dbatch["events_per_base"].data_ptr(), dbatch["read_stat_flag"].data_ptr(),
                  dbatch["n_event_alignment"].data_ptr(), 0, 0]
        # the library launches on its own non-blocking streams: everything torch queued on the current stream for these
        # buffers (uploads, fills) must have finished before the library's kernels read or OR into them
        torch.cuda.current_stream().synchronize()
        db = _DevBatch(dbatch["n_reads"], _p(dbatch["read_ptr"]), _p(dbatch["read_len"]),
                       _p(dbatch["event_ptr"]), _p(dbatch["n_events"]), _p(dbatch["pair_ptr"]),
                       _p(dbatch["scalings"]),
                       dbatch["reads"].data_ptr(), dbatch["events"].data_ptr(), dbatch["pairs"].data_ptr(),
                       dbatch["n_pairs"].data_ptr(), dbatch["diag"].data_ptr() if want_diag else None, *sc)
        self._chk(self._lib.abea_align_batch_device(self._h, C.byref(db)), "abea_align_batch_device")

    def _detect(self, signals, scaling, seqs, cap_div, rna=False, pad_to=8):
        """Shared plumbing of the N2 entry: flatten + upload the signals, run abea_detect_events_device, keep every
        output in HBM.  Returns a dict of the device tensors and host index arrays."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        n = len(signals)
        ns = np.array([len(s) for s in signals], dtype=np.int32)
        # reads start on 16-byte boundaries by default (the fastest layout); pad_to=1 packs them back to back: any 2-byte alignment
        pad = (ns.astype(np.int64) + pad_to - 1) // pad_to * pad_to
        sig_ptr = np.concatenate([[0], np.cumsum(pad)[:-1]]).astype(np.int64)
        cap = (ns // cap_div + 16).astype(np.int32)
        ev_ptr = np.concatenate([[0], np.cumsum(cap.astype(np.int64))[:-1]]).astype(np.int64)
        flat_sig = np.zeros(int(pad.sum()), dtype=np.int16)
        for i, sg in enumerate(signals):
            flat_sig[sig_ptr[i]:sig_ptr[i] + ns[i]] = sg
        # pinned staging: a pageable source is copied in 4-MB pieces through rocclr's own staging buffer, a copy kernel each
        # (round-4 verdict: 1328 __amd_rocclr_copyBuffer launches around 4 detector calls in tools/n2_profile.py)
        d_sig = torch.from_numpy(flat_sig).pin_memory().to(dev)
        d_ev = torch.zeros(int(cap.sum()) * EVENT_DT.itemsize, dtype=torch.uint8, device=dev)
        d_ne = torch.zeros(n, dtype=torch.int32, device=dev)
        sc = np.ascontiguousarray(scaling, dtype=np.float32).reshape(n, 3)
        d_reads = d_scal = None
        rp = rl = None
        if seqs is not None:
            rl = np.array([len(s) for s in seqs], dtype=np.int32)
            rp = np.concatenate([[0], np.cumsum(rl.astype(np.int64) + 1)[:-1]]).astype(np.int64)
            flat = np.zeros(int((rl.astype(np.int64) + 1).sum()), dtype=np.uint8)
            for i, s in enumerate(seqs):
                flat[rp[i]:rp[i] + rl[i]] = np.frombuffer(s, dtype=np.uint8)
            d_reads = torch.from_numpy(flat).pin_memory().to(dev)
            d_scal = torch.zeros(n * SCAL_DT.itemsize, dtype=torch.uint8, device=dev)
        torch.cuda.current_stream().synchronize()        # see align_db_device: the library's streams do not order with torch's
        sb = _SigBatch(n, _p(sig_ptr), _p(ns), _p(sc), _p(ev_ptr), _p(cap), _p(rp) if rp is not None else None,
                       _p(rl) if rl is not None else None, d_sig.data_ptr(),
                       d_reads.data_ptr() if d_reads is not None else None, d_ev.data_ptr(), d_ne.data_ptr(),
                       d_scal.data_ptr() if d_scal is not None else None, 1 if rna else 0, 0)
        self._chk(self._lib.abea_detect_events_device(self._h, C.byref(sb)), "abea_detect_events_device")
        return dict(n=n, cap=cap, ev_ptr=ev_ptr, read_ptr=rp, read_len=rl, d_ev=d_ev, d_ne=d_ne, d_scal=d_scal,
                    d_reads=d_reads)

    @staticmethod
    def _check_event_cap(n_events, cap):
        """A read with more events than its table holds would be aligned on a cut-off table (and its method-of-moments
        scalings would mix the truncated sum with the true count): refuse instead of truncating silently."""
        over = np.nonzero(n_events > cap)[0]
        if len(over):
            raise AbeaError(f"event detection found more events than event_cap on {len(over)} read(s) (first: read "
                            f"{int(over[0])}: {int(n_events[over[0]])} > {int(cap[over[0]])}); call again with a smaller cap_div")

    def detect_events_device(self, signals, scaling, seqs=None, cap_div=4, rna=False, pad_to=8):
        """Row N2: raw ADC signals -> event tables (+ method-of-moments scalings when `seqs` is given) on the
        device. signals: list of int16 arrays; scaling: float32 [n,3] (offset, range, digitisation); rna=True selects
        the RNA detector parameters and returns the tables reversed 3'->5' as event_single does (f5c.c:711-719).
        Returns (list of EVENT_DT arrays, n_events int32[n], scalings SCAL_DT[n] or None)."""
        r = self._detect(signals, scaling, seqs, cap_div, rna, pad_to)
        ne = r["d_ne"].cpu().numpy()
        self._check_event_cap(ne, r["cap"])
        allev = r["d_ev"].cpu().numpy().view(EVENT_DT)
        evs = [allev[r["ev_ptr"][i]:r["ev_ptr"][i] + min(ne[i], r["cap"][i])] for i in range(r["n"])]
        return evs, ne, (r["d_scal"].cpu().numpy().view(SCAL_DT) if r["d_scal"] is not None else None)

    def signals_to_device_batch(self, signals, scaling, seqs, cap_div=4, rna=False):
        """Rows N2 -> hot path without the event tables leaving HBM: run event detection and return a device batch
        for align_db_device whose `events` are the detector's output buffer.  Only the per-read n_events (4 B) and
        estimated scalings (16 B) cross to the host, because abea_device_batch takes them as host arrays."""
        import torch
        r = self._detect(signals, scaling, seqs, cap_div, rna)
        n = r["n"]
        self._check_event_cap(r["d_ne"].cpu().numpy(), r["cap"])
        ne = np.minimum(r["d_ne"].cpu().numpy(), r["cap"]).astype(np.int32)
        cap_pairs = ne.astype(np.int64) + r["read_len"].astype(np.int64)
        dev = r["d_ev"].device
        d = dict(n_reads=n, read_ptr=r["read_ptr"], read_len=r["read_len"], event_ptr=r["ev_ptr"], n_events=ne,
                 pair_ptr=np.concatenate([[0], np.cumsum(cap_pairs)[:-1]]).astype(np.int64),
                 scalings=np.ascontiguousarray(r["d_scal"].cpu().numpy().view(SCAL_DT)),
                 reads=r["d_reads"], events=r["d_ev"])
        d["pairs"] = torch.zeros(max(1, int(cap_pairs.sum())) * 2, dtype=torch.int32, device=dev)
        d["n_pairs"] = torch.zeros(max(1, n), dtype=torch.int32, device=dev)
        d["diag"] = torch.zeros(max(1, n) * DIAG_DT.itemsize, dtype=torch.uint8, device=dev)
        return d

    @staticmethod
    def download_scaling(dbatch):
        """(base_to_event_map int32[ΣK,2], scalings SCAL_DT[n], events_per_base f64[n], flags i32[n], n_align i32[n])"""
        n = dbatch["n_reads"]
        return (dbatch["b2e"].cpu().numpy().reshape(-1, 2), dbatch["scalings_io"].cpu().numpy().view(SCAL_DT)[:n],
                dbatch["events_per_base"].cpu().numpy()[:n], dbatch["read_stat_flag"].cpu().numpy()[:n],
                dbatch["n_event_alignment"].cpu().numpy()[:n])

    @staticmethod
    def download(dbatch):
        """(pairs PAIR_DT[cap], n_pairs int32[n], diag DIAG_DT[n]) from a device batch."""
        n = dbatch["n_reads"]
        pairs = dbatch["pairs"].cpu().numpy().view(PAIR_DT)
        n_pairs = dbatch["n_pairs"].cpu().numpy()[:n]
        diag = dbatch["diag"].cpu().numpy().view(DIAG_DT)[:n]
        return pairs, n_pairs, diag
