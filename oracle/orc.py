"""ctypes binding of oracle/libabea_oracle.so — TEST INFRASTRUCTURE ONLY.

The product package (f5c_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EVENT_DT = np.dtype([("start", "<u8"), ("length", "<f4"), ("mean", "<f4"), ("stdv", "<f4")], align=True)
MODEL_DT = np.dtype([("level_mean", "<f4"), ("level_stdv", "<f4"), ("level_log_stdv", "<f4")])
PAIR_DT = np.dtype([("ref_pos", "<i4"), ("read_pos", "<i4")])
SCAL_DT = np.dtype([("scale", "<f4"), ("shift", "<f4"), ("var", "<f4"), ("log_var", "<f4")])
DIAG_DT = np.dtype([("sum_emission", "<f8"), ("n_aligned", "<i4"), ("best_event", "<i4"),
                    ("max_score", "<f4"), ("max_gap", "<i4"), ("spanned", "<i4"), ("oob", "<i4"),
                    ("pad", "<i4")], align=True)
IDXPAIR_DT = np.dtype([("start", "<i4"), ("stop", "<i4")])
assert EVENT_DT.itemsize == 24 and MODEL_DT.itemsize == 12 and PAIR_DT.itemsize == 8
assert SCAL_DT.itemsize == 16 and DIAG_DT.itemsize == 40


def build(force=False):
    so = os.path.join(_HERE, "libabea_oracle.so")
    src = os.path.join(_HERE, "abea_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp = C.c_void_p
        L.orc_kmer_rank.restype = C.c_uint32
        L.orc_kmer_rank.argtypes = [C.c_char_p, C.c_uint32]
        L.orc_estimate_scalings.restype = C.c_float * 4
        L.orc_align.restype = C.c_int32
        L.orc_align.argtypes = [vp, C.c_char_p, C.c_int32, vp, C.c_size_t, vp, C.c_uint32,
                                C.c_float, C.c_float, vp]
        L.orc_align_single.restype = C.c_int32
        L.orc_align_single.argtypes = [vp, C.c_char_p, C.c_int32, vp, C.c_size_t, C.c_int64, vp,
                                       C.c_uint32, C.c_float, C.c_float, vp]
        L.orc_align_batch.restype = None
        L.orc_align_batch.argtypes = [C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32,
                                      vp, vp, vp, vp, C.c_int32]
        L.orc_scaling_single.restype = C.c_int32
        L.orc_scaling_single.argtypes = [vp, C.c_int32, C.c_char_p, C.c_int32, vp, C.c_size_t, vp,
                                         C.c_uint32, vp, vp, vp, vp]
        _LIB = L
    return _LIB


class _Scal(C.Structure):
    _fields_ = [("scale", C.c_float), ("shift", C.c_float), ("var", C.c_float), ("log_var", C.c_float)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def kmer_rank(s: bytes, k: int) -> int:
    return lib().orc_kmer_rank(s, k)


def estimate_scalings(seq: bytes, model, k, events):
    L = lib()
    L.orc_estimate_scalings.restype = _Scal
    L.orc_estimate_scalings.argtypes = [C.c_char_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    r = L.orc_estimate_scalings(seq, len(seq), _p(model), k, _p(events), len(events))
    return float(r.scale), float(r.shift)


def align(seq: bytes, events, model, k, scale, shift, nsample=1):
    """One read through align_single (+ guards). Returns (pairs[n], diag record)."""
    assert events.dtype == EVENT_DT and model.dtype == MODEL_DT
    out = np.zeros(len(events) + len(seq), dtype=PAIR_DT)
    diag = np.zeros(1, dtype=DIAG_DT)
    n = lib().orc_align_single(_p(out), seq, len(seq), _p(events), len(events), nsample, _p(model), k,
                               np.float32(scale), np.float32(shift), _p(diag))
    return out[:n].copy(), diag[0]


def align_batch(batch, model, k, n_threads=1, want_diag=True):
    """batch: dict with the flattened arrays (see f5c_amd.synth). Returns (pairs, n_pairs, diags)."""
    n = len(batch["read_len"])
    pairs = np.zeros(int(batch["pair_cap"]), dtype=PAIR_DT)
    n_pairs = np.zeros(n, dtype=np.int32)
    diags = np.zeros(n, dtype=DIAG_DT) if want_diag else None
    lib().orc_align_batch(n, _p(batch["reads"]), _p(batch["read_ptr"]), _p(batch["read_len"]),
                          _p(batch["events"]), _p(batch["event_ptr"]), _p(batch["n_events"]),
                          _p(batch["scalings"]), _p(model), k, _p(pairs), _p(batch["pair_ptr"]),
                          _p(n_pairs), _p(diags) if want_diag else None, n_threads)
    return pairs, n_pairs, diags


def getevents(raw_adc, offset, rng, digitisation, rna=False):
    """int16 ADC samples + channel scaling -> (event table, pA signal) as event_single does (f5c.c:682-710).  The table is
    in detection order for RNA too (getevents, events.c:562-582); reverse_events() is event_single's last step."""
    L = lib()
    L.orc_getevents_rna.restype = C.c_size_t
    L.orc_getevents_rna.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_raw_to_pa.restype = None
    L.orc_raw_to_pa.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float]
    pa = np.ascontiguousarray(raw_adc, dtype=np.float32).copy()
    L.orc_raw_to_pa(_p(pa), len(pa), np.float32(offset), np.float32(rng), np.float32(digitisation))
    ev = np.zeros(len(pa), dtype=EVENT_DT)
    n = L.orc_getevents_rna(len(pa), _p(pa), _p(ev), 1 if rna else 0)
    return ev[:n].copy(), pa


def reverse_events(events):
    """f5c.c:711-719 (RNA): the table reversed to 3'->5', in place on a copy."""
    ev = np.ascontiguousarray(events).copy()
    L = lib()
    L.orc_reverse_events.restype = None
    L.orc_reverse_events.argtypes = [C.c_void_p, C.c_size_t]
    L.orc_reverse_events(_p(ev), len(ev))
    return ev


def rsq_format(fmt, read_id, seq_len, k, base_to_event_map, events, n_samples, scale, shift, rna=False):
    """output_db_rsq() for one read (resquiggle.c:319-449): TSV (fmt 0) or PAF (fmt 1) text; None where the reference
    would assert / exit."""
    L = lib()
    L.orc_rsq_format.restype = C.c_long
    L.orc_rsq_format.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_char_p, C.c_int32, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_int]
    m = np.ascontiguousarray(np.asarray(base_to_event_map).view(np.int32).reshape(-1, 2).copy())
    ev = np.ascontiguousarray(events)
    buf = C.create_string_buffer(seq_len * 64 + 4096)
    rid = read_id.encode() if isinstance(read_id, str) else read_id
    n = L.orc_rsq_format(buf, len(buf), fmt, rid, seq_len, k, m.ctypes.data, ev.ctypes.data, n_samples, scale, shift,
                         1 if rna else 0)
    return None if n < 0 else buf.value.decode()


def cpg_kmer_rank(kmer: bytes, k: int) -> int:
    L = lib()
    L.orc_cpg_kmer_rank.restype = C.c_uint32
    L.orc_cpg_kmer_rank.argtypes = [C.c_char_p, C.c_uint32]
    return int(L.orc_cpg_kmer_rank(kmer, k))


def flogsum_table():
    """The 16000-entry float table of p7_FLogsum (logsum.h:33-48) as the oracle (glibc log/exp) builds it."""
    L = lib()
    L.orc_flogsum_table.restype = C.POINTER(C.c_float)
    return np.ctypeslib.as_array(L.orc_flogsum_table(), shape=(16000,)).copy()


class _Scal(C.Structure):
    _fields_ = [("scale", C.c_float), ("shift", C.c_float), ("var", C.c_float), ("log_var", C.c_float)]


def profile_hmm_score(m_seq: bytes, m_rc_seq: bytes, events, scaling, cpgmodel, k, e_start, e_stop, stride, rc,
                      events_per_base, hmm_flags=3):
    """profile_hmm_score (hmm.c:692 -> :628): forward log-likelihood of events e_start..e_stop against m_seq.
    scaling = (scale, shift, var, log_var); pinned to single_read/meth.exp (tests/test_hmm_pin.py)."""
    L = lib()
    L.orc_profile_hmm_score.restype = C.c_float
    L.orc_profile_hmm_score.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, _Scal, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_int8, C.c_uint8, C.c_double, C.c_uint32]
    ev = np.ascontiguousarray(events)
    mod = np.ascontiguousarray(cpgmodel)
    return float(L.orc_profile_hmm_score(m_seq, m_rc_seq, ev.ctypes.data, _Scal(*[float(x) for x in scaling]),
                                         mod.ctypes.data, k, e_start, e_stop, stride, 1 if rc else 0,
                                         float(events_per_base), hmm_flags))


def malloc_tuning(on: bool):
    lib().orc_malloc_tuning(1 if on else 0)


def scaling_single(pairs, seq: bytes, events, model, k, scale, shift):
    sc = np.zeros(1, dtype=SCAL_DT)
    sc["scale"] = scale
    sc["shift"] = shift
    K = len(seq) - k + 1
    bmap = np.zeros(K, dtype=IDXPAIR_DT)
    epb = np.zeros(1, dtype=np.float64)
    flag = np.zeros(1, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs)
    n = lib().orc_scaling_single(_p(pairs), len(pairs), seq, len(seq), _p(events), len(events), _p(model), k,
                                 _p(sc), _p(bmap), _p(epb), _p(flag))
    return dict(n_alignment=n, scalings=sc[0], base_to_event_map=bmap, events_per_base=float(epb[0]),
                flag=int(flag[0]))
