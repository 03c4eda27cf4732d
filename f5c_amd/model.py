"""Pore-model tables (reference src/model.c): text format reader and the cached log(stdv)."""
import ctypes
import ctypes.util
import math
import numpy as np
from .types import MODEL_DT

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]


def log_stdv(stdv, flavour="logf"):
    """level_log_stdv as f5c builds it.  model.c:93,179 write `log(model[i].level_stdv)` with a float argument in a
    translation unit compiled as C++ (Makefile:6 LANGFLAG = -x c++), so <math.h>'s float overload is chosen: glibc
    logf, not (float)log((double)x).  The two differ by 1 ULP on 16 of the 4096 R9.4.1 entries; "log" keeps the
    double-then-round variant for the test that shows the kernels are exact on either table."""
    stdv = np.asarray(stdv, dtype=np.float32)
    if flavour == "logf":
        return np.array([_libm.logf(float(s)) for s in stdv], dtype=np.float32)
    return np.array([math.log(float(s)) for s in stdv], dtype=np.float32)


def _finish(mean, stdv, flavour="logf"):
    m = np.zeros(len(mean), dtype=MODEL_DT)
    m["level_mean"] = mean
    m["level_stdv"] = stdv
    m["level_log_stdv"] = log_stdv(m["level_stdv"], flavour)
    return m


def load_model_f32(path, flavour="logf"):
    """Load a [4^k, 2] float32 (level_mean, level_stdv) table; returns (k, model_t[4^k])."""
    tab = np.fromfile(path, dtype=np.float32).reshape(-1, 2)
    k = int(round(math.log(len(tab), 4)))
    assert 4 ** k == len(tab)
    return k, _finish(tab[:, 0], tab[:, 1], flavour)


def read_model_text(path):
    """Reader for the nanopolish/f5c text model format (model.c:39-128): '#k' header, then
    'kmer level_mean level_stdv ...' rows in lexicographic k-mer order."""
    k = None
    mean, stdv = [], []
    for ln in open(path):
        f = ln.split()
        if not f:
            continue
        if f[0] == "#k":
            k = int(f[1])
        if ln[0] == "#" or f[0] == "kmer":
            continue
        mean.append(np.float32(f[1]))
        stdv.append(np.float32(f[2]))
    if k is None:
        k = int(round(math.log(len(mean), 4)))
    if len(mean) != 4 ** k:
        raise ValueError(f"{path}: expected {4**k} k-mers, found {len(mean)}")
    return k, _finish(np.array(mean, dtype=np.float32), np.array(stdv, dtype=np.float32))


def synthetic_model(k, seed=9):
    """Synthetic 4^k-entry table for configs with no table in the reference mount (R10 9-mer):
    mean ~ N(90, 12^2) clamped [50,140], stdv ~ U[1.5, 4.0]."""
    rng = np.random.default_rng(seed)
    n = 4 ** k
    mean = np.clip(rng.normal(90.0, 12.0, n), 50.0, 140.0).astype(np.float32)
    stdv = rng.uniform(1.5, 4.0, n).astype(np.float32)
    return _finish(mean, stdv)
