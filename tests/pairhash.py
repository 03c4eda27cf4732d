"""64-bit position-dependent hash of every read's pair list ("per-config summary goldens: n_pairs per read +
64-bit hash of the pair list").  Test infrastructure: the goldens of tests/golden/config_goldens_*.npz are minted with it
from the oracle's pair lists (tests/golden/make_config_goldens.py) and the GPU tests hash the library's output the same way.

  h(read) = sum_i  mix((ref_pos_i << 32) | read_pos_i) * (2 i + 1)   mod 2^64,   i = 0-based position in the read's list
  mix(x)  = (y ^ (y >> 29)) * C2 ,  y = x * C1        (two odd 64-bit constants)

Position-dependent (a swap of two pairs changes it), vectorisable with numpy over a whole flattened batch (reduceat).
The pair (0, 0) alone contributes 0 (mix(0) = 0); the tests compare n_pairs beside the hash, so a list is never mistaken
for an empty one.
"""
import ctypes
import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_C1 = np.uint64(0x9E3779B97F4A7C15)
_C2 = np.uint64(0xBF58476D1CE4E5B9)


def hash_pair_lists(pairs, pair_ptr, n_pairs, block=1 << 24):
    """pairs: flat array viewable as int32 [*, 2] (ref_pos, read_pos); read i's list is pairs[pair_ptr[i] : +n_pairs[i]].
    Returns uint64 [n]; 0 for a read with no pairs."""
    p = pairs.view(np.int32).reshape(-1, 2)
    n = len(n_pairs)
    out = np.zeros(n, dtype=np.uint64)
    n_pairs = np.asarray(n_pairs, dtype=np.int64)
    pair_ptr = np.asarray(pair_ptr, dtype=np.int64)
    i = 0
    with np.errstate(over="ignore"):
        while i < n:
            # a run of reads whose lists together are about `block` pairs (bounded temporaries on the 21 GB full-size lists)
            j, tot = i, 0
            while j < n and (tot == 0 or tot + n_pairs[j] <= block):
                tot += int(n_pairs[j])
                j += 1
            m = n_pairs[i:j]
            nz = np.nonzero(m > 0)[0]
            if len(nz):
                mm = m[nz]
                starts = np.concatenate([[0], np.cumsum(mm)[:-1]])
                src = np.repeat(pair_ptr[i:j][nz] - starts, mm) + np.arange(int(mm.sum()), dtype=np.int64)
                seg = p[src]
                x = (seg[:, 0].astype(np.uint32).astype(np.uint64) << np.uint64(32)) | seg[:, 1].astype(np.uint32).astype(np.uint64)
                y = x * _C1
                y = (y ^ (y >> np.uint64(29))) * _C2
                pos = np.arange(int(mm.sum()), dtype=np.int64) - np.repeat(starts, mm)
                y = y * (pos.astype(np.uint64) * np.uint64(2) + np.uint64(1))
                out[i + nz] = np.add.reduceat(y, starts)
            i = j
    return out


_FAST = None


def _fast_lib():
    """tests/pairhash.c compiled into a temp directory (None if no compiler)."""
    global _FAST
    if _FAST is None:
        _FAST = False
        try:
            src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pairhash.c")
            so = os.path.join(tempfile.mkdtemp(prefix="pairhash_"), "libpairhash.so")
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
            lib = ctypes.CDLL(so)
            lib.pairhash_range.restype = None
            lib.pairhash_range.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
            _FAST = lib
        except Exception:
            _FAST = False
    return _FAST or None


def hash_pair_lists_fast(pairs, pair_ptr, n_pairs, threads=None):
    """The same hash through tests/pairhash.c on `threads` threads (ctypes releases the GIL); numpy when gcc is missing."""
    lib = _fast_lib()
    if lib is None:
        return hash_pair_lists(pairs, pair_ptr, n_pairs)
    p = np.ascontiguousarray(pairs).view(np.int32)
    ptr = np.ascontiguousarray(pair_ptr, dtype=np.int64)
    npr = np.ascontiguousarray(n_pairs, dtype=np.int32)
    n = len(npr)
    out = np.zeros(n, dtype=np.uint64)
    threads = threads or max(1, min(16, len(os.sched_getaffinity(0))))
    # equal shares of PAIRS, not of reads
    cum = np.concatenate([[0], np.cumsum(np.maximum(npr, 0).astype(np.int64))])
    cuts = np.unique(np.searchsorted(cum, np.linspace(0, cum[-1], threads + 1)[1:-1]))
    bounds = [0] + [int(c) for c in cuts if 0 < c < n] + [n]

    def run(k):
        lib.pairhash_range(p.ctypes.data, ptr.ctypes.data, npr.ctypes.data, bounds[k], bounds[k + 1], out.ctypes.data)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(run, range(len(bounds) - 1)))
    return out
