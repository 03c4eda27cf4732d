"""CPU: the C-ABI library loads and exports every symbol include/abea.h declares."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="abea.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(abea_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for n in ("abea_init", "abea_free", "abea_align_batch_host", "abea_align_batch_device", "abea_last_error"):
        assert n in names


def test_library_exports_every_declared_symbol():
    from f5c_amd import abea
    if not os.path.exists(abea.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(abea.LIB_PATH)
    for n in _declared():
        assert hasattr(lib, n), f"{n} declared in include/abea.h but not exported"
    assert sorted(abea.EXPORTS) == _declared()
    for n in _declared("abea_f5c_shim.h"):
        assert hasattr(lib, n), f"{n} declared in include/abea_f5c_shim.h but not exported"
    assert sorted(abea.SHIM_EXPORTS) == _declared("abea_f5c_shim.h")


def test_no_gpu_fails_loudly(r9):
    """Without a GPU abea_init must fail with a message, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from f5c_amd import abea
    k, model = r9
    with pytest.raises(abea.AbeaError):
        abea.AbeaContext(model, k)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "f5c_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "abea_oracle" not in src, f


def test_generated_asm_is_current():
    """abea_fill.inc / abea_walk.inc are generated; the committed copies must be what tools/gen_fill_asm.py emits."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_fill_asm.py"), "--check"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("up to date") == 2



def test_static_pmc_counters_belong_to_these_kernel_sources():
    """bench.py quotes HBM traffic / VALU-busy from profiles/pmc_traffic.json only when its code_sha256 — sha256 over abea_fill.inc +
    abea_walk.inc + abea_kernels.hip (up to its event-detection section) at the time of the rocprofv3 passes — equals the tree's (round-4 verdict: a 3 % time window is
    not a guard).  The committed json must describe the committed kernel: re-take the passes (tools/gpu_call.sh prof100a / prof100b /
    prof10, profiles/make_pmc_traffic.py) after any change to those three files."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    sha = bench.kernel_code_sha()
    for config in ("r9_10k_8kb", "r9_100k_mixed"):
        assert t[config]["code_sha256"] == sha, config
        assert bench.pmc_entry(config) is not None and bench.pmc_traffic(config, 1000, 1) == int(t[config]["hbm_bytes_per_event"] * 1000)


def test_hw_queue_setenv_opt_out():
    """abea_init asks for 16 hardware queues by setenv before its first HIP call unless the caller set the variable — and not at all
    with ABEA_KEEP_HW_QUEUES set (round-5 advisor finding: a library writing the environment of a multi-threaded host).  Runs in
    a subprocess without a GPU: abea_init fails with ABEA_ENODEV AFTER that decision, which is all this test needs."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
os.environ["ABEA_KEEP_HW_QUEUES"] = "1"           # keeps f5c_amd/__init__.py from exporting the variable itself
os.environ.pop("GPU_MAX_HW_QUEUES", None)
from f5c_amd import abea
if sys.argv[1] == "default":
    os.environ.pop("ABEA_KEEP_HW_QUEUES")
lib = abea.load_library()
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p
model = (ctypes.c_float * (3 * 4096))()
cfg = abea._Cfg(0, 6, ctypes.addressof(model), 0.5, 1 << 20, 0, 0)
h = ctypes.c_void_p()
rc = lib.abea_init(ctypes.byref(h), ctypes.byref(cfg))
print(rc, libc.getenv(b"GPU_MAX_HW_QUEUES"))
"""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("GPU_MAX_HW_QUEUES", "ABEA_KEEP_HW_QUEUES")}
    env["HIP_VISIBLE_DEVICES"] = "-1"                  # no device even on a GPU box: the decision comes before the device check
    out = {}
    for mode in ("default", "keep"):
        r = subprocess.run([sys.executable, "-c", code % root, mode], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = r.stdout.strip().splitlines()[-1].split()
    assert out["default"][0] != "0" and out["default"][1] == "b'16'", out        # no device: an error code, after the setenv
    assert out["keep"][0] != "0" and out["keep"][1] == "None", out
