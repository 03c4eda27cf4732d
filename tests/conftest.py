import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import f5c_amd  # noqa: E402,F401  (sets GPU_MAX_HW_QUEUES before _have_gpu() below initialises the HIP runtime)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Run order (round-5 verdict item 1b): under `pytest -x` a failing INFRASTRUCTURE test (a subprocess of bench.py, torchrun) must
# never sit in front of a parity test again.  Files run in this order, everything comparing the HIP path with the oracle or the
# reference's printed goldens first (configs[0] = the ecoli reads at the very front), tests that launch bench.py last.
PARITY_FIRST = ["test_oracle_ecoli", "test_gpu_parity", "test_hmm_pin", "test_process_chain", "test_rna_events", "test_chain_pipeline",
                "test_hmm_gpu", "test_fuzz_gpu", "test_full_size", "test_host_plan_and_async", "test_host_pipeline"]
INFRA_LAST = ("test_bench_", "test_zz_")       # subprocess-of-bench.py / torch.distributed.run tests, whatever file they live in


def _run_order(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    infra = item.name.startswith(INFRA_LAST) or mod.startswith("test_zz_")
    return (1 if infra else 0, PARITY_FIRST.index(mod) if mod in PARITY_FIRST else len(PARITY_FIRST))


def pytest_collection_modifyitems(config, items):
    """Parity tests first, infrastructure last (stable sort: the order inside a file is kept).  Every `gpu` test is skipped before
    any batch is generated or context built when no device is present (a plain `pytest tests` on a CPU box used to build the 60 GB
    full-size batch and die; round-2 advisor finding)."""
    items.sort(key=_run_order)
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible (run with -m gpu on an MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def r9():
    from f5c_amd import load_model_f32
    return load_model_f32(os.path.join(GOLDEN, "r9.4_450bps.6mer.f32"))


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def single_read():
    from f5c_amd.types import EVENT_DT
    g = np.load(os.path.join(GOLDEN, "single_read.npz"))
    ev = np.zeros(len(g["mean"]), dtype=EVENT_DT)
    for f in ("start", "length", "mean", "stdv"):
        ev[f] = g[f]
    return dict(seq=g["seq"].tobytes(), events=ev, g=g)


@pytest.fixture(scope="session")
def ctx(r9):
    """Shared device context for the gpu tests (small arena so sub-batching is exercised elsewhere)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from f5c_amd import abea
    k, model = r9
    c = abea.AbeaContext(model, k, device_id=0, max_arena_bytes=4 << 30)
    yield c
    c.close()
