"""f5c_amd — MI355X-native adaptive banded event alignment (ABEA), the f5c `align_db` hot path.

The product is the C-ABI library f5c_amd/libabea_hip.so (include/abea.h) built from
f5c_amd/csrc/*.hip; this Python package is the thin host-side mirror used by tests and bench.py.
It never imports the CPU oracle (oracle/) and has no CPU fallback: without the HIP library and a
GPU every compute entry point raises.
"""
import os as _os

# The host entry keeps 8 chunks in flight on 8 HIP streams.  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default
# 4) and serialises the kernels of streams that share one: measured on MI355X, 422 ms per 100 k-read batch with 4 queues, 370
# with 16 (profiles/r05/hw_queues_ab.txt).  The runtime reads the variable once, when it initialises (first HIP call), so it is
# set here, before torch or the library touches the device; abea_init() does the same for C callers.
if not _os.environ.get("ABEA_KEEP_HW_QUEUES"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .types import EVENT_DT, MODEL_DT, PAIR_DT, SCAL_DT, DIAG_DT  # noqa: F401
from .model import load_model_f32, read_model_text, synthetic_model  # noqa: F401
