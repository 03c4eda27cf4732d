"""f5c_amd — MI355X-native adaptive banded event alignment (ABEA), the f5c `align_db` hot path.

The product is the C-ABI library f5c_amd/libabea_hip.so (include/abea.h) built from
f5c_amd/csrc/*.hip; this Python package is the thin host-side mirror used by tests and bench.py.
It never imports the CPU oracle (oracle/) and has no CPU fallback: without the HIP library and a
GPU every compute entry point raises.
"""
from .types import EVENT_DT, MODEL_DT, PAIR_DT, SCAL_DT, DIAG_DT  # noqa: F401
from .model import load_model_f32, read_model_text, synthetic_model  # noqa: F401
