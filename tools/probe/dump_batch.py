"""Write a flattened synthetic batch in the tests/shim_driver.cpp input format (a C++ caller without Python/torch)."""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from f5c_amd import synth, load_model_f32
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
k, model = load_model_f32(os.path.join(ROOT, "tests/golden/r9.4_450bps.6mer.f32"))
cfg = synth.CONFIGS[sys.argv[1]]
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else cfg["n_reads"]
b = synth.make_batch(n_reads, model, k, seed=cfg["seed"], law=cfg["law"], workers=16)
n = len(b["read_len"])
with open(sys.argv[2], "wb") as f:
    f.write(struct.pack("<4i", n, k, len(model), 0)); f.write(model.tobytes()); f.write(b["read_len"].tobytes())
    f.write(b["n_events"].tobytes()); f.write(b["scalings"].tobytes())
    for i in range(n):
        s, L = int(b["read_ptr"][i]), int(b["read_len"][i])
        f.write(b["reads"][s:s + L].tobytes())
    f.write(b["events"].tobytes())
print(n, "reads", int(b["n_events"].sum()), "events")
