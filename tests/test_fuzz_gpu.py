"""Seeded, time-boxed slices of the two randomised parity sweeps (tools/fuzz_parity.py, tools/fuzz_events.py) under
-m gpu, so that the class of defect they found in round 2 (a wave shuffle under a divergent select in the N1 kernel)
stays covered by the driver's run.  Same seeds every run; the batch sequence is deterministic, the time box only bounds
how far along it a slow box gets."""
import os
import sys
import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _tool(name):
    if TOOLS not in sys.path:
        sys.path.insert(0, TOOLS)
    return __import__(name)


@pytest.mark.parametrize("seed,k9", [(20250901, False), (20250902, True)])
def test_fuzz_alignment_and_scaling_slice(seed, k9):
    """Extreme read shapes (k-base reads, single events, events/base either side of the 15.0 guard, repeated / truncated
    event tables, constant signals, odd scalings) through the device entry, the host entry and the device scaling_single,
    all bit-exact against the oracle."""
    nb, nr, npass, dt = _tool("fuzz_parity").run(budget=25.0, seed=seed, k9=k9, max_batches=400)
    assert nb >= 5 and nr > 50 and npass > 0, (nb, nr, npass, dt)


def test_fuzz_event_detection_slice():
    """Signals of every length from one sample, steps / plateaus / ramps / noise / full-range ADC values, offsets next to
    0 pA: device event detection + method-of-moments scalings bit-exact against the oracle."""
    nb, ns, ne, dt = _tool("fuzz_events").run(budget=40.0, seed=20250903, max_batches=400)
    assert nb >= 3 and ns > 30 and ne > 1000, (nb, ns, ne, dt)


def test_fuzz_event_db_host_entry_slice(monkeypatch):
    """The same adversarial signals as float ADC counts through abea_events_batch_host (the chunk pipeline of abea_chain.cpp) with
    random chunk sizes, slot counts and first-guess table capacities — overflowing tables are redone from the int16 staging after
    the pipeline has drained — DNA and RNA: tables and method-of-moments scalings bit-exact against the oracle."""
    for var in ("ABEA_CHAIN_CAP_DIV", "ABEA_CHAIN_SLOTS", "ABEA_CHAIN_CHUNK_SAMPLES", "ABEA_CHAIN_CHUNK_READS"):
        monkeypatch.setenv(var, "1")                        # restored by monkeypatch after the fuzzer has overwritten them
    nb, ns, ne, dt = _tool("fuzz_events").run(budget=20.0, seed=20250905, max_batches=200, host_entry=True)
    assert nb >= 3 and ns > 30 and ne > 1000, (nb, ns, ne, dt)
