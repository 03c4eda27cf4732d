"""Row N3: the resquiggle text of one read (abea_rsq_format, host-only code in libabea_hip.so) against a hand-derived
expectation, against the oracle's restatement of output_db_rsq (src/resquiggle.c:319-449), and on the maps that the
oracle's own scaling_single produces for synthetic reads.  No GPU needed: the formatter is host code."""
import numpy as np
import pytest

from f5c_amd.types import EVENT_DT


def _events(starts, lengths):
    ev = np.zeros(len(starts), dtype=EVENT_DT)
    ev["start"] = np.asarray(starts, dtype=np.uint64)
    ev["length"] = np.asarray(lengths, dtype=np.float32)
    return ev


def test_hand_derived_tsv_and_paf():
    """8 bases, k=6 -> 3 k-mers.  k-mer 0: events 0-1 (samples 100..130), k-mer 1: no event, k-mer 2: event 3 (samples 140..155);
    event 2 (130..140) belongs to no k-mer, so the PAF string has a 10-sample insertion after the deletion."""
    from f5c_amd import abea
    ev = _events([100, 112, 130, 140], [12, 18, 10, 15])
    m = np.array([[0, 1], [-1, -1], [3, 3]], dtype=np.int32)
    tsv = abea.rsq_format(0, "r1", 8, 6, m, ev, 1000, 1.25, -3.5)
    assert tsv == "r1\t0\t100\t130\nr1\t1\t.\t.\nr1\t2\t140\t155\n"
    paf = abea.rsq_format(1, "r1", 8, 6, m, ev, 1000, 1.25, -3.5)
    assert paf == "r1\t1000\t100\t155\t+\tr1\t3\t0\t3\t2\t3\t255\tsc:f:1.250000\tsh:f:-3.500000\tss:Z:30,1D10I15,\n"
    # leading k-mers without events are not deletions (the alignment has not started), trailing ones are never flushed
    m2 = np.array([[-1, -1], [0, 1], [-1, -1]], dtype=np.int32)
    assert abea.rsq_format(1, "r1", 8, 6, m2, ev, 1000, 1.0, 0.0).endswith("\tr1\t3\t1\t2\t1\t3\t255\tsc:f:1.000000\tsh:f:0.000000\tss:Z:30,\n")
    # RNA: the map is reversed and start/stop swapped; k-mer indices are printed from the other end
    m3 = np.array([[3, 3], [-1, -1], [1, 0]], dtype=np.int32)
    assert abea.rsq_format(0, "r1", 8, 6, m3, ev, 1000, 1.0, 0.0, rna=True) == "r1\t2\t100\t130\nr1\t1\t.\t.\nr1\t0\t140\t155\n"
    assert (m3 == [[3, 3], [-1, -1], [1, 0]]).all()             # the wrapper passes a copy
    with pytest.raises(abea.AbeaError):                          # end <= start: the reference exits
        abea.rsq_format(0, "r1", 8, 6, np.array([[1, 0], [-1, -1], [3, 3]], dtype=np.int32), ev, 1000, 1.0, 0.0)


@pytest.mark.parametrize("rna", [False, True])
def test_against_oracle_on_aligned_synthetic_reads(orc, r9, rna):
    from f5c_amd import abea, synth
    k, model = r9
    batch = synth.make_batch(12, model, k, seed=301, law=900, bad_frac=0.0)
    n_checked = 0
    for i in range(12):
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        seq, ev = batch["reads"][s:s + L].tobytes(), batch["events"][es:es + E]
        sc = batch["scalings"][i]
        pairs, _ = orc.align(seq, ev, model, k, sc["scale"], sc["shift"])
        if len(pairs) == 0:
            continue
        r = orc.scaling_single(pairs, seq, ev, model, k, sc["scale"], sc["shift"])
        m = np.stack([r["base_to_event_map"]["start"], r["base_to_event_map"]["stop"]], axis=1).astype(np.int32)
        if rna:                                                  # what the map of a 3'->5' signal looks like before the reversal
            m = m[::-1, ::-1].copy()
        nsample = int(ev["start"][-1] + ev["length"][-1])
        for fmt in (0, 1):
            want = orc.rsq_format(fmt, f"read{i}", L, k, m, ev, nsample, 1.01, 2.5, rna=rna)
            got = abea.rsq_format(fmt, f"read{i}", L, k, m, ev, nsample, 1.01, 2.5, rna=rna)
            assert want is not None and got == want
            if fmt == 0:
                assert got.count("\n") == L - k + 1
            else:
                assert got.count("\n") == 1 and "ss:Z:" in got
        n_checked += 1
    assert n_checked >= 8
