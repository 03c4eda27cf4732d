"""Row N3 (and the event_db shim of N2) with NO Python in the chain: tests/process_driver.cpp, compiled with plain g++,
calls abea_f5c_process / abea_f5c_event_db + abea_f5c_align_scale / abea_rsq_format_batch through the C ABI, as f5c's
process_db_rsq + output_db_rsq (src/resquiggle.c:283-449) and process_db's --print-scaling block (src/f5c.c:1008-1020)
would.  Checked bit for bit against the oracle chain (getevents -> estimate_scalings -> align -> scaling_single ->
rsq_format) and against the reference's PRINTED recalib_scalings.exp / adaptive.exp records."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EVENT_DT = np.dtype([("start", "<u8"), ("length", "<f4"), ("mean", "<f4"), ("stdv", "<f4")], align=True)


def _build_driver(tmp_path):
    exe = str(tmp_path / "process_driver")
    subprocess.check_call(["g++", "-std=c++11", "-O2", os.path.join(ROOT, "tests", "process_driver.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "f5c_amd"), "-labea_hip", "-Wl,-rpath," + os.path.join(ROOT, "f5c_amd")])
    return exe


def _dump(path, model, k, mode, reads, rna=0):
    """reads: dicts with seq (bytes), read_id, n_samples, offset/range/digitisation, and sig (int16) or events + scale/shift"""
    n = len(reads)
    with open(path, "wb") as f:
        f.write(np.array([n, k, len(model), mode, rna, 0], dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(model).tobytes())
        f.write(np.array([len(r["seq"]) for r in reads], dtype=np.int32).tobytes())
        f.write(np.array([r["n_samples"] for r in reads], dtype=np.int64).tobytes())
        f.write(np.array([[r["offset"], r["range"], r["digitisation"]] for r in reads], dtype=np.float32).tobytes())
        for r in reads:
            f.write(r["read_id"].encode()[:39].ljust(40, b"\0"))
        for r in reads:
            f.write(r["seq"])
        if mode in (0, 2):
            for r in reads:
                if r["n_samples"] > 0:
                    f.write(r["sig"].astype(np.float32).tobytes())        # f5c holds the ADC counts as float (f5c.h:276-286)
        else:
            f.write(np.array([len(r["events"]) for r in reads], dtype=np.int32).tobytes())
            sc = np.zeros((n, 4), dtype=np.float32)
            sc[:, 0] = [r["scale"] for r in reads]
            sc[:, 1] = [r["shift"] for r in reads]
            sc[:, 2] = 1.0
            f.write(sc.tobytes())
            for r in reads:
                f.write(np.ascontiguousarray(r["events"]["mean"], dtype=np.float32).tobytes())


def _read_events(path, n):
    out = []
    with open(path, "rb") as f:
        for _ in range(n):
            ne = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            out.append(np.frombuffer(f.read(24 * ne), dtype=EVENT_DT).copy())
    return out


def _raw_reads():
    z = np.load(os.path.join(GOLD, "ecoli_reads.npz"))
    out = []
    for i in range(int(z["n"])):
        off, rng, dig = z["scaling"][i]
        out.append(dict(sig=z[f"sig{i}"], seq=z[f"seq{i}"].tobytes(), offset=off, range=rng, digitisation=dig,
                        n_samples=len(z[f"sig{i}"]), read_id=str(z["read_id"][i]), rec=str(z["rec"][i]),
                        ada_printed=str(z["ada_printed"][i])))
    return out


@pytest.mark.gpu
def test_cpp_process_chain_on_real_reads(tmp_path, orc, r9):
    """10 real reads + one read without signal: raw float signals in, everything f5c holds after process_db_rsq out."""
    k, model = r9
    reads = _raw_reads()
    reads.insert(3, dict(sig=np.zeros(0, np.int16), seq=b"ACGTACGTACGTAAC", offset=1.0, range=1400.0, digitisation=8192.0,
                         n_samples=0, read_id="no-signal", rec="", ada_printed=""))
    n = len(reads)
    exe = _build_driver(tmp_path)
    _dump(str(tmp_path / "b0.bin"), model, k, 0, reads)
    _dump(str(tmp_path / "b2.bin"), model, k, 2, reads)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([exe, str(tmp_path / "b0.bin"), str(tmp_path / "o0")], env=env)
    subprocess.check_call([exe, str(tmp_path / "b2.bin"), str(tmp_path / "o2")], env=env)
    for ext in (".events", ".state", ".pairs", ".b2e", ".scaling", ".tsv", ".paf", ".pa"):       # one call or two: same bytes
        assert open(str(tmp_path / "o0") + ext, "rb").read() == open(str(tmp_path / "o2") + ext, "rb").read(), ext
    evs = _read_events(str(tmp_path / "o0.events"), n)
    state = [l.rstrip("\n").split("\t") for l in open(str(tmp_path / "o0.state"))]
    b2e = [l.rstrip("\n") for l in open(str(tmp_path / "o0.b2e"))]
    pair_lines = open(str(tmp_path / "o0.pairs")).read().split("\n")
    pa = np.fromfile(str(tmp_path / "o0.pa"), dtype=np.float32)
    want_tsv, want_paf, want_scaling, want_pairs = "", "", "read\tshift\tscale\tvar\n", []
    first_sc = None
    pa_off = 0
    for i, r in enumerate(reads):
        st = state[i]
        if r["n_samples"] == 0:                                            # f5c.c:727-731, 826-828, 786-794
            assert len(evs[i]) == 0 and int(st[7]) == 0 and int(st[1]) & 2 and b2e[i] == "NULL"
            if first_sc is None:
                first_sc = (np.float32(float.fromhex(st[5])), np.float32(float.fromhex(st[4])))
            continue
        o_ev, o_pa = orc.getevents(r["sig"], r["offset"], r["range"], r["digitisation"])
        assert len(evs[i]) == len(o_ev)
        for f in ("start", "length", "mean", "stdv"):
            assert (evs[i][f] == o_ev[f]).all(), (i, f)
        assert (pa[pa_off:pa_off + r["n_samples"]] == o_pa).all()         # the signal is left in pA (f5c.c:693-696)
        pa_off += r["n_samples"]
        scale, shift = orc.estimate_scalings(r["seq"], model, k, o_ev)
        o_pairs, d = orc.align(r["seq"], o_ev, model, k, scale, shift)
        rec = orc.scaling_single(o_pairs, r["seq"], o_ev, model, k, scale, shift)
        sc = rec["scalings"]
        assert int(st[1]) == rec["flag"] and int(st[2]) == rec["n_alignment"] and int(st[7]) == len(o_pairs)
        assert float.fromhex(st[3]) == rec["events_per_base"]
        assert (np.float32(float.fromhex(st[4])), np.float32(float.fromhex(st[5])), np.float32(float.fromhex(st[6]))) == \
               (sc["shift"], sc["scale"], sc["var"])
        assert np.float32(float.fromhex(st[8])) == sc["log_var"]           # align.c:758-760: what the HMM stage reads (hmm.c:101)
        m = rec["base_to_event_map"]
        assert b2e[i] == "".join("%d,%d " % (a, b) for a, b in zip(m["start"], m["stop"]))
        gn = r["ada_printed"].split()[1]
        assert len(o_pairs) == int(gn)                                     # adaptive.exp as the reference printed it
        want_pairs += [">%s\tN_ALGN_PAIR:%d\t{ref_pos,read_pos}" % (r["read_id"], len(o_pairs)),
                       "".join("{%d,%d}\t" % (a, b) for a, b in zip(o_pairs["ref_pos"], o_pairs["read_pos"]))]
        if first_sc is None:
            first_sc = (sc["scale"], sc["shift"])
        if rec["flag"] == 0:
            assert "%.2f %.2f %.2f" % (sc["shift"], sc["scale"], sc["var"]) == r["rec"]      # recalib_scalings.exp
            want_scaling += "%s\t%.2f\t%.2f\t%.2f\n" % (r["read_id"], sc["shift"], sc["scale"], sc["var"])
    # output_db_rsq prints the FIRST read's scalings in every PAF line (db->scalings->scale, resquiggle.c:443-444)
    for i, r in enumerate(reads):
        if r["n_samples"] == 0 or int(state[i][1]) != 0:
            continue
        o_ev, _ = orc.getevents(r["sig"], r["offset"], r["range"], r["digitisation"])
        scale, shift = orc.estimate_scalings(r["seq"], model, k, o_ev)
        o_pairs, _ = orc.align(r["seq"], o_ev, model, k, scale, shift)
        m = orc.scaling_single(o_pairs, r["seq"], o_ev, model, k, scale, shift)["base_to_event_map"]
        mm = np.stack([m["start"], m["stop"]], axis=1).astype(np.int32)
        want_tsv += orc.rsq_format(0, r["read_id"], len(r["seq"]), k, mm, o_ev, r["n_samples"], first_sc[0], first_sc[1])
        want_paf += orc.rsq_format(1, r["read_id"], len(r["seq"]), k, mm, o_ev, r["n_samples"], first_sc[0], first_sc[1])
    assert pair_lines[:-1] == want_pairs
    assert open(str(tmp_path / "o0.scaling")).read() == want_scaling
    assert open(str(tmp_path / "o0.tsv")).read() == want_tsv
    assert open(str(tmp_path / "o0.paf")).read() == want_paf
    assert want_tsv.count("\n") > 50000 and want_paf.count("\n") == 10


@pytest.mark.gpu
def test_cpp_align_scale_all_111_reads_print_scaling(tmp_path, orc, r9):
    """configs[0]'s 111 reads from a g++ caller through abea_f5c_align_scale: the --print-scaling block (f5c.c:1008-1020)
    equals the reference's printed recalib_scalings.exp records, the --print-banded-aln counts its adaptive.exp."""
    k, model = r9
    z = np.load(os.path.join(GOLD, "ecoli_events.npz"))
    reads = []
    for i in range(len(z["read_id"])):
        a, b = int(z["ev_ptr"][i]), int(z["ev_ptr"][i + 1])
        ev = np.zeros(b - a, dtype=EVENT_DT)
        ev["mean"] = z["mean"][a:b]
        s0, s1 = int(z["seq_ptr"][i]), int(z["seq_ptr"][i + 1])
        reads.append(dict(seq=z["seq"][s0:s1].tobytes(), read_id=str(z["read_id"][i]), n_samples=int(z["n_samples"][i]),
                          offset=0.0, range=1.0, digitisation=1.0, events=ev, scale=float(z["scale"][i]), shift=float(z["shift"][i]),
                          rec=str(z["rec"][i]), printed_n=int(z["printed_n"][i])))
    exe = _build_driver(tmp_path)
    _dump(str(tmp_path / "b1.bin"), model, k, 1, reads)
    subprocess.check_call([exe, str(tmp_path / "b1.bin"), str(tmp_path / "o1")], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = open(str(tmp_path / "o1.scaling")).read().split("\n")
    assert lines[0] == "read\tshift\tscale\tvar"
    got = {l.split("\t")[0]: " ".join(l.split("\t")[1:]) for l in lines[1:] if l}
    n_rec = 0
    for r in reads:
        if r["rec"]:                                                       # 110 of the 111 printed lines (read 95ff70e7)
            assert got[r["read_id"]] == r["rec"], r["read_id"]
            n_rec += 1
    assert n_rec == 110 and len(got) == 111
    heads = [l for l in open(str(tmp_path / "o1.pairs")) if l.startswith(">")]
    assert [int(h.split("N_ALGN_PAIR:")[1].split("\t")[0]) for h in heads] == [r["printed_n"] for r in reads]


def test_rsq_format_batch_against_per_read_calls(orc, r9):
    """abea_rsq_format_batch (host-only) = output_db_rsq's loop: flagged reads are skipped, the PAF tags carry the first
    read's scalings, and the caller's maps are left alone (RNA too), so that the size query and the real call agree."""
    import ctypes as C
    from f5c_amd import abea, synth
    k, model = r9
    batch = synth.make_batch(9, model, k, seed=77, law=700, bad_frac=0.0)
    lib = abea.load_library()
    lib.abea_rsq_format_batch.restype = C.c_int64
    ids, lens, maps, evs, ns, flags = [], [], [], [], [], []
    for i in range(9):
        s, L = int(batch["read_ptr"][i]), int(batch["read_len"][i])
        es, E = int(batch["event_ptr"][i]), int(batch["n_events"][i])
        seq, ev = batch["reads"][s:s + L].tobytes(), np.ascontiguousarray(batch["events"][es:es + E])
        sc = batch["scalings"][i]
        pairs, _ = orc.align(seq, ev, model, k, sc["scale"], sc["shift"])
        assert len(pairs)
        m = orc.scaling_single(pairs, seq, ev, model, k, sc["scale"], sc["shift"])["base_to_event_map"]
        maps.append(np.stack([m["start"], m["stop"]], axis=1).astype(np.int32).copy())
        ids.append(f"read-{i}".encode()); lens.append(L); evs.append(ev); ns.append(int(ev["start"][-1] + ev["length"][-1]))
        flags.append(1 if i in (2, 5) else 0)
    scal = np.zeros((9, 4), dtype=np.float32)
    scal[:, 0] = 1.0 + 0.01 * np.arange(9)
    scal[:, 1] = 3.0 - np.arange(9)
    for rna in (0, 1):
        use = [m[::-1, ::-1].copy() if rna else m.copy() for m in maps]
        before = [m.copy() for m in use]
        for fmt in (0, 1):
            args = (fmt, 9, (C.c_char_p * 9)(*ids), np.array(lens, np.int32).ctypes.data_as(C.c_void_p), C.c_uint32(k),
                    (C.c_void_p * 9)(*[m.ctypes.data for m in use]), (C.c_void_p * 9)(*[e.ctypes.data for e in evs]),
                    np.array(ns, np.int64).ctypes.data_as(C.c_void_p), scal.ctypes.data_as(C.c_void_p),
                    np.array(flags, np.int32).ctypes.data_as(C.c_void_p), rna)
            printed = C.c_int32(-1)
            need = lib.abea_rsq_format_batch(None, C.c_size_t(0), *args, C.byref(printed))
            assert need > 0 and printed.value == 7
            buf = C.create_string_buffer(need + 1)
            assert lib.abea_rsq_format_batch(buf, C.c_size_t(need + 1), *args, C.byref(printed)) == need
            want = "".join(orc.rsq_format(fmt, ids[i].decode(), lens[i], k, before[i], evs[i], ns[i], scal[0, 0], scal[0, 1], rna=bool(rna))
                           for i in range(9) if not flags[i])
            assert buf.value.decode() == want
        assert all((a == b).all() for a, b in zip(use, before))
