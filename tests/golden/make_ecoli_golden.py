#!/usr/bin/env python3
"""Pin the oracle's full per-read chain (event detection -> method-of-moments scalings -> ABEA ->
recalibration) against the reference's own goldens for test/ecoli_2kb_region, and mint a small fixture.

Run in the build container only (needs /root/reference and /opt/conda/bin/h5dump).  Reads DATA files only:
  fast5_files/*.fast5 (raw signal + channel scaling), reads.fasta,
  adaptive.exp / est_scalings.exp / recalib_scalings.exp (143 BAM records over 112 reads; order unknown
  without htslib, so lines are matched as a multiset of printed values).
Writes tests/golden/ecoli_reads.npz: int16 signals + scaling + sequence + the golden line of a subset of
reads, so the chain is also checked where /root/reference is absent (GPU box); `ada` is the oracle's own
"sum n_aligned", `ada_printed` the reference's printed adaptive.exp record of that read.
Writes tests/golden/ecoli_events.npz: for EVERY read of the set the mean-only event table (float32; ABEA and
scaling_single read nothing else of event_t, align.c:131,738), the sequence, the method-of-moments scalings as float32
bits, and the reference's PRINTED adaptive.exp / est_scalings.exp / recalib_scalings.exp records it matched — so that
configs[0] is checked GPU-vs-reference on all 111 reads of the BAM on the GPU box (round-2 verdict item 3).
"""
import glob, os, re, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/test/ecoli_2kb_region"
H5DUMP = "/opt/conda/bin/h5dump"
OUT = os.path.dirname(os.path.abspath(__file__))


def h5attr(path, attr):
    o = subprocess.check_output([H5DUMP, "-a", attr, "-m", "%.17g", path], text=True)
    m = re.search(r"\(0\):\s*(.+)", o)
    return m.group(1).strip().strip('"')


def read_fast5(path):
    names = subprocess.check_output([H5DUMP, "-n", "1", path], text=True)
    rd = re.search(r"group\s+(/Raw/Reads/Read_\d+)", names).group(1)
    with tempfile.NamedTemporaryFile(suffix=".bin") as t:
        subprocess.check_call([H5DUMP, "-d", rd + "/Signal", "-b", "LE", "-o", t.name, path],
                              stdout=subprocess.DEVNULL)
        sig = np.fromfile(t.name, dtype=np.int16)
    ch = "/UniqueGlobalKey/channel_id/"
    return dict(read_id=h5attr(path, rd + "/read_id"), signal=sig,
                digitisation=float(h5attr(path, ch + "digitisation")), offset=float(h5attr(path, ch + "offset")),
                range=float(h5attr(path, ch + "range")))


def chain(orc, model, k, seq, f5):
    ev, _ = orc.getevents(f5["signal"], f5["offset"], f5["range"], f5["digitisation"])
    scale, shift = orc.estimate_scalings(seq, model, k, ev)
    pairs, d = orc.align(seq, ev, model, k, scale, shift)
    rec = orc.scaling_single(pairs, seq, ev, model, k, scale, shift) if len(pairs) else None
    return ev, (scale, shift), pairs, d, rec


def main(write=True):
    from oracle import orc
    from f5c_amd import load_model_f32
    k, model = load_model_f32(os.path.join(OUT, "r9.4_450bps.6mer.f32"))
    seqs = {}
    name = None
    for ln in open(f"{REF}/reads.fasta"):
        if ln.startswith(">"):
            name = ln[1:].split()[0]; seqs[name] = []
        else:
            seqs[name].append(ln.strip())
    seqs = {n: "".join(v).encode() for n, v in seqs.items()}
    g = np.load(os.path.join(OUT, "ecoli_summary.npz"))
    # adaptive.exp was printed by a build whose emission sum differs from today's align.c in the 7th
    # significant digit (e.g. single_read: -20697.529925 there, -20697.528040 from align.c:476 semantics on the
    # same events); the integer n_aligned_events must match exactly, sum/avg to 1e-6 relative
    gold_rows = g["adaptive"]
    def ada_match(d):
        s_, n_ = d["sum_emission"], d["n_aligned"]
        hit = gold_rows[(gold_rows[:, 1] == n_) & (np.abs(gold_rows[:, 0] - s_) <= 1e-6 * abs(s_) + 1e-6)]
        return hit[0] if len(hit) else None
    gold_ada = {"%.0f" % r[1] for r in gold_rows}
    gold_rec = {"%.2f %.2f %.2f" % tuple(r) for r in g["recalib"]}
    gold_est = {"%.2f %.2f" % (a, b) for a, b in zip(g["est_shift"], g["est_scale"])}
    files = sorted(glob.glob(f"{REF}/fast5_files/*.fast5"))
    ok_ada = ok_rec = ok_est = n = 0
    seen_ada = set()
    keep = []
    allreads = []
    for path in files:
        f5 = read_fast5(path)
        seq = seqs[f5["read_id"]]
        ev, (scale, shift), pairs, d, rec = chain(orc, model, k, seq, f5)
        n += 1
        ada = "%.0f" % d["n_aligned"]
        ada_row = ada_match(d)
        ada_ok = ada_row is not None
        est = "%.2f %.2f" % (shift, scale)
        recs = "%.2f %.2f %.2f" % (rec["scalings"]["shift"], rec["scalings"]["scale"], rec["scalings"]["var"]) if rec else None
        ok_ada += ada_ok; ok_est += est in gold_est; ok_rec += (recs in gold_rec) if recs else 0
        seen_ada.add(ada)
        if ada_ok and (est in gold_est) and recs in gold_rec and len(keep) < 10 and len(f5["signal"]) < 120000:
            keep.append(dict(f5, seq=seq, ada="%.6f %d" % (d["sum_emission"], d["n_aligned"]), est=est, rec=recs,
                             ada_printed="%.6f %d" % (ada_row[0], ada_row[1]), n_events=len(ev)))
        if ada_ok:                                        # the 111 reads of the BAM (the 112th FAST5 has no record)
            allreads.append(dict(read_id=f5["read_id"], seq=seq, mean=ev["mean"].astype(np.float32),
                                 scale=np.float32(scale), shift=np.float32(shift),
                                 printed_sum=ada_row[0], printed_n=int(ada_row[1]), printed_avg=ada_row[2],
                                 oracle_sum=float(d["sum_emission"]),
                                 est=est if est in gold_est else "", rec=recs if recs in gold_rec else "",
                                 n_samples=len(f5["signal"])))
        if not ada_ok or est not in gold_est or recs not in gold_rec:
            print("MISMATCH", f5["read_id"], "n_aligned", ada, ada_ok, "est", est, est in gold_est, "recalib", recs,
                  recs in gold_rec)
    print(f"{n} reads: adaptive.exp lines matched {ok_ada}, est_scalings {ok_est}, recalib {ok_rec}; "
          f"golden unique adaptive lines {len(gold_ada)}, covered {len(seen_ada & gold_ada)}")
    if not write:
        return dict(n=n, adaptive=ok_ada, est=ok_est, recalib=ok_rec, gold_unique=len(gold_ada),
                    covered=len(seen_ada & gold_ada))
    np.savez_compressed(os.path.join(OUT, "ecoli_reads.npz"),
                        n=len(keep),
                        **{f"sig{i}": r["signal"] for i, r in enumerate(keep)},
                        **{f"seq{i}": np.frombuffer(r["seq"], dtype=np.uint8) for i, r in enumerate(keep)},
                        scaling=np.array([[r["offset"], r["range"], r["digitisation"]] for r in keep]),
                        ada=np.array([r["ada"] for r in keep]), ada_printed=np.array([r["ada_printed"] for r in keep]),
                        est=np.array([r["est"] for r in keep]),
                        rec=np.array([r["rec"] for r in keep]), n_events=np.array([r["n_events"] for r in keep]),
                        read_id=np.array([r["read_id"] for r in keep]),
                        summary=np.array([n, ok_ada, ok_est, ok_rec, len(gold_ada), len(seen_ada & gold_ada)]))
    ev_ptr = np.concatenate([[0], np.cumsum([len(r["mean"]) for r in allreads])]).astype(np.int64)
    seq_ptr = np.concatenate([[0], np.cumsum([len(r["seq"]) for r in allreads])]).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "ecoli_events.npz"),
                        read_id=np.array([r["read_id"] for r in allreads]),
                        mean=np.concatenate([r["mean"] for r in allreads]), ev_ptr=ev_ptr,
                        seq=np.frombuffer(b"".join(r["seq"] for r in allreads), dtype=np.uint8), seq_ptr=seq_ptr,
                        scale=np.array([r["scale"] for r in allreads], dtype=np.float32),
                        shift=np.array([r["shift"] for r in allreads], dtype=np.float32),
                        printed_sum=np.array([r["printed_sum"] for r in allreads]),
                        printed_n=np.array([r["printed_n"] for r in allreads], dtype=np.int64),
                        printed_avg=np.array([r["printed_avg"] for r in allreads]),
                        oracle_sum=np.array([r["oracle_sum"] for r in allreads]),
                        est=np.array([r["est"] for r in allreads]), rec=np.array([r["rec"] for r in allreads]),
                        n_samples=np.array([r["n_samples"] for r in allreads], dtype=np.int64))
    print("ecoli_events.npz: %d reads, %d events" % (len(allreads), ev_ptr[-1]))
    return 0 if ok_ada >= n - 1 else 1


if __name__ == "__main__":
    sys.exit(main(write="--check" not in sys.argv))
