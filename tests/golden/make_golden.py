#!/usr/bin/env python3
"""Mint the committed golden fixtures from the reference's own test DATA files.

Run in the build container only (needs /root/reference). It reads data files the reference's
tests hold (no reference source code is read or copied):
  test/r9-models/r9.4_450bps.nucleotide.6mer.template.model  -> r9.4_450bps.6mer.f32  (4096 x {mean,stdv})
  test/ecoli_2kb_region/single_read/{read1.fasta,read1.events.exp,adaptive.exp,read1.scalings.exp}
                                                             -> single_read.npz
  test/ecoli_2kb_region/{adaptive.exp,est_scalings.exp,recalib_scalings.exp} -> ecoli_summary.npz
"""
import os
import re
import numpy as np

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))


def model_table(path, alphabet=4):
    rows = []
    k = None
    for ln in open(path):
        if ln.split()[:1] == ["#k"]:
            k = int(ln.split()[1])
        if ln.startswith("#") or ln.startswith("kmer") or not ln.strip():
            continue
        f = ln.split()
        rows.append((f[0], np.float32(f[1]), np.float32(f[2])))
    kmers = [r[0] for r in rows]
    assert kmers == sorted(kmers) and len(rows) == alphabet ** k
    return k, np.array([[r[1], r[2]] for r in rows], dtype=np.float32)


def main():
    k, tab = model_table(f"{REF}/r9-models/r9.4_450bps.nucleotide.6mer.template.model")
    assert k == 6
    tab.tofile(f"{OUT}/r9.4_450bps.6mer.f32")

    d = f"{REF}/ecoli_2kb_region/single_read"
    seq = "".join(l.strip() for l in open(f"{d}/read1.fasta") if not l.startswith(">"))
    ev = re.findall(r"\{(\d+),([-\d.]+),([-\d.]+),([-\d.]+),-1,-1\}", open(f"{d}/read1.events.exp").read())
    start = np.array([int(e[0]) for e in ev], dtype=np.uint64)
    length = np.array([np.float32(e[1]) for e in ev], dtype=np.float32)
    mean = np.array([np.float32(e[2]) for e in ev], dtype=np.float32)
    stdv = np.array([np.float32(e[3]) for e in ev], dtype=np.float32)
    m = re.search(r"sum_emission ([-\d.]+), n_aligned_events ([\d.]+), avg_log_emission ([-\d.]+)",
                  open(f"{d}/adaptive.exp").read())
    sc = open(f"{d}/read1.scalings.exp").read()
    shift = float(re.search(r"shift: ([-\d.]+)", sc).group(1))
    scale = float(re.search(r"scale: ([-\d.]+)", sc).group(1))
    ev_mean, km_mean = map(float, re.search(r"event mean: ([-\d.]+) kmer mean: ([-\d.]+)", sc).groups())
    np.savez_compressed(f"{OUT}/single_read.npz", seq=np.frombuffer(seq.encode(), dtype=np.uint8),
                        start=start, length=length, mean=mean, stdv=stdv,
                        exp_sum_emission=float(m.group(1)), exp_n_aligned=int(float(m.group(2))),
                        exp_avg_log_emission=float(m.group(3)), exp_shift=shift, exp_scale=scale,
                        exp_event_mean=ev_mean, exp_kmer_mean=km_mean)

    # ---- N4 pin: the reference prints the arguments (meth_input.exp) and the results (meth.exp) of read1's HMM calls
    kc, cpg = model_table(f"{REF}/r9-models/r9.4_450bps.cpg.6mer.template.model", alphabet=5)   # sorted ACGMT = rank order (hmm.c:30-61)
    assert kc == 6
    cpg.tofile(f"{OUT}/r9.4_450bps.cpg.6mer.f32")
    kr, rna = model_table(f"{REF}/r9-models/r9.4_70bps.u_to_t_rna.5mer.template.model")
    assert kr == 5
    rna.tofile(f"{OUT}/r9.4_70bps.rna.5mer.f32")
    txt = open(f"{d}/meth_input.exp").read().splitlines()
    assert len(txt) % 3 == 0
    mseq, mrc, arg = [], [], []
    for i in range(0, len(txt), 3):
        mseq.append(txt[i].split(" : ")[1].strip())
        mrc.append(txt[i + 1].split(" : ")[1].strip())
        g = re.match(r"event_start_idx (\d+), event_stop_idx (\d+), event_stride (-?\d+), rc (\d)", txt[i + 2])
        arg.append([int(x) for x in g.groups()])
    rows = [l.split("\t") for l in open(f"{d}/meth.exp").read().splitlines()[1:]]
    assert len(rows) * 2 == len(mseq)
    for gi, r in enumerate(rows):            # job 2g = unmethylated, 2g+1 = methylated sequence of group g (meth.c:473-474)
        assert "M" not in mseq[2 * gi] and "M" in mseq[2 * gi + 1] and r[9] in mseq[2 * gi]
    np.savez_compressed(f"{OUT}/single_read_meth.npz", m_seq=np.array(mseq), m_rc_seq=np.array(mrc),
                        args=np.array(arg, dtype=np.int32),           # event_start_idx, event_stop_idx, stride, rc
                        exp_log_lik_methylated=np.array([float(r[5]) for r in rows]),
                        exp_log_lik_unmethylated=np.array([float(r[6]) for r in rows]),
                        exp_log_lik_ratio=np.array([float(r[4]) for r in rows]),
                        exp_num_cpgs=np.array([int(r[8]) for r in rows]), group_seq=np.array([r[9] for r in rows]))

    d = f"{REF}/ecoli_2kb_region"
    ada = np.array([[float(x) for x in re.findall(r"-?\d+\.?\d*|-?nan|-?inf", l)] for l in open(f"{d}/adaptive.exp") if l.startswith("sum_emission")])
    est = open(f"{d}/est_scalings.exp").read()
    est_shift = np.array([float(x) for x in re.findall(r"shift: ([-\d.]+)", est)])
    est_scale = np.array([float(x) for x in re.findall(r"scale: ([-\d.]+)", est)])
    rec = np.array([[float(x) for x in re.findall(r"-?\d+\.?\d*", l)] for l in open(f"{d}/recalib_scalings.exp") if l.strip()])
    np.savez_compressed(f"{OUT}/ecoli_summary.npz", adaptive=ada, est_shift=est_shift, est_scale=est_scale,
                        recalib=rec)
    print("single_read: L=%d E=%d; ecoli: %d adaptive rows, %d est, %d recalib" %
          (len(seq), len(ev), len(ada), len(est_shift), len(rec)))


if __name__ == "__main__":
    main()
