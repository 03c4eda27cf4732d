#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the committed rocprofv3 PMC passes of `bench.py --mode device` (tools/profile_r03.sh):
   python profiles/make_pmc_traffic.py profiles/r03/a_      (prefix of the *_counter_collection.csv files)
   python profiles/make_pmc_traffic.py detector profiles/r06/ <samples per call>     (adds the "detector" entry: the abea_ev_* kernels)
Every number of the json is an average over the abea_align_kernel dispatches of one pass; nothing is inferred from another
config.  Corrections / definitions (MI355X_MICROARCH.md, HBM and rocprofv3 sections):
  hbm bytes   = FETCH_SIZE(KB) x 1024 x 2  (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B)  +  WRITE_SIZE(KB) x 1024
  valu_busy   = SQ_ACTIVE_INST_VALU x 4 / (n_simd x GRBM_GUI_ACTIVE / n_xcd)     the gfx9 VALUBusy formula: SQ_ACTIVE_INST_* count
                quad-cycles summed over waves, GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs
  wave-time split: SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY = SQ_WAVE_CYCLES (issuing / issue-stalled / parked)"""
import collections, csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("f5c_amd/csrc/abea_fill.inc", "f5c_amd/csrc/abea_walk.inc", "f5c_amd/csrc/abea_kernels.hip")
ALIGN_SECTION_END = b"event detection on the device (row N2)"     # banner in abea_kernels.hip: everything before it is pre / align / copy-out


def code_sha():
    """sha256 over the sources abea_align_kernel is built from — the two generated statements and abea_kernels.hip UP TO the
    event-detection section (the abea_ev_* kernels behind that banner are other kernels: the profiled command does not run them):
    the counters below describe THIS code (run this script on the tree the passes were taken with); bench.py recomputes it and
    refuses the static numbers when it differs (round-4 verdict: a time window is not a guard)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        data = open(os.path.join(ROOT, rel), "rb").read()
        if rel.endswith("abea_kernels.hip"):
            cut = data.find(ALIGN_SECTION_END)
            assert cut > 0, "section banner of the event-detection kernels not found"
            data = data[:cut]
        h.update(data)
    return h.hexdigest()


EVENTS = {"10k": ("r9_10k_8kb", 158727291), "100k": ("r9_100k_mixed", 2514312019)}
N_SIMD, N_XCD = 1024, 8


def avg(path, kernel="abea_align_kernel"):
    acc, cnt, dur = collections.defaultdict(float), collections.Counter(), []
    for r in csv.DictReader(open(path)):
        if r["Kernel_Name"].startswith(kernel):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return {c: acc[c] / cnt[c] for c in acc}, (sum(dur) / len(dur) if dur else None), max(cnt.values()) if cnt else 0


def main(prefix):
    out = {}
    for tag, (config, events) in EVENTS.items():
        c, src, kms = {}, {}, {}
        for p in ("sqa", "sqb", "fetch", "write"):
            path = f"{prefix}pmc_{p}{tag}_counter_collection.csv"
            if not os.path.exists(path):
                continue
            v, ms, n = avg(path)
            c.update(v); kms[p] = round(ms, 3); src[p] = f"{path} ({n} launches)"
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        hbm = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
        e = {"kernel": "abea_align_kernel", "events_per_launch": events,
             "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
             "fetch_bytes_per_event_x2": c["FETCH_SIZE"] * 2048 / events, "write_bytes_per_event": c["WRITE_SIZE"] * 1024 / events,
             "hbm_bytes_per_launch": hbm, "hbm_bytes_per_event": hbm / events,
             "code_sha256": code_sha(), "code_sha256_of": list(KERNEL_SOURCES[:2]) + [KERNEL_SOURCES[2] + " (up to the event-detection section)"],
             "kernel_ms_in_each_pass": kms, "passes": src,
             "sq_counters_per_launch": {k: v for k, v in sorted(c.items()) if k.startswith("SQ_") or k.startswith("GRBM")}}
        # abea_pre_kernel of the same passes (the device-resident leg launches it once per step, in front of the alignment)
        pc, psrc, pms = {}, {}, {}
        for p in ("fetch", "write"):
            path = f"{prefix}pmc_{p}{tag}_counter_collection.csv"
            v, ms, n = avg(path, "abea_pre_kernel")
            pc.update(v); pms[p] = round(ms, 3) if ms else None; psrc[p] = f"{path} ({n} launches)"
        if "FETCH_SIZE" in pc and "WRITE_SIZE" in pc:
            phbm = pc["FETCH_SIZE"] * 1024 * 2 + pc["WRITE_SIZE"] * 1024
            e["pre_kernel"] = {"kernel": "abea_pre_kernel", "FETCH_SIZE_KB": pc["FETCH_SIZE"], "WRITE_SIZE_KB": pc["WRITE_SIZE"],
                               "fetch_bytes_per_event_x2": pc["FETCH_SIZE"] * 2048 / events, "write_bytes_per_event": pc["WRITE_SIZE"] * 1024 / events,
                               "hbm_bytes_per_launch": phbm, "hbm_bytes_per_event": phbm / events, "kernel_ms_in_each_pass": pms, "passes": psrc}
        if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            e["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (N_SIMD * c["GRBM_GUI_ACTIVE"] / N_XCD)
            e["valu_busy_formula"] = "SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs), measured on this config"
            e["shader_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / N_XCD / (kms["sqb"] * 1e6)
            w = c["SQ_WAVE_CYCLES"]
            e["wave_time_split"] = {"issuing": c["SQ_ACTIVE_INST_ANY"] / w, "issue_stalled": c["SQ_WAIT_INST_ANY"] / w,
                                    "parked_on_waitcnt": c["SQ_WAIT_ANY"] / w}
        if "SQ_INSTS_VALU" in c:
            e["valu_wave_instr_per_event"] = c["SQ_INSTS_VALU"] / events
            e["salu_wave_instr_per_event"] = c["SQ_INSTS_SALU"] / events
            e["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT")
        out[config] = e
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json"), "w"), indent=1)
    for k, e in out.items():
        print(k, {x: (round(v, 4) if isinstance(v, float) else v) for x, v in e.items() if not isinstance(v, dict)})


def detector(prefix, samples_per_call):
    """pmc_traffic.json["detector"]: HBM bytes per SAMPLE of every abea_ev_* kernel of one DNA call of tools/n2_profile.py (2048 reads)
    from its FETCH_SIZE / WRITE_SIZE passes (`prefix`n2_fetch_counter_collection.csv, `prefix`n2_write_counter_collection.csv) and, when
    the SQ passes are there (`prefix`n2_sqa_..., `prefix`n2_sqb_...), the instruction counts and the wave-time split of each kernel.
    The script runs the DNA parameters twice, then the RNA parameters twice: the DNA calls are the dispatches in front of the first
    RNA-only kernel (abea_ev_spec2_rna_kernel); a kernel's counters are summed over the two DNA calls and halved."""
    def dna_calls(path):
        rows = [r for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith("abea_ev_")]
        rna = [int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"].startswith("abea_ev_spec2_rna_kernel")]
        cut = min(rna) if rna else None
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); ms = collections.defaultdict(float)
        seen = set()
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        if cut is None:                                  # no RNA-only kernel in the trace (the array form): first half of the dispatches
            cut = ids[len(ids) // 2]
        for r in rows:
            d = int(r["Dispatch_Id"])
            if d >= cut:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]) / 2
            if (k, d) not in seen:
                seen.add((k, d)); ms[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 / 2
        return acc, ms
    f, fms = dna_calls(prefix + "n2_fetch_counter_collection.csv")
    w, _ = dna_calls(prefix + "n2_write_counter_collection.csv")
    sqa = sqb = None
    if os.path.exists(prefix + "n2_sqa_counter_collection.csv") and os.path.exists(prefix + "n2_sqb_counter_collection.csv"):
        sqa, _ = dna_calls(prefix + "n2_sqa_counter_collection.csv")
        sqb, sqb_ms = dna_calls(prefix + "n2_sqb_counter_collection.csv")
    per = {}
    for k in sorted(set(f) | set(w)):
        fb, wb = f[k].get("FETCH_SIZE", 0.0) * 2048, w[k].get("WRITE_SIZE", 0.0) * 1024
        e = {"fetch_x2_bytes_per_sample": round(fb / samples_per_call, 3), "write_bytes_per_sample": round(wb / samples_per_call, 3),
             "kernel_ms": round(fms[k], 3)}
        if sqa and k in sqa and k in sqb and sqb[k].get("SQ_WAVE_CYCLES"):
            a_, b_ = sqa[k], sqb[k]
            wc = b_["SQ_WAVE_CYCLES"]
            e["valu_wave_instr_per_64_samples"] = round(a_["SQ_INSTS_VALU"] / (samples_per_call / 64), 2)
            e["salu_wave_instr_per_64_samples"] = round(a_["SQ_INSTS_SALU"] / (samples_per_call / 64), 2)
            e["lds_wave_instr_per_64_samples"] = round(a_["SQ_INSTS_LDS"] / (samples_per_call / 64), 2)
            e["lds_bank_conflict_cycle_frac"] = round(a_["SQ_LDS_BANK_CONFLICT"] / a_["SQ_LDS_IDX_ACTIVE"], 3) if a_.get("SQ_LDS_IDX_ACTIVE") else None
            e["wavefronts"] = int(a_["SQ_WAVES"])
            e["wave_time_split"] = {"issuing": round(b_["SQ_ACTIVE_INST_ANY"] / wc, 3), "issue_stalled": round(b_["SQ_WAIT_INST_ANY"] / wc, 3),
                                    "parked_on_waitcnt": round(b_["SQ_WAIT_ANY"] / wc, 3)}
            e["valu_share_of_issued_cycles"] = round(b_["SQ_ACTIVE_INST_VALU"] / b_["SQ_ACTIVE_INST_ANY"], 3)
            e["shader_clock_ghz"] = round(b_["GRBM_GUI_ACTIVE"] / N_XCD / (sqb_ms[k] * 1e6), 2)
        per[k] = e
    import hashlib
    data = open(os.path.join(ROOT, "f5c_amd/csrc/abea_kernels.hip"), "rb").read()
    sha = hashlib.sha256(data[data.find(ALIGN_SECTION_END):]).hexdigest()
    total = sum(v["fetch_x2_bytes_per_sample"] + v["write_bytes_per_sample"] for v in per.values())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    j = json.load(open(path))
    passes = {"fetch": prefix + "n2_fetch_counter_collection.csv", "write": prefix + "n2_write_counter_collection.csv"}
    if sqa:
        passes.update({"sq_a": prefix + "n2_sqa_counter_collection.csv", "sq_b": prefix + "n2_sqb_counter_collection.csv"})
    j["detector"] = {"kernels": "abea_ev_*", "samples_per_call": samples_per_call, "hbm_bytes_per_sample": round(total, 3),
                     "kernels_ms_per_call": round(sum(v["kernel_ms"] for v in per.values()), 3), "per_kernel": per,
                     "code_sha256": sha, "code_sha256_of": "f5c_amd/csrc/abea_kernels.hip from the event-detection banner to the end",
                     "passes": passes, "command": "tools/n2_profile.py 2048 (DNA parameters: the two calls in front of the first RNA kernel)"}
    json.dump(j, open(path, "w"), indent=1)
    print("detector", round(total, 2), "B per sample;", {k: round(v["fetch_x2_bytes_per_sample"] + v["write_bytes_per_sample"], 2) for k, v in per.items()})


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "detector":
        detector(sys.argv[2], int(sys.argv[3]))
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r03/a_")
