#!/usr/bin/env python3
"""Static checks of the hand-written gfx950 inline asm (f5c_amd/csrc/abea_fill.inc, abea_walk.inc).

Inside an inline-asm statement nobody inserts the wait states or the s_waitcnt the hardware needs: the generator
(tools/gen_fill_asm.py) places them "by construction".  This lint re-derives them from the generated text, over the
control-flow graph of the statement (labels, s_branch, s_cbranch_*), with a forward data-flow analysis to a fixed point:

  manual wait states of the MI300 / gfx940 ISA ("Manually inserted wait states", and LLVM's GCNHazardRecognizer):
    H1  VALU writes VGPR           -> DPP reads that VGPR                          >= 2
    H2  VALU writes VGPR           -> v_readlane / v_readfirstlane reads it        >= 1
    H3  VALU writes SGPR / VCC     -> VALU reads it as a constant / mask           >= 2
    H4  VALU writes SGPR           -> v_readlane / v_writelane lane select         >= 4
    H5  VALU writes SGPR           -> VMEM reads it (scalar address)               >= 5
    H6  SALU writes M0             -> LDS add-TID instruction                      >= 1
    H7  global store of > 64 bits  -> VALU overwrites its data VGPRs               >= 2
  definite assignment:
    U1  a write-only operand of the statement ("=&s" / "=&v" in abea_kernels.hip) or a clobbered fixed register is written
        on every path before it is read (DPP moves and v_writelane keep part of their destination: they read it too)
  clobbers:
    C1  every fixed register the statement writes is in its CLOBBERS macro or tied to an operand (m0 and exec are
        reserved: the statement restores exec and the compiler keeps nothing in m0 across an asm volatile)
  memory results (the counters return in order per class):
    W1  a register that an outstanding ds_read / global_load will write is neither read nor written before an
        s_waitcnt lgkmcnt(n) / vmcnt(n) that covers it (n <= the number of same-class operations issued after it)
    W2  the data / address registers of an outstanding ds_write / global_store ... are covered by H7 only (stores read
        their operands at issue, except H7's late data read)

"Wait states" are instructions issued in between (s_nop n counts n + 1).  At a join the analysis keeps the worst case
of the incoming paths (youngest write, fewest later memory operations), so a clean run holds on every path.

  python tools/asm_lint.py            # exit status 1 and a list of findings if anything is wrong
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = 8                       # ages saturate here (largest requirement is 5)

# operands of the two statements that the compiler binds to VGPRs (everything else named %[x] is an SGPR / SGPR pair)
VGPR_OPERANDS = {"Pf0", "Pf1", "x0", "x1", "g0", "c0", "g1", "c1", "nkg", "nkc", "nx", "e_pend", "kpg", "kpc", "a1", "a2", "a3",
                 "acc", "toff", "i0", "i1", "nki", "kpi", "L0", "L1", "U0", "U1", "lane", "o_cv",
                 "px0", "px1", "kag", "kac", "kai", "kbg", "kbc", "kbi"}      # the last row: the ABEA_FIFO experiment


def statement(path, macro):
    """The instruction lines of `#define <macro> "..." "..."` as a list of strings."""
    text = open(path).read()
    start = text.index("#define " + macro)
    lines = []
    for ln in text[start:].split("\n")[1:]:
        m = re.match(r'\s*"(.*)\\n\\t"', ln)
        if not m:
            break
        lines.append(m.group(1).strip())
    return lines


def regs_of(tok):
    """Registers named by one operand token: v12 -> [v12]; v[4:5] -> [v4, v5]; %[t0] -> [%t0]; vcc -> [vcc_lo, vcc_hi]."""
    tok = tok.strip()
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.fullmatch(r"[vs]\d+", tok):
        return [tok]
    m = re.fullmatch(r"%\[(\w+)\]", tok)
    if m:
        return ["%" + m.group(1)]
    if tok == "vcc":
        return ["vcc_lo", "vcc_hi"]
    if tok == "exec":
        return ["exec_lo", "exec_hi"]
    if tok in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "m0", "scc"):
        return [tok]
    return []                 # immediates, modifiers


def is_vgpr(r):
    return r.startswith("v") and r[1:].isdigit() or (r.startswith("%") and r[1:] in VGPR_OPERANDS)


def is_sgpr(r):
    return not is_vgpr(r) and r not in ("scc",)


class Ins:
    def __init__(self, text, idx):
        self.text, self.idx = text, idx
        parts = text.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        rest = re.split(r"\s+(?:wave_sh[lr]|wave_ro[lr]|row_|quad_perm|offset:|op_sel|neg_|clamp|bound_ctrl)", rest)[0]
        self.ops = [t for t in (x.strip() for x in rest.split(",")) if t]
        self.dpp = "_dpp" in self.op
        op = self.op
        self.kind = ("label" if text.endswith(":") else
                     "salu" if op.startswith("s_") else
                     "lds" if op.startswith("ds_") else
                     "vmem" if op.startswith("global_") else "valu")
        self.dst, self.src = [], []
        self.lane_select = []
        if self.kind == "label":
            return
        R = [regs_of(t) for t in self.ops]
        if self.kind == "salu":
            if op in ("s_branch", "s_nop", "s_waitcnt") or op.startswith("s_cbranch"):
                pass
            elif op.startswith("s_cmp") or op.startswith("s_bitcmp"):
                self.src = sum(R, [])
            else:
                self.dst = R[0] if R else []
                self.src = sum(R[1:], [])
                if op in ("s_cselect_b32", "s_cselect_b64", "s_addc_u32"):
                    self.src.append("scc")
        elif self.kind == "lds":
            if op.startswith("ds_read"):
                self.dst = R[0]
                self.src = sum(R[1:], [])
                if "addtid" in op:
                    self.src.append("m0")
            else:
                self.src = sum(R, [])
        elif self.kind == "vmem":
            if "load" in op:
                self.dst = R[0]
                self.src = sum(R[1:], [])
            else:
                self.src = sum(R, [])
                self.store_data = R[1] if len(R) > 1 else []
        else:  # valu
            if op.startswith("v_cmp"):
                self.dst = R[0]
                self.src = sum(R[1:], [])
            elif op in ("v_readlane_b32",):
                self.dst = R[0]; self.src = R[1]; self.lane_select = R[2] if len(R) > 2 else []
            elif op == "v_readfirstlane_b32":
                self.dst = R[0]; self.src = R[1]
            elif op == "v_writelane_b32":
                self.dst = R[0]; self.src = R[1]; self.lane_select = R[2] if len(R) > 2 else []
            else:
                self.dst = R[0] if R else []
                self.src = sum(R[1:], [])
                if op.startswith("v_addc") or op.startswith("v_subb"):
                    self.src += ["vcc_lo", "vcc_hi"]

    def wait_states(self):
        if self.op == "s_nop":
            return int(self.ops[0]) + 1
        return 1


def parse(lines):
    ins = [Ins(t, i) for i, t in enumerate(lines)]
    labels = {x.text[:-1]: i for i, x in enumerate(ins) if x.kind == "label"}
    return ins, labels


def successors(ins, labels, i):
    x = ins[i]
    out = []
    if x.kind != "label" and x.op == "s_branch":
        return [labels[x.ops[0]]]
    if x.kind != "label" and x.op.startswith("s_cbranch"):
        out.append(labels[x.ops[0]])
    if i + 1 < len(ins):
        out.append(i + 1)
    return out


def write_only_operands():
    """Names bound with "=&s" / "=&v" in the kernel source: undefined when the statement starts."""
    text = open(os.path.join(ROOT, "f5c_amd", "csrc", "abea_kernels.hip")).read()
    return {"%" + m for m in re.findall(r'\[(\w+)\]\s*"=&[sv]"', text)}


class State:
    """ages[r] = instructions issued since r was last written by (VALU, SALU); store[r] = since a >64-bit store named r
    as data; pend[(cls, r)] = memory operations of that class issued after the outstanding load that will write r."""
    __slots__ = ("valu", "salu", "store", "pend", "undef")

    def __init__(self, undef=()):
        self.valu, self.salu, self.store, self.pend = {}, {}, {}, {}
        self.undef = {r: 0 for r in undef}          # registers that may still be unwritten (a dict so that merge() is shared)

    def copy(self):
        s = State()
        s.valu, s.salu, s.store, s.pend = dict(self.valu), dict(self.salu), dict(self.store), dict(self.pend)
        s.undef = dict(self.undef)
        return s

    def merge(self, o):
        """Worst case of the two; True if self changed."""
        changed = False
        for mine, theirs in ((self.valu, o.valu), (self.salu, o.salu), (self.store, o.store), (self.pend, o.pend),
                             (self.undef, o.undef)):
            for k, v in theirs.items():
                if k not in mine or v < mine[k]:
                    mine[k] = v; changed = True
        return changed


def step(x, st, report):
    """Check x against st, then apply x."""
    if x.kind == "label":
        return
    # ---------------- checks
    def need(table, r, n, what):
        a = table.get(r, CAP)
        if a < n:
            report(x, f"{what}: {r} written {a} wait state(s) before, needs {n}")
    if x.kind == "valu":
        for r in x.src:
            if is_vgpr(r):
                if x.dpp:
                    need(st.valu, r, 2, "H1 VALU write -> DPP read")
                if x.op in ("v_readlane_b32", "v_readfirstlane_b32"):
                    need(st.valu, r, 1, "H2 VALU write -> v_readlane")
            elif r != "scc":
                need(st.valu, r, 2, "H3 VALU-written SGPR -> VALU read")
        for r in x.lane_select:
            need(st.valu, r, 4, "H4 VALU-written SGPR -> lane select")
        for r in x.dst:
            if is_vgpr(r):
                need(st.store, r, 2, "H7 >64-bit store -> VALU overwrites its data")
    if x.kind == "vmem":
        for r in x.src:
            if is_sgpr(r):
                need(st.valu, r, 5, "H5 VALU-written SGPR -> VMEM")
    if x.kind == "lds" and "addtid" in x.op:
        need(st.salu, "m0", 1, "H6 SALU write M0 -> LDS add-TID")
    if x.op != "s_waitcnt":
        for r in x.src + x.dst + x.lane_select:
            for cls in ("lgkm", "vm"):
                if (cls, r) in st.pend:
                    report(x, f"W1 {r} is the target of an outstanding {'LDS' if cls == 'lgkm' else 'global'} load "
                              f"({st.pend[(cls, r)]} later operation(s) of its class, no covering s_waitcnt)")
    reads = list(x.src) + list(x.lane_select)
    if (x.dpp and "wave_ro" not in x.text) or x.op == "v_writelane_b32":
        reads += x.dst                                     # lanes without a source / the other 63 lanes keep the old value
                                                           # (a wave rotation has a source for every lane)
    for r in reads:
        if r in st.undef:
            report(x, f"U1 {r} may be read before it is written")
    for r in x.dst:
        st.undef.pop(r, None)
    # ---------------- effects
    w = x.wait_states()
    for table in (st.valu, st.salu, st.store):
        for k in list(table):
            table[k] = min(CAP, table[k] + w)
            if table[k] >= CAP:
                del table[k]
    if x.op == "s_waitcnt":
        for cls, pat in (("lgkm", r"lgkmcnt\((\d+)\)"), ("vm", r"vmcnt\((\d+)\)")):
            m = re.search(pat, x.text)
            if m:
                n = int(m.group(1))
                for k in [k for k in st.pend if k[0] == cls and st.pend[k] >= n]:
                    del st.pend[k]
        return
    if x.kind in ("lds", "vmem"):
        cls = "lgkm" if x.kind == "lds" else "vm"
        for k in st.pend:
            if k[0] == cls:
                st.pend[k] += 1
        for r in x.dst:
            st.pend[(cls, r)] = 0
        if x.kind == "vmem" and "store" in x.op and len(getattr(x, "store_data", [])) > 2:
            for r in x.store_data:
                st.store[r] = 0
        return
    table = st.valu if x.kind == "valu" else st.salu
    other = st.salu if x.kind == "valu" else st.valu
    for r in x.dst:
        table[r] = 0
        other.pop(r, None)
    if x.kind == "valu" and x.op.startswith("v_cmp") is False and x.op in ("v_add_co_u32", "v_sub_co_u32"):
        st.valu["vcc_lo"] = st.valu["vcc_hi"] = 0


def lint(lines, name, undef=()):
    ins, labels = parse(lines)
    findings = {}

    def report(x, msg):
        findings.setdefault((x.idx, msg.split(":")[0]), f"{name}:{x.idx + 1}: `{x.text}`  {msg}")
    states = [None] * len(ins)
    states[0] = State(undef)
    work = [0]
    while work:
        i = work.pop()
        st = states[i].copy()
        step(ins[i], st, report)
        for j in successors(ins, labels, i):
            if states[j] is None:
                states[j] = st.copy(); work.append(j)
            elif states[j].merge(st):
                work.append(j)
    unreachable = [x for x, s in zip(ins, states) if s is None and x.kind != "label"]
    return list(findings.values()), len(ins), len(unreachable)


# Findings that hold on a path the data flow cannot rule out but the values do, each with its reason.
WAIVERS = {
    ("abea_walk.inc", "U1", "s_lshr_b64 s[94:95], s[90:91], s96"):
        "the first step of a group always branches to reload_lp_hi / _lo, which loads s[88:91]: s86 is set to -1 at group_top and a lane pair (0..49) never equals it",
}


def unclobbered_writes(path, macro):
    """C1: fixed registers written by the statement but neither clobbered nor tied."""
    text = open(path).read()
    m = re.search(r"#define " + macro.replace("_ASM", "_CLOBBERS") + r" (.*)", text)
    declared = set(re.findall(r'"(\w+)"', m.group(1))) if m else set()
    declared |= set(re.findall(r"\{(v\d+)\}", text))                       # tied operands of the experiment layouts
    for a, b in re.findall(r"\{v\[(\d+):(\d+)\]\}", text):
        declared |= {f"v{i}" for i in range(int(a), int(b) + 1)}
    if "vcc" in declared:
        declared |= {"vcc_lo", "vcc_hi"}
    declared |= {"m0", "exec_lo", "exec_hi"}
    bad = {}
    for x in parse(statement(path, macro))[0]:
        for r in x.dst:
            if not r.startswith("%") and r not in declared:
                bad.setdefault(r, f"{os.path.basename(path)}:{x.idx + 1}: `{x.text}`  C1 {r} is written but not clobbered")
    return list(bad.values())


def undefined_at_entry(path, macro):
    """Write-only operands + the fixed registers the statement clobbers (its CLOBBERS macro)."""
    text = open(path).read()
    m = re.search(r"#define " + macro.replace("_ASM", "_CLOBBERS") + r" (.*)", text)
    regs = set(re.findall(r'"([vs]\d+)"', m.group(1))) if m else set()
    return regs | write_only_operands()


def apply_waivers(found, fn):
    keep = [f for f in found if not any(w[0] == fn and f"  {w[1]} " in f and f"`{w[2]}`" in f for w in WAIVERS)]
    return keep, len(found) - len(keep)


def main():
    csrc = os.path.join(ROOT, "f5c_amd", "csrc")
    bad = 0
    for fn, macro in (("abea_fill.inc", "ABEA_FILL_ASM"), ("abea_walk.inc", "ABEA_WALK_ASM")):
        path = os.path.join(csrc, sys.argv[sys.argv.index("--suffix") + 1].join(os.path.splitext(fn))
                            if "--suffix" in sys.argv else fn)
        found, n, dead = lint(statement(path, macro), os.path.basename(path), undefined_at_entry(path, macro))
        found, waived = apply_waivers(found, fn)
        found += unclobbered_writes(path, macro)
        print(f"{fn}: {waived} finding(s) waived (see WAIVERS)") if waived else None
        print(f"{os.path.basename(path)}: {n} lines, {len(found)} finding(s), {dead} unreachable instruction(s)")
        for f in found[:40]:
            print("  " + f)
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
