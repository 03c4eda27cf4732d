"""The hand-written inline asm carries its own wait states and s_waitcnt (nothing inserts them inside an asm statement).
tools/asm_lint.py re-derives them over the statement's control-flow graph; here: the shipped statements are clean, and the
lint does find each kind of fault when one is planted."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_lint as L   # noqa: E402

CSRC = os.path.join(ROOT, "f5c_amd", "csrc")
DPP = "wave_shr:1 row_mask:0xf bank_mask:0xf"


def findings(lines):
    return L.lint(lines, "t")[0]


def test_shipped_statements_are_clean():
    for fn, macro, min_lines in (("abea_fill.inc", "ABEA_FILL_ASM", 8000), ("abea_walk.inc", "ABEA_WALK_ASM", 100)):
        path = os.path.join(CSRC, fn)
        found, n, dead = L.lint(L.statement(path, macro), fn, L.undefined_at_entry(path, macro))
        found, waived = L.apply_waivers(found, fn)
        assert waived <= 2, waived                          # the one documented value-dependent path (two registers of a pair)
        assert n >= min_lines, (fn, n)                      # the whole statement was parsed
        assert dead == 0, f"{fn}: {dead} unreachable instructions"
        assert not found, "\n".join(found[:10])
        assert not L.unclobbered_writes(path, macro)


def test_planted_wait_state_faults_are_found():
    cases = {
        "H1": ["v_mov_b32 v1, v2", "s_nop 0", f"v_mov_b32_dpp v3, v1 {DPP}"],                      # 1 wait state, needs 2
        "H2": ["v_max3_f32 v64, v1, v2, v3", "v_readlane_b32 %[t0], v64, 0"],
        "H3": ["v_readlane_b32 %[t0], v64, 0", "s_nop 0", "v_cmp_lt_f32 vcc, %[t0], v65"],
        "H4": ["v_readlane_b32 %[t1], v64, 0", "s_nop 2", "v_readlane_b32 %[t0], v65, %[t1]"],
        "H5": ["v_readfirstlane_b32 %[t1], v64", "s_nop 3", "global_load_dword v1, v2, %[t1]"],
        "H6": ["s_mov_b32 m0, %[e_addr]", "ds_read_addtid_b32 v94"],
        "H7": ["global_store_dwordx4 v1, v[106:109], %[trace]", "s_nop 0", "v_mov_b32 v107, v2"],
    }
    for rule, prog in cases.items():
        f = findings(prog)
        assert len(f) == 1 and rule in f[0], (rule, f)
    # the same programs with the required distance are clean
    ok = [
        ["v_mov_b32 v1, v2", "s_nop 1", f"v_mov_b32_dpp v3, v1 {DPP}"],
        ["v_max3_f32 v64, v1, v2, v3", "v_mov_b32 v9, v8", "v_readlane_b32 %[t0], v64, 0"],
        ["v_readlane_b32 %[t0], v64, 0", "s_nop 1", "v_cmp_lt_f32 vcc, %[t0], v65"],
        ["v_readlane_b32 %[t1], v64, 0", "s_nop 3", "v_readlane_b32 %[t0], v65, %[t1]"],
        ["v_readfirstlane_b32 %[t1], v64", "s_nop 4", "global_load_dword v1, v2, %[t1]"],
        ["s_mov_b32 m0, %[e_addr]", "v_mov_b32 v9, v8", "ds_read_addtid_b32 v94"],
        ["global_store_dwordx4 v1, v[106:109], %[trace]", "s_nop 1", "v_mov_b32 v107, v2"],
        ["global_store_dword v1, v106, %[trace]", "v_mov_b32 v106, v2"],                           # 32-bit store: no late data read
    ]
    for prog in ok:
        assert not findings(prog), prog


def test_planted_missing_waitcnt_is_found():
    assert findings(["ds_read_addtid_b32 v94", "v_mov_b32 v1, v94"])
    assert not findings(["ds_read_addtid_b32 v94", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v1, v94"])
    # counters return in order: with one younger read outstanding, lgkmcnt(1) covers the older one only
    prog = ["ds_read_addtid_b32 v94", "ds_read_addtid_b32 v82", "s_waitcnt lgkmcnt(1)"]
    assert not findings(prog + ["v_mov_b32 v1, v94"])
    assert findings(prog + ["v_mov_b32 v1, v82"])
    # overwriting the target of a load in flight is as wrong as reading it
    assert findings(["global_load_dword v95, v1, %[evm]", "v_mov_b32 v95, v2"])
    assert not findings(["global_load_dword v95, v1, %[evm]", "s_waitcnt vmcnt(0)", "v_mov_b32 v95, v2"])
    # LDS and global counters are separate
    assert findings(["global_load_dword v95, v1, %[evm]", "s_waitcnt lgkmcnt(0)", "v_mov_b32 v2, v95"])


def test_worst_path_wins_at_a_join():
    # the fault is only on the taken path: the DPP at `join` is two instructions behind the write on the fall-through path
    prog = ["s_cmp_eq_u32 %[b], 0", "s_cbranch_scc1 fast_%=", "v_mov_b32 v1, v2", "s_nop 1", "s_branch join_%=",
            "fast_%=:", "v_mov_b32 v1, v3", "join_%=:", f"v_mov_b32_dpp v3, v1 {DPP}"]
    f = findings(prog)
    assert len(f) == 1 and "H1" in f[0], f
    # a loop: the write at the bottom reaches the read at the top through the back edge
    loop = ["top_%=:", f"v_mov_b32_dpp v3, v1 {DPP}", "s_nop 4", "s_cmp_eq_u32 %[b], 0", "v_mov_b32 v1, v2", "s_cbranch_scc0 top_%="]
    f = findings(loop)
    assert len(f) == 1 and "H1" in f[0], f


def test_planted_read_before_write_is_found():
    und = {"%t1", "v110"}
    assert L.lint(["s_add_u32 %[t0], %[t1], 1"], "t", und)[0]
    assert not L.lint(["s_mov_b32 %[t1], 0", "s_add_u32 %[t0], %[t1], 1"], "t", und)[0]
    # written on one path only
    prog = ["s_cmp_eq_u32 %[b], 0", "s_cbranch_scc1 skip_%=", "v_mov_b32 v110, v1", "skip_%=:", "v_add_f32 v2, v110, v110"]
    f = L.lint(prog, "t", und)[0]
    assert len(f) == 1 and "U1" in f[0], f
    # a DPP move keeps the lanes without a source: it reads its destination
    assert L.lint([f"v_mov_b32_dpp v110, v1 {DPP}"], "t", und)[0]
    assert len(L.write_only_operands()) >= 15               # the "=&s" / "=&v" operands were found in the kernel source
