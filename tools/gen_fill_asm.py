#!/usr/bin/env python3
"""Generate f5c_amd/csrc/abea_fill_interior.inc: the hand-scheduled gfx950 band-fill loop for the
interior stretch of a read (every band cell in range, no trim column, no end-scan), as ONE inline-asm
statement.  Run:  python tools/gen_fill_asm.py

Why asm: the loop is issue-bound; hipcc's version carries ~25 phi-copy v_movs, a 10-branch decision
diamond and conservative vmcnt(0) waits per band.  This version keeps the loop state in
fixed VGPRs v64..v127, is unrolled over (band parity, previous move, this move) = 8 straight-line bodies so
that the diagonal / up / left operands are pure register NAMES (no copies), and places its own waits.

Score state per band: the float scores mf0,mf1 of the previous band, and a "triple" of exact doubles
  c0 = (double)mf0, c1 = (double)mf1, cs = (double)shift(mf)   (shift = lane+1's mf0 for a right move,
                                                                lane-1's mf1 for a down move)
  right move: left = (c0,c1)  up = (c1,cs);   down move: left = (cs,c0)  up = (c0,c1)
  diagonal of this band = up (right move) / left (down move) of the PREVIOUS band's triple.
Two triples ping-pong with band parity.

Hazards honoured by construction (gfx940/950, no assembler help inside inline asm):
  VALU write VGPR -> v_readlane of it: >=1 wait state;  -> DPP read of it: >=2
  VALU write SGPR/VCC -> VALU read of it (v_cndmask mask, v_cmp operand): >=2
  global_store_dwordx4 data registers are private copies (v106..v109), rewritten only 32 bands later
"""
import os

# Round 4 settled the candidates of round 3 on the GPU (profiles/r04/candidates_ab.txt, >= 10 launches each on configs[1] and
# [2], every output bit compared): the three that held outside the noise are now the design and their switches are gone —
#   * no LDS rings: the idle lanes 52..63 hold the NEXT 24 events (in the event registers themselves: a down move shifts them
#     with wave_ror, so lane 63's cell 1 arrives in lane 0) and the next 24 k-mers (offsets 104..127); every 24th move of a
#     kind overwrites those twelve lanes under an EXEC mask from registers that a global load filled 24 moves earlier.  Per
#     band no ring read, no M0 set-up, no lgkmcnt wait: -3 instructions on a down move, -6 on a right move; the refill
#     trigger is a two-instruction countdown;
#   * the move decision of the next band (v_readlane of the lower-left score, v_cmp with the upper-right one) is issued right
#     after the two v_max3 instead of behind the trace packing: the scalar test and branch at the end of the band no longer
#     wait for the compare;
#   * the traceback walk's step loop exists twice, once for the upper 16 bands of a trace group (from-codes in s[90:91]) and
#     once for the lower 16 (s[88:89]), so the half is not selected per step; the band-move word is kept shifted to the
#     current band; the gap counter is reset with a multiply; the lower-left corner is updated through a popcount
#     (46.9 -> 42.6 scalar instructions per step).
# Measured and dropped (their generator paths are deleted; the records are in profiles/r03/experiments/ and DESIGN.md §4.3):
# packed f32 (v_pk_mul/add/mov: fewer instructions, same time), the emission-chain interleave for a lone wave (-0.6 % on a
# 512-read batch), the loop state tied to physical registers for 5-7 waves per SIMD, the LDS rings, the one-loop walk.
VB = 64            # first fixed VGPR (v64..v127)
assert VB % 2 == 0
MF0, MF1, SHR, SHD = VB + 0, VB + 1, VB + 2, VB + 3
TR = [dict(c0=VB + 4, c1=VB + 6, cs=VB + 8), dict(c0=VB + 10, c1=VB + 12, cs=VB + 14)]
X0, X1 = VB + 16, VB + 17
# k-mer parameter quads {gpm, ck, istd lo, istd hi} (the layout of abea_kpar_t, so ds_read_b128 lands one whole):
# the roles "cell 0", "cell 1", "incoming k-mer" ROTATE over the three quads with every right move (state rs =
# number of right moves mod 3: cell 0 = KQ[rs], cell 1 = KQ[rs+1], incoming = KQ[rs+2]), so a right move is four DPP
# shifts into the incoming quad and no register-to-register moves; rs = 0 is the layout at entry / exit.
KQ = [VB + 18, VB + 22, VB + 26]
G0, C0, I0 = KQ[0], KQ[0] + 1, KQ[0] + 2
G1, C1, I1 = KQ[1], KQ[1] + 1, KQ[1] + 2
NK = KQ[2]         # v[90:93] = incoming k-mer {gpm, ck, istd} at rs = 0
NX = VB + 30
KPEND = VB + 32    # v[96:99]: pending k-mer quad of the FIFO lanes' cell 0
A0, A1, A2, ACC = VB + 36, VB + 37, VB + 38, VB + 39
NINF = VB + 40
LANE = VB + 41
LPD = [VB + 42, VB + 44]
TD = [VB + 46, VB + 48]    # per-cell f64 temps
TU = [VB + 50, VB + 52]
Q = VB + 42        # store quad v[106:109] aliases the LPD temps (free at the end of a band)
TMP = VB + 54      # 32-bit address temp
TOFF = VB + 55     # trace store offset (lane*16 + group*1024)
F = [VB + 47, VB + 49]     # from codes: high halves of the TD pairs (free once sd is rounded)
O0, O1 = VB + 56, VB + 57   # band offsets owned by the lane (border variant only)
PX = VB + 58           # v[122:123]: pending events of the FIFO lanes (-> X1, X0)
PKA, PKB = KPEND, VB + 60   # v[96:99], v[124:127]: pending k-mer quads of the FIFO lanes' cell 0 / cell 1
VEND = VB + 64
FIFO_EXEC_HI = "0xFFF00000"   # lanes 52..63
DPP_ROR = "wave_ror:1 row_mask:0xf bank_mask:0xf"
BORDER = False     # generator mode: True adds validity masks + the online end-point scan
# extra per-cell 32-bit temps reuse the low halves of f64 temps where noted

out = []
lbl_id = [0]


def emit(s):
    out.append(s)


CHECK_ONLY = "--check" in __import__("sys").argv     # compare with the committed .inc files instead of writing them
stale = []


def _emit_file(path, text, n_lines):
    if CHECK_ONLY:
        same = os.path.exists(path) and open(path).read() == text
        print(f"{'up to date' if same else 'STALE'}: {path}")
        if not same:
            stale.append(path)
    else:
        open(path, "w").write(text)
        print(f"wrote {path}: {n_lines} asm lines")


def v(n):
    return f"v{n}"


def vp(n):
    return f"v[{n}:{n+1}]"


def vq(n):
    return f"v[{n}:{n+3}]"


def newid():
    lbl_id[0] += 1
    return lbl_id[0]


DPP_SHL = "wave_shl:1 row_mask:0xf bank_mask:0xf"
DPP_SHR = "wave_shr:1 row_mask:0xf bank_mask:0xf"


def cell_ops(j, D, U, L, quad):
    """Instruction list for cell j (0/1); D,U,L = first register of the f64 pairs; quad = the cell's k-mer quad."""
    x = X0 + j
    g, ck, i = quad, quad + 1, quad + 2
    lpd, td, tu = LPD[j], TD[j], TU[j]
    t32 = td          # 32-bit scratch: low half of td before td is used
    sd, su, sl = td, tu, lpd   # after the adds the pair's low dword holds the float result (cvt in place)
    cm1, cm2 = f"%[cm{j}a]", f"%[cm{j}b]"
    mf = MF0 + j
    ops = [
        f"v_sub_f32 {v(t32)}, {v(x)}, {v(g)}",
        f"v_cvt_f64_f32 {vp(lpd)}, {v(t32)}",
        f"v_mul_f64 {vp(lpd)}, {vp(lpd)}, {vp(i)}",
        f"v_cvt_f32_f64 {v(t32)}, {vp(lpd)}",
        # lp = ck + (-0.5f*a)*a (align.c:113) in two instructions: halving is exact, so RN((-0.5a)*a) = -0.5*RN(a*a), and the
        # fma adds that exact product to ck with the ONE rounding the reference's add performs.  (Differs from the three-
        # instruction form only when a*a underflows AND ck == 0: by one subnormal ulp, 1e-45, of a term that is added to
        # scores in fp64 — DESIGN.md §4.3.)
        f"v_mul_f32 {v(tu)}, {v(t32)}, {v(t32)}",
        f"v_fma_f32 {v(tu)}, -0.5, {v(tu)}, {v(ck)}",
        f"v_cvt_f64_f32 {vp(lpd)}, {v(tu)}",
        f"v_add_f64 {vp(td)}, {vp(D)}, %[lp_step]",
        f"v_add_f64 {vp(tu)}, {vp(U)}, %[lp_stay]",
        f"v_add_f64 {vp(td)}, {vp(td)}, {vp(lpd)}",
        f"v_add_f64 {vp(tu)}, {vp(tu)}, {vp(lpd)}",
        f"v_add_f64 {vp(lpd)}, {vp(L)}, %[lp_skip]",
        f"v_cvt_f32_f64 {v(sd)}, {vp(td)}",
        f"v_cvt_f32_f64 {v(su)}, {vp(tu)}",
        f"v_cvt_f32_f64 {v(sl)}, {vp(lpd)}",
    ]
    if BORDER:
        ops += [
            f"v_max3_f32 {v(F[j])}, {v(sd)}, {v(su)}, {v(sl)}",          # F[j] temporarily holds the max
            f"v_cmp_ge_f32 {cm1}, {v(su)}, {v(sd)}",
            f"v_cmp_eq_f32 {cm2}, {v(sl)}, {v(F[j])}",
            # >= 2 wait states before the masks are read: the mf write and a nop-equivalent come first
            f"v_cndmask_b32 {v(mf)}, {v(NINF)}, {v(F[j])}, %[cv{j}]",     # out-of-matrix cells -> -inf
            "s_nop 0",
            f"v_cndmask_b32 {v(F[j])}, 0, 1, {cm1}",
            f"v_cndmask_b32 {v(F[j])}, {v(F[j])}, 2, {cm2}",
        ]
    else:
        # Interior: the two tie-break facts are SIGN BITS of float differences (x - x = +0, (-inf) - (-inf) = +NaN on
        # gfx950, so "not less" covers equality and the all--inf cell exactly like align.c:386-392):
        #   sign(su - sd) = [su < sd]   -> FROM_U loses to FROM_D      sign(sl - max) = [sl < max] -> FROM_L loses
        # v_alignbit shifts them straight into the trace accumulator (tail of body()); no masks, no selects.
        ops += [
            f"v_max3_f32 {v(mf)}, {v(sd)}, {v(su)}, {v(sl)}",
            f"v_sub_f32 {v(F[j])}, {v(su)}, {v(sd)}",
            f"v_sub_f32 {v(sd)}, {v(sl)}, {v(mf)}",
        ]
    return ops


def interleave(a, b):
    """Alternate two independent instruction streams (fills dependent-issue and hazard slots)."""
    res = []
    for i in range(max(len(a), len(b))):
        if i < len(a):
            res.append(a[i])
        if i < len(b):
            res.append(b[i])
    return res


def countdown():
    """%[per] = bands to the next tick = min(8 - (b & 7), b_end - b); %[cnt] = 1 << (32 - per): a sentinel bit that the
    per-band `s_lshl1_add_u32 cnt, cnt, move` pushes out (carry -> SCC) at the per-th band, with the moves of the bands
    since the tick collecting below it (1 = right).  One instruction per band counts AND records the move.  Needs b < b_end."""
    emit("s_and_b32 %[t1], %[b], 7")
    emit("s_sub_u32 %[t1], 8, %[t1]")
    emit("s_sub_u32 %[per], %[b_end], %[b]")
    emit("s_min_u32 %[per], %[per], %[t1]")
    emit("s_sub_u32 %[t1], 32, %[per]")
    emit("s_lshl_b32 %[cnt], 1, %[t1]")


def decide(p_next, m_last, rs, entry=False):
    """Tail of a band (or the entry stub): pick the move of the NEXT band and jump to its body.
    Expects: %[t0] = readlane(mf0, lane 0) already issued, vcc = (t0 < mf1) per lane already issued."""
    tag = f"{p_next}{m_last}"
    if entry:                                              # inside the loop the end-of-run test lives in the tick
        emit("s_cmp_eq_u32 %[b], %[b_end]")
        emit(f"s_cbranch_scc1 exit_{tag}{rs}_%=")
        countdown()
    emit("s_bitcmp1_b32 vcc_hi, 17")                     # lane 49: ll < ur  -> right (align.c:313); -inf < finite too
    emit(f"s_cbranch_scc1 body_{tag}R{rs}_%=")
    emit("s_cmp_eq_u32 %[t0], 0xff800000")               # not (ll < ur): both may be -inf
    emit(f"s_cbranch_scc1 llinf_{tag}{rs}_{newid()}_%=")
    lbl = f"llinf_{tag}{rs}_{lbl_id[0]}_%="
    emit(f"s_branch body_{tag}D{rs}_%=")
    return lbl


def llinf_block(lbl, p_next, m_last, rs):
    tag = f"{p_next}{m_last}"
    emit(f"{lbl}:")
    emit(f"v_readlane_b32 %[t1], {v(MF1)}, 49")
    emit("s_cmp_eq_u32 %[t1], 0xff800000")
    emit(f"s_cbranch_scc0 body_{tag}R{rs}_%=")           # ll = -inf < finite ur
    emit("s_flbit_i32_b32 %[t1], %[cnt]")                 # leading zeros of the sentinel = per - 1 - bands done since the tick
    emit("s_sub_u32 %[t1], %[per], %[t1]")
    emit("s_add_u32 %[t1], %[t1], %[b]")
    emit("s_sub_u32 %[t1], %[t1], 1")                     # index of the band about to be computed
    emit("s_bitcmp1_b32 %[t1], 0")                        # both -inf: alternate, right on odd bands (align.c:311)
    emit(f"s_cbranch_scc1 body_{tag}R{rs}_%=")
    emit(f"s_branch body_{tag}D{rs}_%=")


def body(p, ml, m, rs):
    """Band body for parity p, previous move ml, this move m ('R'/'D'), k-mer quad rotation rs at entry."""
    T, Tp = TR[p], TR[p ^ 1]
    tag = f"{p}{ml}{m}{rs}"
    emit(f"body_{tag}_%=:")
    if m == 'R':
        c0q, inq = KQ[rs], KQ[(rs + 2) % 3]
        rs = (rs + 1) % 3                                    # roles after the move: cell 0 = old cell 1, cell 1 = old incoming
        emit("s_sub_u32 %[k_cnt], %[k_cnt], 1")             # before ll_k moves: the refill wants the frame of the previous band
        emit(f"s_cbranch_scc1 krefill_{tag}_%=")
        emit(f"kcont_{tag}_%=:")
        emit("s_add_u32 %[ll_k], %[ll_k], 1")
        emit(f"v_mov_b32_dpp {v(SHR)}, {v(MF0)} {DPP_SHL}")
        # lanes >= 50 are the k-mer FIFO and their "scores" are garbage; the only one a band cell ever reads is lane 50's
        # slot 0 = offset 100, through this shift into lane 49: pin THAT to -inf (the band ends at offset 99).  Round 2
        # masked slot 0 of every band with a v_cndmask; a down move never looks at it.
        emit(f"v_writelane_b32 {v(SHR)}, %[ninf], 49")
        # every offset takes the k-mer of the offset above: cell 0's quad slides down one lane INTO the incoming quad (its
        # lane 63 keeps what it has: nothing reads offsets beyond the FIFO lanes); cell 1's quad becomes cell 0's by renaming
        for j in range(4):
            emit(f"v_mov_b32_dpp {v(inq + j)}, {v(c0q + j)} {DPP_SHL}")
        emit(f"v_cvt_f64_f32 {vp(T['c0'])}, {v(MF0)}")
        sh = SHR
        U = (T['c1'], T['cs']); L = (T['c0'], T['c1'])
        D = (Tp['c1'], Tp['cs']) if ml == 'R' else (Tp['c0'], Tp['c1'])
    else:
        if BORDER:                                          # the border masks need ll_e every band; the interior loop brings
            emit("s_add_u32 %[ll_e], %[ll_e], 1")           # it up to date at the tick (popcount of the recorded moves)
        emit(f"v_mov_b32_dpp {v(SHD)}, {v(MF1)} {DPP_SHR}")
        emit("s_sub_u32 %[e_cnt], %[e_cnt], 1")
        emit(f"s_cbranch_scc1 erefill_{tag}_%=")
        emit(f"econt_{tag}_%=:")
        emit(f"v_mov_b32_dpp {v(NX)}, {v(X1)} {DPP_ROR}")         # lane 0 <- lane 63's cell 1: the next event
        emit(f"v_mov_b32 {v(X1)}, {v(X0)}")
        emit(f"v_mov_b32 {v(X0)}, {v(NX)}")
        emit(f"v_cvt_f64_f32 {vp(T['c0'])}, {v(MF0)}")
        sh = SHD
        U = (T['c0'], T['c1']); L = (T['cs'], T['c0'])
        D = (Tp['c0'], Tp['c1']) if ml == 'R' else (Tp['cs'], Tp['c0'])
    # exact doubles of the previous band's scores (c0 issued above)
    emit(f"v_cvt_f64_f32 {vp(T['c1'])}, {v(MF1)}")
    emit(f"v_cvt_f64_f32 {vp(T['cs'])}, {v(sh)}")
    if BORDER:
        # in-matrix offsets [min_off, max_off) (align.c:337-346)
        emit("s_sub_u32 %[t2], %[ll_e], %[Em1]")
        emit("s_sub_u32 %[t3], 0, %[ll_k]")
        emit("s_max_i32 %[t2], %[t2], %[t3]")
        emit("s_max_i32 %[t2], %[t2], 0")                   # min_off = max(-ll_k, ll_e-(E-1), 0)
        emit("s_sub_u32 %[t3], %[Km1], %[ll_k]")
        emit("s_add_u32 %[t3], %[t3], 1")
        emit("s_add_u32 %[t4], %[ll_e], 1")
        emit("s_min_i32 %[t3], %[t3], %[t4]")
        emit("s_min_i32 %[t3], %[t3], 100")                 # max_off = min(K-ll_k, ll_e+1, 100)
        emit("s_sub_u32 %[t3], %[t3], %[t2]")
        emit("s_max_i32 %[t3], %[t3], 0")                   # width
        emit(f"v_subrev_u32 {v(F[0])}, %[t2], {v(O0)}")
        emit(f"v_subrev_u32 {v(F[1])}, %[t2], {v(O1)}")
        emit(f"v_cmp_gt_u32 %[cv0], %[t3], {v(F[0])}")
        emit(f"v_cmp_gt_u32 %[cv1], %[t3], {v(F[1])}")
    ops0 = cell_ops(0, D[0], U[0], L[0], KQ[rs]); ops1 = cell_ops(1, D[1], U[1], L[1], KQ[(rs + 1) % 3])
    if BORDER:
        # split off the two trailing from-code selects of each cell (and the s_nop before them)
        tail0, tail1 = ops0[-2:], ops1[-2:]
        for ins in interleave(ops0[:-3], ops1[:-3]):
            emit(ins)
    else:
        # the next band's move decision rides in the tail of the cells: issued as soon as the two new scores exist
        il = interleave(ops0, ops1)
        for ins in il[:-4]:                                  # ... v_max3 cell 0, v_max3 cell 1
            emit(ins)
        emit(f"v_readlane_b32 %[t0], {v(MF0)}, 0")          # mf0 written two instructions ago
        emit(il[-4]); emit(il[-3])                          # the two [su < sd] differences
        emit(f"v_cmp_lt_f32 vcc, %[t0], {v(MF1)}")          # t0 written three instructions ago
        emit(il[-2]); emit(il[-1])                          # the two [sl < max] differences
    if BORDER:
        emit("s_nop 0")
        emit(tail0[0]); emit(tail1[0]); emit(tail0[1]); emit(tail1[1])
        emit(f"v_cndmask_b32 {v(F[0])}, 0, {v(F[0])}, %[cv0]")     # trace stays 0 outside the matrix (align.c:257)
        emit(f"v_cndmask_b32 {v(F[1])}, 0, {v(F[1])}, %[cv1]")
        # ---- trim column, k-mer -1 (align.c:324-333): score lp_trim*(event+1), FROM_U
        nt = f"notrim_{tag}_%="
        emit("s_not_b32 %[t2], %[ll_k]")                    # offset of k-mer -1 = -1 - ll_k
        emit("s_cmp_lt_u32 %[t2], 100")
        emit(f"s_cbranch_scc0 {nt}")
        emit("s_sub_u32 %[t3], %[ll_e], %[t2]")             # its event
        emit("s_cmp_le_u32 %[t3], %[Em1]")
        emit(f"s_cbranch_scc0 {nt}")
        emit("s_add_u32 %[t3], %[t3], 1")
        emit(f"v_cvt_f64_i32 {vp(LPD[0])}, %[t3]")
        emit(f"v_cmp_eq_u32 %[cm0a], %[t2], {v(O0)}")
        emit(f"v_cmp_eq_u32 %[cm0b], %[t2], {v(O1)}")
        emit(f"v_mul_f64 {vp(LPD[0])}, {vp(LPD[0])}, %[lp_trim]")
        emit(f"v_cvt_f32_f64 {v(TMP)}, {vp(LPD[0])}")
        emit(f"v_cndmask_b32 {v(MF0)}, {v(MF0)}, {v(TMP)}, %[cm0a]")
        emit(f"v_cndmask_b32 {v(MF1)}, {v(MF1)}, {v(TMP)}, %[cm0b]")
        emit(f"v_cndmask_b32 {v(F[0])}, {v(F[0])}, 1, %[cm0a]")
        emit(f"v_cndmask_b32 {v(F[1])}, {v(F[1])}, 1, %[cm0b]")
        emit(f"{nt}:")
        # ---- online end-point scan (align.c:424-445): cell of the last k-mer, if in band and in range
        es = f"es_{tag}_%="
        emit("s_sub_u32 %[t2], %[Km1], %[ll_k]")            # offset of k-mer K-1
        emit("s_cmp_lt_u32 %[t2], 100")
        emit(f"s_cbranch_scc0 {es}")
        emit("s_sub_u32 %[t3], %[ll_e], %[t2]")             # its event
        emit("s_cmp_le_u32 %[t3], %[Em1]")
        emit(f"s_cbranch_scc0 {es}")
        emit("s_lshr_b32 %[t1], %[t2], 1")
        emit(f"v_readlane_b32 %[t0], {v(MF0)}, %[t1]")
        emit(f"v_readlane_b32 %[t4], {v(MF1)}, %[t1]")
        emit("s_bitcmp1_b32 %[t2], 0")
        emit("s_cselect_b32 %[t0], %[t4], %[t0]")
        emit("s_sub_u32 %[t4], %[Em1], %[t3]")
        emit("s_add_u32 %[t4], %[t4], 1")                   # E - e
        emit(f"v_cvt_f64_u32 {vp(LPD[0])}, %[t4]")
        emit(f"v_cvt_f64_f32 {vp(LPD[1])}, %[t0]")
        emit(f"v_mul_f64 {vp(LPD[0])}, {vp(LPD[0])}, %[lp_trim]")
        emit(f"v_add_f64 {vp(LPD[0])}, {vp(LPD[1])}, {vp(LPD[0])}")
        emit(f"v_cvt_f32_f64 {v(TMP)}, {vp(LPD[0])}")
        emit(f"v_mov_b32 {v(LPD[1])}, %[best]")
        emit("s_nop 0")
        emit(f"v_cmp_gt_f32 vcc, {v(TMP)}, {v(LPD[1])}")    # strict >: keeps the first maximum (align.c:440)
        emit("s_and_b64 %[cm0a], vcc, exec")
        emit(f"s_cbranch_scc0 {es}")
        emit(f"v_readfirstlane_b32 %[best], {v(TMP)}")
        emit("s_mov_b32 %[best_e], %[t3]")
        emit("s_mov_b32 %[best_llk], %[ll_k]")
        emit(f"{es}:")
        emit(f"v_readlane_b32 %[t0], {v(MF0)}, 0")
        emit("s_nop 1")
        emit(f"v_cmp_lt_f32 vcc, %[t0], {v(MF1)}")
        # trace nibble f0 | f1 << 2, stored inverted: the accumulator is complemented when its dword completes
        emit(f"v_lshl_or_b32 {v(F[0])}, {v(F[1])}, 2, {v(F[0])}")
        emit(f"v_xor_b32 {v(F[0])}, 15, {v(F[0])}")
        emit(f"v_lshl_or_b32 {v(ACC)}, {v(ACC)}, 4, {v(F[0])}")
    else:
        # trace bits, oldest first: cell 1 [sl<max], cell 1 [su<sd], cell 0 [sl<max], cell 0 [su<sd]; each v_alignbit
        # is acc = acc << 1 | sign(difference).  Complemented per dword they read f = 2*[sl==max] + [su>=sd]:
        # 0 FROM_D, 1 FROM_U, 2 or 3 FROM_L (align.c:386-392 priority).
        b_l1, b_u1, b_l0, b_u0 = TD[1], F[1], TD[0], F[0]
        emit(f"v_alignbit_b32 {v(ACC)}, {v(ACC)}, {v(b_l1)}, 31")
        emit(f"v_alignbit_b32 {v(ACC)}, {v(ACC)}, {v(b_u1)}, 31")
        emit(f"v_alignbit_b32 {v(ACC)}, {v(ACC)}, {v(b_l0)}, 31")
        emit(f"v_alignbit_b32 {v(ACC)}, {v(ACC)}, {v(b_u0)}, 31")
    # %[cnt] counts the bands to the next "tick" (a trace dword completes every 8th band; the run ends at b_end): the
    # band index itself is only brought up to date there
    emit(f"s_lshl1_add_u32 %[cnt], %[cnt], {1 if m == 'R' else 0}")
    emit(f"s_cbranch_scc1 rot_{tag}_%=")                 # the sentinel fell out: this was the last band before the tick
    emit(f"rotret_{tag}_%=:")
    lbl = decide(p ^ 1, m, rs)
    llinf_block(lbl, p ^ 1, m, rs)
    # ---- out-of-line tick: band index, dword rotation / group store, end of run, next countdown
    emit(f"rot_{tag}_%=:")
    emit("s_lshl_b32 %[mvacc], %[mvacc], %[per]")         # band-move bits, oldest band in the top bit: append the last `per`
    emit("s_or_b32 %[mvacc], %[mvacc], %[cnt]")
    if not BORDER:
        emit("s_bcnt1_i32_b32 %[t1], %[cnt]")             # right moves among them; the others advanced ll_e
        emit("s_sub_u32 %[t1], %[per], %[t1]")
        emit("s_add_u32 %[ll_e], %[ll_e], %[t1]")
    emit("s_add_u32 %[b], %[b], %[per]")
    emit("s_and_b32 %[t1], %[b], 7")
    emit(f"s_cbranch_scc1 norot_{tag}_%=")               # not on a dword boundary: the run ends here
    emit("s_and_b32 %[t1], %[b], 31")
    emit(f"s_cbranch_scc1 nostore_{tag}_%=")             # group complete only when the new b is a multiple of 32
    emit(f"v_mov_b32 {v(Q)}, %[mvacc]")
    emit(f"v_mov_b32 {v(Q+1)}, %[mvprev]")
    emit(f"v_mov_b32 {v(Q+2)}, {v(A2)}")
    emit(f"v_not_b32 {v(Q+3)}, {v(ACC)}")
    emit(f"v_cndmask_b32 {v(Q)}, {v(A0)}, {v(Q)}, %[m50]")
    emit(f"v_cndmask_b32 {v(Q+1)}, {v(A1)}, {v(Q+1)}, %[m50]")
    emit("s_nop 1")
    emit("s_mov_b32 exec_hi, 0x7FFFF")                    # lanes 0..50 only: 25.5 B per band instead of 32 (lanes 51..63 hold nothing)
    emit(f"global_store_dwordx4 {v(TOFF)}, {vq(Q)}, %[trace]")
    emit("s_mov_b32 exec_hi, -1")
    emit("s_nop 0")                                       # store data registers are rewritten by the next band
    emit(f"v_add_u32 {v(TOFF)}, 0x400, {v(TOFF)}")
    emit("s_mov_b32 %[mvprev], %[mvacc]")
    emit(f"nostore_{tag}_%=:")
    emit(f"v_mov_b32 {v(A0)}, {v(A1)}")
    emit(f"v_mov_b32 {v(A1)}, {v(A2)}")
    emit(f"v_not_b32 {v(A2)}, {v(ACC)}")                   # the accumulator collects complemented bits
    emit(f"norot_{tag}_%=:")
    emit("s_cmp_eq_u32 %[b], %[b_end]")
    emit(f"s_cbranch_scc1 exit_{p ^ 1}{m}{rs}_%=")
    countdown()
    emit(f"s_branch rotret_{tag}_%=")
    # ---- out-of-line: the FIFO lanes are topped up
    if m == 'R':
        c0q_in, c1q_in = KQ[(rs + 2) % 3], KQ[rs]             # rs is already the state AFTER the move: cell 0 / cell 1 quads before it
        emit(f"krefill_{tag}_%=:")                       # 24 right moves since the last one: lanes 52..63 hold nothing useful
        emit("s_waitcnt vmcnt(0)")
        emit("s_mov_b32 exec_lo, 0")
        emit(f"s_mov_b32 exec_hi, {FIFO_EXEC_HI}")
        for q, pk in ((c0q_in, PKA), (c1q_in, PKB)):        # offsets 2*lane and 2*lane + 1 of the frame before this move
            emit(f"v_mov_b64 {vp(q)}, {vp(pk)}")
            emit(f"v_mov_b64 {vp(q + 2)}, {vp(pk + 2)}")
        emit("s_add_u32 %[t0], %[ll_k], 24")              # the frame 24 right moves from now
        emit(f"v_lshl_add_u32 {v(TMP)}, {v(LANE)}, 1, %[t0]")
        emit(f"v_add_u32 {v(NX)}, 1, {v(TMP)}")
        emit(f"v_min_i32 {v(TMP)}, %[Km1], {v(TMP)}")
        emit(f"v_min_i32 {v(NX)}, %[Km1], {v(NX)}")
        emit(f"v_lshlrev_b32 {v(TMP)}, 4, {v(TMP)}")
        emit(f"v_lshlrev_b32 {v(NX)}, 4, {v(NX)}")
        emit(f"global_load_dwordx4 {vq(PKA)}, {v(TMP)}, %[kpar]")
        emit(f"global_load_dwordx4 {vq(PKB)}, {v(NX)}, %[kpar]")
        emit("s_mov_b64 exec, -1")
        emit("s_mov_b32 %[k_cnt], 23")
        emit("s_nop 1")
        emit(f"s_branch kcont_{tag}_%=")
    else:
        emit(f"erefill_{tag}_%=:")                       # 24 down moves since the last one
        emit("s_waitcnt vmcnt(0)")
        if BORDER:
            emit("s_mov_b32 %[t0], %[ll_e]")                # already moved: the event entering with this band
        else:
            emit("s_flbit_i32_b32 %[t0], %[cnt]")
            emit("s_sub_u32 %[t0], %[per], %[t0]")          # done + 1
            emit("s_bcnt1_i32_b32 %[t1], %[cnt]")           # rights + 1
            emit("s_sub_u32 %[t0], %[t0], %[t1]")           # downs since the tick, this band not included
            emit("s_add_u32 %[t0], %[t0], %[ll_e]")
            emit("s_add_u32 %[t0], %[t0], 1")               # the event entering with this band
        emit("s_mov_b32 exec_lo, 0")
        emit(f"s_mov_b32 exec_hi, {FIFO_EXEC_HI}")
        emit(f"v_mov_b32 {v(X1)}, {v(PX)}")                 # lane 63: events t0, t0+1 (cell 1, cell 0); lane 62: t0+2, t0+3; ...
        emit(f"v_mov_b32 {v(X0)}, {v(PX + 1)}")
        emit("s_add_u32 %[t0], %[t0], 24")
        emit(f"v_sub_u32 {v(TMP)}, 63, {v(LANE)}")
        emit(f"v_lshl_add_u32 {v(TMP)}, {v(TMP)}, 1, %[t0]")
        emit(f"v_add_u32 {v(NX)}, 1, {v(TMP)}")
        emit(f"v_min_i32 {v(TMP)}, %[Em1], {v(TMP)}")
        emit(f"v_min_i32 {v(NX)}, %[Em1], {v(NX)}")
        emit(f"v_lshlrev_b32 {v(TMP)}, 2, {v(TMP)}")
        emit(f"v_lshlrev_b32 {v(NX)}, 2, {v(NX)}")
        emit(f"global_load_dword {v(PX)}, {v(TMP)}, %[evm]")
        emit(f"global_load_dword {v(PX + 1)}, {v(NX)}, %[evm]")
        emit("s_mov_b64 exec, -1")
        emit("s_mov_b32 %[e_cnt], 23")
        emit("s_nop 1")
        emit(f"s_branch econt_{tag}_%=")


def exit_stub(p, ml, rs):
    tag = f"{p}{ml}{rs}"
    Tl = TR[p ^ 1]
    emit(f"exit_{tag}_%=:")
    emit("s_waitcnt lgkmcnt(0)")
    if rs:
        # back to the entry layout: cell 0 -> KQ[0], cell 1 -> KQ[1], incoming -> KQ[2], through the per-cell temps
        tq = [LPD[0], TD[0], TU[0]]                          # three free quads (v106..v117)
        for j in range(3):
            emit(f"v_mov_b64 {vp(tq[j])}, {vp(KQ[(rs + j) % 3])}")
            emit(f"v_mov_b64 {vp(tq[j] + 2)}, {vp(KQ[(rs + j) % 3] + 2)}")
        for j in range(3):
            emit(f"v_mov_b64 {vp(KQ[j])}, {vp(tq[j])}")
            emit(f"v_mov_b64 {vp(KQ[j] + 2)}, {vp(tq[j] + 2)}")
    if ml == 'R':
        mv = [("L0", Tl['c0']), ("L1", Tl['c1']), ("U0", Tl['c1']), ("U1", Tl['cs'])]
    else:
        mv = [("U0", Tl['c0']), ("U1", Tl['c1']), ("L0", Tl['cs']), ("L1", Tl['c0'])]
    for name, reg in mv:
        emit(f"v_mov_b64 %[{name}], {vp(reg)}")
    emit("s_branch done_%=")


def variant_code(border):
    """Decision + 8 bodies + exit stubs of one variant, labels suffixed so both fit in one statement."""
    global BORDER
    BORDER = border
    del out[:]
    emit("s_nop 1")
    emit(f"v_readlane_b32 %[t0], {v(MF0)}, 0")
    emit("s_nop 1")
    emit(f"v_cmp_lt_f32 vcc, %[t0], {v(MF1)}")
    lbl = decide(0, 'R', 0, entry=True)
    llinf_block(lbl, 0, 'R', 0)
    for rs in (0, 1, 2):
        for p in (0, 1):
            for ml in "RD":
                for m in "RD":
                    body(p, ml, m, rs)
    for rs in (0, 1, 2):
        for p in (0, 1):
            for ml in "RD":
                exit_stub(p, ml, rs)
    sfx = "B" if border else "I"
    return [ln.replace("done_%=", "DONE").replace("_%=", f"_{sfx}%=").replace("DONE", "done_%=") for ln in out]


def main():
    # ---- entry: operands -> fixed registers (common to both variants)
    # the incoming quad and NX are dead between bands; the pending refill registers of the FIFO lanes travel with the state
    ent = [
        (MF0, "Pf0"), (MF1, "Pf1"), (X0, "x0"), (X1, "x1"), (G0, "g0"), (C0, "c0"), (G1, "g1"), (C1, "c1"),
        (A0, "a1"), (A1, "a2"), (A2, "a3"), (ACC, "acc"), (TOFF, "toff"), (LANE, "lane"),
        (PX, "px1"), (PX + 1, "px0"), (PKA, "kag"), (PKA + 1, "kac"), (PKB, "kbg"), (PKB + 1, "kbc"),
    ]
    wide = [(I0, "i0"), (I1, "i1"), (PKA + 2, "kai"), (PKB + 2, "kbi")]
    head = []
    for reg, name in ent:
        head.append(f"v_mov_b32 {v(reg)}, %[{name}]")
    for reg, name in wide + [(TR[1]['c0'], "L0"), (TR[1]['c1'], "L1"), (TR[1]['cs'], "U1")]:
        head.append(f"v_mov_b64 {vp(reg)}, %[{name}]")
    # the incoming quad's lane 63 keeps what it has on every right move: give it a defined value once
    head += [f"v_mov_b64 {vp(NK)}, 0", f"v_mov_b64 {vp(NK + 2)}, 0"]
    head += [f"v_mov_b32 {v(NINF)}, 0xff800000", f"v_mov_b32 {v(SHR)}, 0xff800000", f"v_mov_b32 {v(SHD)}, 0xff800000",
             f"v_lshlrev_b32 {v(O0)}, 1, {v(LANE)}", f"v_lshl_or_b32 {v(O1)}, {v(LANE)}, 1, 1",
             "s_cmp_eq_u32 %[mode], 0", "s_cbranch_scc0 border_start_%="]
    interior = variant_code(False)
    border = ["border_start_%=:"] + variant_code(True)
    tail = ["done_%=:", "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    for reg, name in ent:
        if name in ("lane",):
            continue
        tail.append(f"v_mov_b32 %[{name}], {v(reg)}")
    for reg, name in wide:
        tail.append(f"v_mov_b64 %[{name}], {vp(reg)}")
    tail.append("s_nop 1")
    lines = head + interior + border + tail
    text = "\n".join(f'    "{ln}\\n\\t"' for ln in lines)
    clob = ", ".join(f'"v{i}"' for i in range(VB, VEND))
    inc = f"""/* GENERATED by tools/gen_fill_asm.py — do not edit. See that file for the register map and hazards.
 * One statement, two variants selected by %[mode]: 0 = interior (no masks), 1 = border (masks + end-point scan). */
#define ABEA_FILL_ASM \\
{text.replace(chr(10), " " + chr(92) + chr(10))}
#define ABEA_FILL_CLOBBERS {clob}, "vcc", "scc", "memory"
"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "f5c_amd", "csrc", "abea_fill.inc")
    _emit_file(path, inc, len(lines))


if __name__ == "__main__":
    main()


# =====================================================================================================
# Traceback walk (align.c:452-499) on the scalar unit: one inline-asm statement, ~40 SALU per step.
# Fixed SGPRs s72..s99, VGPRs v64..v75 (the fill statement's range; the two statements never overlap).
# =====================================================================================================
WALK_RADIUS = 12     # lane pairs either side of the path that a prefetch covers (measured at 4 / 8 / 12: DESIGN.md §4.3)


def gen_walk():
    """Scalar-unit traceback walk.  Round 3: a trace group is no longer read back whole (1 KiB per 32 bands).  The
    prefetch of the group below loads only the 64-byte sectors holding the lanes within WALK_RADIUS lane pairs of the
    path's current lane pair, plus the sector of the move lane (lane 50); if the path leaves that window before the group
    is done with (rare: the adaptive band keeps the path near its middle) the whole group is loaded again."""
    o = []
    e = o.append
    K, E, BI, LLK, G, GAP, MAXGAP, CWD, SH2, NFL, LP, DK = (f"s{i}" for i in range(76, 88))
    TLO, THI, MV, T64 = "s[88:89]", "s[90:91]", "s[92:93]", "s[94:95]"
    T64LO = "s94"
    T, U, X, Y = "s96", "s97", "s98", "s99"
    WC, WR, NWC, RLD = "s72", "s73", "s74", "s75"      # window centre / radius of the current group, centre of the next, reload count
    CW = [f"v{i}" for i in range(VB, VB + 4)]      # current 32-band trace group: one uint4 per lane
    NXG = [f"v{i}" for i in range(VB + 4, VB + 8)] # prefetched group below
    CV, VT, L16, L4 = f"v{VB + 8}", f"v{VB + 9}", f"v{VB + 10}", f"v{VB + 11}"
    W = WALK_RADIUS
    # ---- entry
    e(f"s_mov_b32 {K}, %[k0]"); e(f"s_mov_b32 {E}, %[e0]"); e(f"s_mov_b32 {LLK}, %[llk0]")
    e(f"s_add_u32 {T}, {K}, {E}"); e(f"s_add_u32 {T}, {T}, 2")           # b = e + k + 2
    e(f"s_lshr_b32 {G}, {T}, 5"); e(f"s_and_b32 {BI}, {T}, 31")
    for r in (GAP, MAXGAP, CWD, SH2, NFL, DK, RLD, WC):
        e(f"s_mov_b32 {r}, 0")
    e(f"s_mov_b32 {WR}, 64")                                               # the first group is loaded whole
    e(f"v_mov_b32 {CV}, 0")
    e(f"v_lshlrev_b32 {L16}, 4, %[lane]"); e(f"v_lshlrev_b32 {L4}, 2, %[lane]")
    e(f"s_lshl_b32 {T}, {G}, 10")
    # 64-bit base + (g << 10): trace groups are 1 KiB
    e(f"s_mov_b64 {T64}, %[trace]"); e(f"s_add_u32 s94, s94, {T}"); e("s_addc_u32 s95, s95, 0")
    e(f"global_load_dwordx4 v[{VB}:{VB + 3}], {L16}, {T64}")
    e("group_top_%=:")
    e(f"s_max_i32 {T}, {G}, 1"); e(f"s_sub_u32 {T}, {T}, 1"); e(f"s_lshl_b32 {T}, {T}, 10")
    e(f"s_mov_b64 {T64}, %[trace]"); e(f"s_add_u32 s94, s94, {T}"); e("s_addc_u32 s95, s95, 0")
    # window of the prefetch: lanes [(c - W) & ~3, (c + W) | 3] clamped to [0, 51], c = lane pair of the path now
    e(f"s_sub_u32 {NWC}, {K}, {LLK}"); e(f"s_lshr_b32 {NWC}, {NWC}, 1")
    e(f"s_sub_i32 {X}, {NWC}, {W}"); e(f"s_max_i32 {X}, {X}, 0"); e(f"s_andn2_b32 {X}, {X}, 3")      # first lane
    e(f"s_add_u32 {Y}, {NWC}, {W}"); e(f"s_min_u32 {Y}, {Y}, 51"); e(f"s_or_b32 {Y}, {Y}, 3")        # last lane
    e(f"s_sub_u32 {Y}, {Y}, {X}"); e(f"s_add_u32 {Y}, {Y}, 1")                                       # lane count (<= 2W + 7)
    e(f"s_bfm_b64 {TLO}, {Y}, {X}")                                          # TLO/THI are reloaded at the first step of the group
    e("s_or_b32 s89, s89, 0xF0000")                                         # lanes 48..51: the sector of the move lane
    e(f"s_mov_b64 exec, {TLO}")
    e(f"global_load_dwordx4 v[{VB + 4}:{VB + 7}], {L16}, {T64}")                       # prefetch the group below
    e("s_mov_b64 exec, -1")
    e("s_waitcnt vmcnt(1)")                                                 # the current group has landed
    e(f"v_readlane_b32 s92, {CW[0]}, 50")                                  # band moves of this group ...
    e(f"v_readlane_b32 s93, {CW[1]}, 50")                                  # ... and of the group below
    e(f"s_mov_b32 {LP}, -1")
    _walk_loops(e, locals())
    _finish_walk(o)


def _walk_loops(e, r):
    """The two step loops (upper / lower 16 bands of the trace group) and their out-of-line blocks; r = gen_walk's register names."""
    K, E, BI, LLK, G, GAP, MAXGAP, CWD, SH2, NFL, LP, DK = (r[n] for n in "K E BI LLK G GAP MAXGAP CWD SH2 NFL LP DK".split())
    TLO, THI, MV, T64, T64LO, T, U, X, Y = (r[n] for n in "TLO THI MV T64 T64LO T U X Y".split())
    WC, WR, NWC, RLD, CW, NXG, CV, VT, L16, L4, W = (r[n] for n in "WC WR NWC RLD CW NXG CV VT L16 L4 W".split())
    # band-move word shifted so that bit 0 = move(b), bit 1 = move(b-1) for the current band; kept so by every step
    e(f"s_sub_u32 {T}, 31, {BI}"); e(f"s_lshr_b64 {MV}, {MV}, {T}")
    e(f"s_bitcmp1_b32 {BI}, 4")
    e("s_cbranch_scc0 step_lo_%=")
    for half in ("hi", "lo"):
        e(f"step_{half}_%=:")
        e(f"s_sub_u32 {T}, {K}, {LLK}")                                     # band offset of (e,k)
        e(f"s_lshr_b32 {U}, {T}, 1")
        e(f"s_cmp_eq_u32 {U}, {LP}")
        e(f"s_cbranch_scc0 reload_lp_{half}_%=")
        e(f"step_cont_{half}_%=:")
        e(f"s_and_b32 {T}, {T}, 1"); e(f"s_lshl_b32 {T}, {T}, 1")
        e(f"s_xor_b32 {X}, {BI}, 7")
        e(f"s_lshl2_add_u32 {T}, {X}, {T}")                                 # bit position in the 128 bits; a 64-bit shift takes it mod 64
        e(f"s_lshr_b64 {T64}, {THI if half == 'hi' else TLO}, {T}")
        e(f"s_and_b32 {U}, {T64LO}, 3")
        e(f"s_min_u32 {U}, {U}, 2")                                          # from code: 0 D, 1 U, 2 L (3 = L and U tie)
        e(f"s_and_b32 {T}, s92, 3")                                          # bit0 = move(b), bit1 = move(b-1)
        e(f"s_lshl_b32 {X}, {U}, {SH2}"); e(f"s_or_b32 {CWD}, {CWD}, {X}"); e(f"s_add_u32 {SH2}, {SH2}, 2")
        e(f"s_bitcmp1_b32 {SH2}, 5")
        e(f"s_cbranch_scc1 flush_{half}_%=")
        e(f"flush_ret_{half}_%=:")
        e(f"s_and_b32 {DK}, {U}, 1"); e(f"s_xor_b32 {DK}, {DK}, 1")           # D,L step the k-mer
        e(f"s_lshr_b32 {X}, {U}, 1")                                         # isL
        e(f"s_xor_b32 {Y}, {X}, 1")                                          # de: D,U step the event
        e(f"s_add_u32 {GAP}, {GAP}, 1"); e(f"s_mul_i32 {GAP}, {GAP}, {X}")    # gap = isL ? gap + 1 : 0
        e(f"s_max_i32 {MAXGAP}, {MAXGAP}, {GAP}")
        e(f"s_and_b32 {U}, {DK}, {Y}")                                       # isD
        e(f"s_lshl1_add_u32 {X}, {U}, 1")                                    # 1 | isD << 1: move(b-1) counts only on a diagonal step
        e(f"s_and_b32 {T}, {T}, {X}"); e(f"s_bcnt1_i32_b32 {T}, {T}")
        e(f"s_sub_u32 {LLK}, {LLK}, {T}")
        e(f"s_sub_u32 {K}, {K}, {DK}"); e("s_cbranch_scc1 done_%=")          # borrow: ran off k-mer 0
        e(f"s_sub_u32 {E}, {E}, {Y}"); e("s_cbranch_scc1 done_%=")           # borrow: ran off event 0
        e(f"s_add_u32 {T}, {DK}, {Y}")
        e(f"s_lshr_b64 {MV}, {MV}, {T}")                                     # the move word follows the band
        e(f"s_sub_u32 {BI}, {BI}, {T}")
        if half == "hi":
            e(f"s_bitcmp1_b32 {BI}, 4")                                      # 16..31: still the upper half (no borrow possible from >= 16)
            e("s_cbranch_scc1 step_hi_%=")                                  # else fall into the lower-half loop
        else:
            e("s_cbranch_scc0 step_lo_%=")                                  # no borrow: still in this 32-band group
    e(f"s_add_u32 {BI}, {BI}, 32"); e(f"s_sub_u32 {G}, {G}, 1")
    e("s_waitcnt vmcnt(0)")
    for a, b in zip(CW, NXG):
        e(f"v_mov_b32 {a}, {b}")
    e(f"s_mov_b32 {WC}, {NWC}"); e(f"s_mov_b32 {WR}, {W}")
    e("s_branch group_top_%=")
    for half in ("hi", "lo"):
        e(f"reload_lp_{half}_%=:")
        e(f"s_mov_b32 {LP}, {U}")
        e(f"s_sub_i32 {X}, {U}, {WC}"); e(f"s_abs_i32 {X}, {X}")
        e(f"s_cmp_le_u32 {X}, {WR}")
        e(f"s_cbranch_scc0 full_reload_{half}_%=")
        e(f"reload_ret_{half}_%=:")
        e(f"v_readlane_b32 s88, {CW[0]}, {U}"); e(f"v_readlane_b32 s89, {CW[1]}, {U}")
        e(f"v_readlane_b32 s90, {CW[2]}, {U}"); e(f"v_readlane_b32 s91, {CW[3]}, {U}")
        e(f"s_branch step_cont_{half}_%=")
        e(f"full_reload_{half}_%=:")
        e(f"s_lshl_b32 {X}, {G}, 10")
        e(f"s_mov_b64 {T64}, %[trace]"); e(f"s_add_u32 s94, s94, {X}"); e("s_addc_u32 s95, s95, 0")
        e("s_waitcnt vmcnt(0)")
        e(f"global_load_dwordx4 v[{VB}:{VB + 3}], {L16}, {T64}")
        e(f"s_mov_b32 {WR}, 64"); e(f"s_add_u32 {RLD}, {RLD}, 1")
        e("s_waitcnt vmcnt(0)")
        e(f"s_branch reload_ret_{half}_%=")
        e(f"flush_{half}_%=:")
        e(f"s_and_b32 {X}, {NFL}, 63")
        e(f"v_cmp_eq_u32 vcc, {X}, %[lane]")
        e(f"v_mov_b32 {VT}, {CWD}")
        e(f"s_mov_b32 {CWD}, 0"); e(f"s_mov_b32 {SH2}, 0")
        e(f"v_cndmask_b32 {CV}, {CV}, {VT}, vcc")
        e(f"s_add_u32 {NFL}, {NFL}, 1")
        e(f"s_and_b32 {X}, {NFL}, 63")
        e(f"s_cbranch_scc1 flush_ret_{half}_%=")
        e(f"s_sub_u32 {X}, {NFL}, 64"); e(f"s_lshl_b32 {X}, {X}, 2")
        e(f"v_add_u32 {VT}, {X}, {L4}")
        e("s_nop 1")
        e(f"global_store_dword {VT}, {CV}, %[codes]")
        e(f"s_branch flush_ret_{half}_%=")
    e("done_%=:")
    e("s_waitcnt vmcnt(0)")
    e(f"s_add_u32 %[last_k], {K}, {DK}")
    e(f"s_mov_b32 %[o_cwd], {CWD}"); e(f"s_mov_b32 %[o_sh2], {SH2}"); e(f"s_mov_b32 %[o_nfl], {NFL}")
    e(f"s_mov_b32 %[o_maxgap], {MAXGAP}"); e(f"s_mov_b32 %[o_reloads], {RLD}")
    e(f"v_mov_b32 %[o_cv], {CV}")


def _finish_walk(o):
    text = "\n".join(f'    "{ln}\\n\\t"' for ln in o)
    clob = ", ".join([f'"s{i}"' for i in range(72, 100)] + [f'"v{i}"' for i in range(VB, VB + 12)])
    inc = f"""/* GENERATED by tools/gen_fill_asm.py — do not edit. Scalar-unit traceback walk. */
#define ABEA_WALK_ASM \\
{text.replace(chr(10), " " + chr(92) + chr(10))}
#define ABEA_WALK_CLOBBERS {clob}, "vcc", "scc", "memory"
"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "f5c_amd", "csrc", "abea_walk.inc")
    _emit_file(path, inc, len(o))


if __name__ == "__main__":
    gen_walk()
    if stale:
        raise SystemExit(f"generated files differ from tools/gen_fill_asm.py output: {stale}")
