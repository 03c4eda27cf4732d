#!/usr/bin/env python3
"""A small interpreter for the subset of the gfx950 ISA that the hand-written statements use (f5c_amd/csrc/abea_fill.inc,
abea_walk.inc): ONE wavefront of 64 lanes, its VGPRs / SGPRs / VCC / EXEC / SCC / M0, its LDS and a flat global memory.

Test infrastructure (tests/test_asm_emulated.py): the generated text is executed instruction by instruction on the CPU and
the result compared with the oracle, so that the arithmetic and the control flow of the loop are checked where there is
no GPU — timing, hazards and s_waitcnt are NOT modelled here (tools/asm_lint.py checks those statically).

Floating point is numpy's IEEE float32 / float64 (round to nearest even, denormals kept — the kernel's mode); v_fma_f32 goes
through glibc's fmaf.  Invalid operations give the GPU's canonical +qNaN (0x7FC00000; x86 would give the negative one),
because the loop reads sign bits of differences.  Every instruction honours EXEC for its vector writes.
"""
import ctypes
import ctypes.util
import re
import numpy as np

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.fmaf.restype = ctypes.c_float
_libm.fmaf.argtypes = [ctypes.c_float] * 3
LANES = np.arange(64, dtype=np.int64)


def _f32(bits):
    return np.asarray(bits, dtype=np.uint32).view(np.float32)


def _bits32(f):
    a = np.asarray(f, dtype=np.float32)
    b = a.view(np.uint32).copy()
    b[np.isnan(a)] = 0x7FC00000
    return b


def _f64(bits):
    return np.asarray(bits, dtype=np.uint64).view(np.float64)


def _bits64(f):
    a = np.asarray(f, dtype=np.float64)
    b = a.view(np.uint64).copy()
    b[np.isnan(a)] = 0x7FF8000000000000
    return b


def _s32(x):
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


class Wave:
    def __init__(self, lds_bytes=65536, heap_bytes=1 << 20):
        self.v = np.zeros((512, 64), dtype=np.uint32)      # rows 0..255 physical, 256.. named VGPR operands
        self.s = np.zeros(128, dtype=np.uint64)            # physical SGPRs (32-bit values)
        self.sym = {}                                      # named scalar operands: name -> int (up to 64 bits)
        self.vsym = {}                                     # named vector operands: name -> first row
        self.vcc = 0
        self.exec = M64
        self.scc = 0
        self.m0 = 0
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.heap = np.zeros(heap_bytes, dtype=np.uint8)
        self.heap_top = 4096
        self.n_exec = 0
        self.count = {}                                    # executed instructions by mnemonic
        self._next_row = 256

    # ------------------------------------------------------------------ memory
    def alloc(self, data):
        """Copy a numpy array into the heap (4 KiB aligned); returns its address."""
        raw = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        addr = self.heap_top
        end = addr + len(raw)
        if end > len(self.heap):
            self.heap = np.concatenate([self.heap, np.zeros(max(len(self.heap), end), dtype=np.uint8)])
        self.heap[addr:end] = raw
        self.heap_top = (end + 4095) & ~4095
        return addr

    def view(self, addr, dtype, n):
        dt = np.dtype(dtype)
        return self.heap[addr:addr + n * dt.itemsize].view(dt)

    def bind_v(self, name, values, wide=False):
        """Named vector operand: 64 uint32 (or, wide, 64 uint64 over two rows)."""
        if name not in self.vsym:
            self.vsym[name] = self._next_row
            self._next_row += 2
        r = self.vsym[name]
        vals = np.asarray(values)
        if wide:
            u = vals.astype(np.uint64) if vals.dtype != np.uint64 else vals
            self.v[r] = (u & M32).astype(np.uint32)
            self.v[r + 1] = (u >> np.uint64(32)).astype(np.uint32)
        else:
            self.v[r] = vals.astype(np.uint32)

    def get_v(self, name, wide=False):
        r = self.vsym[name]
        if wide:
            return self.v[r].astype(np.uint64) | (self.v[r + 1].astype(np.uint64) << np.uint64(32))
        return self.v[r].copy()

    # ------------------------------------------------------------------ operands
    def _tok(self, t):
        """-> (kind, payload): 'v' first row, n rows; 's' first index, n; 'sym' name; 'vsym' row; 'imm' int; 'fimm' float;
        'special' name."""
        t = t.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", t)
        if m:
            return ("v", int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1)
        m = re.fullmatch(r"v(\d+)", t)
        if m:
            return ("v", int(m.group(1)), 1)
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", t)
        if m:
            return ("s", int(m.group(1)), int(m.group(2)) - int(m.group(1)) + 1)
        m = re.fullmatch(r"s(\d+)", t)
        if m:
            return ("s", int(m.group(1)), 1)
        m = re.fullmatch(r"%\[(\w+)\]", t)
        if m:
            return ("vsym", m.group(1), 0) if m.group(1) in self.vsym else ("sym", m.group(1), 0)
        if t in ("vcc", "vcc_lo", "vcc_hi", "exec", "exec_lo", "exec_hi", "m0"):
            return ("special", t, 0)
        if re.fullmatch(r"-?\d+\.\d*", t):
            return ("fimm", float(t), 0)
        return ("imm", int(t, 0), 0)

    def rd32(self, t, as_float=False):
        """32-bit source: a (64,) uint32 array for vector registers, else a Python int."""
        k, a, n = self._tok(t)
        if k == "v":
            return self.v[a]
        if k == "vsym":
            return self.v[self.vsym[a]]
        if k == "s":
            return int(self.s[a]) & M32
        if k == "sym":
            return int(self.sym[a]) & M32
        if k == "imm":
            return a & M32
        if k == "fimm":
            return int(np.float32(a).view(np.uint32))
        return {"vcc_lo": self.vcc & M32, "vcc_hi": self.vcc >> 32, "exec_lo": self.exec & M32, "exec_hi": self.exec >> 32,
                "m0": self.m0 & M32, "vcc": self.vcc & M32, "exec": self.exec & M32}[a]

    def rd64(self, t, fp=False):
        k, a, n = self._tok(t)
        if k == "v":
            return self.v[a].astype(np.uint64) | (self.v[a + 1].astype(np.uint64) << np.uint64(32))
        if k == "vsym":
            r = self.vsym[a]
            return self.v[r].astype(np.uint64) | (self.v[r + 1].astype(np.uint64) << np.uint64(32))
        if k == "s":
            return (int(self.s[a]) & M32) | ((int(self.s[a + 1]) & M32) << 32)
        if k == "sym":
            return int(self.sym[a]) & M64
        if k == "imm":
            if fp:
                raise ValueError("integer literal as an f64 operand")
            return a & M64 if a >= 0 else a & M64
        if k == "fimm":
            return int(np.float64(a).view(np.uint64))
        return {"vcc": self.vcc, "exec": self.exec}[a]

    def wr_v(self, t, vals, rows=1):
        """Write vector register(s) under EXEC; vals uint32 (rows=1) or uint64 (rows=2) or a list of uint32 arrays."""
        k, a, n = self._tok(t)
        r = self.vsym[a] if k == "vsym" else a
        assert k in ("v", "vsym"), t
        if rows == 2:
            u = np.asarray(vals, dtype=np.uint64)
            parts = [(u & np.uint64(M32)).astype(np.uint32), (u >> np.uint64(32)).astype(np.uint32)]
        elif isinstance(vals, list):
            parts = vals
        else:
            parts = [np.broadcast_to(np.asarray(vals, dtype=np.uint32), (64,))]
        if self.exec == M64:
            for i, p in enumerate(parts):
                self.v[r + i] = p
        else:
            mask = self._lanebits(self.exec)
            for i, p in enumerate(parts):
                self.v[r + i] = np.where(mask, p, self.v[r + i])

    def wr_s(self, t, val, bits=32):
        k, a, n = self._tok(t)
        val = int(val)
        if k == "s":
            self.s[a] = val & M32
            if bits == 64:
                self.s[a + 1] = (val >> 32) & M32
        elif k == "sym":
            self.sym[a] = val & (M64 if bits == 64 else M32)
        elif k == "special":
            if a == "vcc":
                self.vcc = val & M64
            elif a == "vcc_lo":
                self.vcc = (self.vcc & ~M32) | (val & M32)
            elif a == "vcc_hi":
                self.vcc = (self.vcc & M32) | ((val & M32) << 32)
            elif a == "exec":
                self.exec = val & M64
            elif a == "exec_lo":
                self.exec = (self.exec & ~M32) | (val & M32)
            elif a == "exec_hi":
                self.exec = (self.exec & M32) | ((val & M32) << 32)
            elif a == "m0":
                self.m0 = val & M32
        else:
            raise ValueError("scalar write to " + t)

    @staticmethod
    def _mask_of(boolarr):
        return int(np.packbits(np.asarray(boolarr, dtype=np.uint8), bitorder="little").view(np.uint64)[0])

    def _lanebits(self, m):
        return ((np.uint64(int(m) & M64) >> LANES.astype(np.uint64)) & np.uint64(1)).astype(bool)

    # ------------------------------------------------------------------ execution
    def run(self, lines, max_steps=50_000_000):
        prog = []
        labels = {}
        for ln in lines:
            ln = ln.strip()
            if ln.endswith(":"):
                labels[ln[:-1]] = len(prog)
                continue
            parts = ln.split(None, 1)
            op = parts[0]
            rest = parts[1] if len(parts) > 1 else ""
            mods = ""
            m = re.search(r"\s(wave_sh[lr]:\d+|wave_ro[lr]:\d+|offset:\d+|lgkmcnt|vmcnt)", " " + rest)
            if op in ("s_waitcnt",):
                ops = []
            else:
                if m:
                    cut = rest.find(m.group(1))
                    mods = rest[cut:]
                    rest = rest[:cut]
                ops = [x.strip() for x in rest.split(",") if x.strip()]
            prog.append((op, ops, mods, ln))
        pc = 0
        steps = 0
        with np.errstate(all="ignore"):
            while pc < len(prog):
                op, ops, mods, text = prog[pc]
                steps += 1
                if steps > max_steps:
                    raise RuntimeError("emulator: step limit")
                self.count[op] = self.count.get(op, 0) + 1
                if op == "s_branch":
                    pc = labels[ops[0]]
                    continue
                if op == "s_cbranch_scc1":
                    pc = labels[ops[0]] if self.scc else pc + 1
                    continue
                if op == "s_cbranch_scc0":
                    pc = labels[ops[0]] if not self.scc else pc + 1
                    continue
                fn = getattr(self, "i_" + op, None)
                if fn is None:
                    raise NotImplementedError(text)
                fn(ops, mods)
                pc += 1
        self.n_exec += steps

    # ------------------------------------------------------------------ SALU
    def i_s_nop(self, o, m): pass
    def i_s_waitcnt(self, o, m): pass

    def i_s_mov_b32(self, o, m): self.wr_s(o[0], self.rd32(o[1]))
    def i_s_mov_b64(self, o, m):
        k, a, n = self._tok(o[1])
        self.wr_s(o[0], (a & M64) if k == "imm" else self.rd64(o[1]), 64)

    def i_s_add_u32(self, o, m):
        r = self.rd32(o[1]) + self.rd32(o[2]); self.wr_s(o[0], r); self.scc = int(r > M32)
    def i_s_addc_u32(self, o, m):
        r = self.rd32(o[1]) + self.rd32(o[2]) + self.scc; self.wr_s(o[0], r); self.scc = int(r > M32)
    def i_s_sub_u32(self, o, m):
        a, b = self.rd32(o[1]), self.rd32(o[2]); self.wr_s(o[0], a - b); self.scc = int(b > a)
    def i_s_sub_i32(self, o, m):
        a, b = _s32(self.rd32(o[1])), _s32(self.rd32(o[2])); r = a - b
        self.wr_s(o[0], r); self.scc = int(r < -(1 << 31) or r >= (1 << 31))
    def _logic(self, o, f, bits=32):
        rd = self.rd32 if bits == 32 else self.rd64
        r = f(rd(o[1]), rd(o[2])) & (M32 if bits == 32 else M64)
        self.wr_s(o[0], r, bits); self.scc = int(r != 0)
    def i_s_and_b32(self, o, m): self._logic(o, lambda a, b: a & b)
    def i_s_or_b32(self, o, m): self._logic(o, lambda a, b: a | b)
    def i_s_xor_b32(self, o, m): self._logic(o, lambda a, b: a ^ b)
    def i_s_andn2_b32(self, o, m): self._logic(o, lambda a, b: a & ~b)
    def i_s_and_b64(self, o, m): self._logic(o, lambda a, b: a & b, 64)
    def i_s_not_b32(self, o, m):
        r = ~self.rd32(o[1]) & M32; self.wr_s(o[0], r); self.scc = int(r != 0)
    def i_s_lshl_b32(self, o, m): self._logic(o, lambda a, b: a << (b & 31))
    def i_s_lshr_b32(self, o, m): self._logic(o, lambda a, b: a >> (b & 31))
    def i_s_lshr_b64(self, o, m):
        r = self.rd64(o[1]) >> (self.rd32(o[2]) & 63); self.wr_s(o[0], r, 64); self.scc = int(r != 0)
    def i_s_min_u32(self, o, m):
        a, b = self.rd32(o[1]), self.rd32(o[2]); self.wr_s(o[0], min(a, b)); self.scc = int(a < b)
    def i_s_min_i32(self, o, m):
        a, b = _s32(self.rd32(o[1])), _s32(self.rd32(o[2])); self.wr_s(o[0], min(a, b)); self.scc = int(a < b)
    def i_s_max_i32(self, o, m):
        a, b = _s32(self.rd32(o[1])), _s32(self.rd32(o[2])); self.wr_s(o[0], max(a, b)); self.scc = int(a > b)
    def i_s_mul_i32(self, o, m): self.wr_s(o[0], (_s32(self.rd32(o[1])) * _s32(self.rd32(o[2]))) & M32)
    def i_s_abs_i32(self, o, m):
        r = abs(_s32(self.rd32(o[1]))) & M32; self.wr_s(o[0], r); self.scc = int(r != 0)
    def i_s_cmp_eq_u32(self, o, m): self.scc = int(self.rd32(o[0]) == self.rd32(o[1]))
    def i_s_cmp_lt_u32(self, o, m): self.scc = int(self.rd32(o[0]) < self.rd32(o[1]))
    def i_s_cmp_le_u32(self, o, m): self.scc = int(self.rd32(o[0]) <= self.rd32(o[1]))
    def i_s_bitcmp1_b32(self, o, m): self.scc = (self.rd32(o[0]) >> (self.rd32(o[1]) & 31)) & 1
    def i_s_cselect_b32(self, o, m): self.wr_s(o[0], self.rd32(o[1]) if self.scc else self.rd32(o[2]))
    def i_s_cselect_b64(self, o, m): self.wr_s(o[0], self.rd64(o[1]) if self.scc else self.rd64(o[2]), 64)
    def i_s_flbit_i32_b32(self, o, m):
        a = self.rd32(o[1]); self.wr_s(o[0], (32 - a.bit_length()) if a else M32)
    def i_s_bcnt1_i32_b32(self, o, m):
        r = bin(self.rd32(o[1])).count("1"); self.wr_s(o[0], r); self.scc = int(r != 0)
    def i_s_lshl1_add_u32(self, o, m):
        r = (self.rd32(o[1]) << 1) + self.rd32(o[2]); self.wr_s(o[0], r); self.scc = int(r > M32)
    def i_s_lshl2_add_u32(self, o, m):
        r = (self.rd32(o[1]) << 2) + self.rd32(o[2]); self.wr_s(o[0], r); self.scc = int(r > M32)
    def i_s_bfm_b64(self, o, m):
        self.wr_s(o[0], (((1 << (self.rd32(o[1]) & 63)) - 1) << (self.rd32(o[2]) & 63)) & M64, 64)

    # ------------------------------------------------------------------ VALU
    def _vec(self, x):
        return np.broadcast_to(np.asarray(x, dtype=np.uint32), (64,))

    def i_v_mov_b32(self, o, m): self.wr_v(o[0], self._vec(self.rd32(o[1])).copy())
    def i_v_mov_b64(self, o, m): self.wr_v(o[0], np.broadcast_to(np.asarray(self.rd64(o[1]), dtype=np.uint64), (64,)).copy(), 2)
    def i_v_not_b32(self, o, m): self.wr_v(o[0], ~self._vec(self.rd32(o[1])))
    def i_v_xor_b32(self, o, m): self.wr_v(o[0], self._vec(self.rd32(o[1])) ^ self._vec(self.rd32(o[2])))

    def i_v_mov_b32_dpp(self, o, m):
        src = self._vec(self.rd32(o[1]))
        k, a, n = self._tok(o[0])
        old = self.v[self.vsym[a] if k == "vsym" else a]
        new = old.copy()
        if "wave_shr:1" in m:
            new[1:] = src[:-1]
        elif "wave_shl:1" in m:
            new[:-1] = src[1:]
        elif "wave_ror:1" in m:                           # rotate towards higher lanes: lane 0 <- lane 63
            new = np.roll(src, 1)
        elif "wave_rol:1" in m:
            new = np.roll(src, -1)
        else:
            raise NotImplementedError("dpp " + m)
        self.wr_v(o[0], new)

    def _f32op(self, o, f):
        a = _f32(self._vec(self.rd32(o[1]))); b = _f32(self._vec(self.rd32(o[2])))
        self.wr_v(o[0], _bits32(f(a, b)))
    def i_v_sub_f32(self, o, m): self._f32op(o, lambda a, b: a - b)
    def i_v_add_f32(self, o, m): self._f32op(o, lambda a, b: a + b)
    def i_v_mul_f32(self, o, m): self._f32op(o, lambda a, b: a * b)
    def i_v_fma_f32(self, o, m):
        a, b, c = (_f32(self._vec(self.rd32(t))) for t in o[1:4])
        r = np.array([_libm.fmaf(float(x), float(y), float(z)) for x, y, z in zip(a, b, c)], dtype=np.float32)
        self.wr_v(o[0], _bits32(r))
    def i_v_max3_f32(self, o, m):
        a, b, c = (_f32(self._vec(self.rd32(t))) for t in o[1:4])
        self.wr_v(o[0], _bits32(np.fmax(np.fmax(a, b), c)))

    def _f64src(self, t):
        return _f64(np.broadcast_to(np.asarray(self.rd64(t, fp=True), dtype=np.uint64), (64,)))
    def i_v_add_f64(self, o, m): self.wr_v(o[0], _bits64(self._f64src(o[1]) + self._f64src(o[2])), 2)
    def i_v_mul_f64(self, o, m): self.wr_v(o[0], _bits64(self._f64src(o[1]) * self._f64src(o[2])), 2)
    def i_v_cvt_f64_f32(self, o, m):
        self.wr_v(o[0], _bits64(_f32(self._vec(self.rd32(o[1]))).astype(np.float64)), 2)
    def i_v_cvt_f32_f64(self, o, m): self.wr_v(o[0], _bits32(self._f64src(o[1]).astype(np.float32)))
    def i_v_cvt_f64_i32(self, o, m):
        self.wr_v(o[0], _bits64(self._vec(self.rd32(o[1])).astype(np.int32).astype(np.float64)), 2)
    def i_v_cvt_f64_u32(self, o, m):
        self.wr_v(o[0], _bits64(self._vec(self.rd32(o[1])).astype(np.float64)), 2)

    def i_v_alignbit_b32(self, o, m):
        hi = self._vec(self.rd32(o[1])).astype(np.uint64); lo = self._vec(self.rd32(o[2])).astype(np.uint64)
        sh = self._vec(self.rd32(o[3])).astype(np.uint64) & np.uint64(31)
        self.wr_v(o[0], ((((hi << np.uint64(32)) | lo) >> sh) & np.uint64(M32)).astype(np.uint32))
    def i_v_add_u32(self, o, m): self.wr_v(o[0], self._vec(self.rd32(o[1])) + self._vec(self.rd32(o[2])))
    def i_v_sub_u32(self, o, m): self.wr_v(o[0], self._vec(self.rd32(o[1])) - self._vec(self.rd32(o[2])))
    def i_v_subrev_u32(self, o, m): self.wr_v(o[0], self._vec(self.rd32(o[2])) - self._vec(self.rd32(o[1])))
    def i_v_min_i32(self, o, m):
        self.wr_v(o[0], np.minimum(self._vec(self.rd32(o[1])).astype(np.int32), self._vec(self.rd32(o[2])).astype(np.int32)).astype(np.uint32))
    def i_v_lshlrev_b32(self, o, m):
        self.wr_v(o[0], self._vec(self.rd32(o[2])) << (self._vec(self.rd32(o[1])) & np.uint32(31)))
    def i_v_lshl_or_b32(self, o, m):
        self.wr_v(o[0], (self._vec(self.rd32(o[1])) << (self._vec(self.rd32(o[2])) & np.uint32(31))) | self._vec(self.rd32(o[3])))
    def i_v_lshl_add_u32(self, o, m):
        self.wr_v(o[0], (self._vec(self.rd32(o[1])) << (self._vec(self.rd32(o[2])) & np.uint32(31))) + self._vec(self.rd32(o[3])))

    def i_v_cndmask_b32(self, o, m):
        sel = self._lanebits(self.rd64(o[3]))
        self.wr_v(o[0], np.where(sel, self._vec(self.rd32(o[2])), self._vec(self.rd32(o[1]))))

    def _cmp(self, o, f, as_float):
        a, b = self._vec(self.rd32(o[1])), self._vec(self.rd32(o[2]))
        if as_float:
            a, b = _f32(a), _f32(b)
        self.wr_s(o[0], self._mask_of(f(a, b)) & self.exec, 64)
    def i_v_cmp_lt_f32(self, o, m): self._cmp(o, lambda a, b: a < b, True)
    def i_v_cmp_gt_f32(self, o, m): self._cmp(o, lambda a, b: a > b, True)
    def i_v_cmp_ge_f32(self, o, m): self._cmp(o, lambda a, b: a >= b, True)
    def i_v_cmp_eq_f32(self, o, m): self._cmp(o, lambda a, b: a == b, True)
    def i_v_cmp_gt_u32(self, o, m): self._cmp(o, lambda a, b: a > b, False)
    def i_v_cmp_eq_u32(self, o, m): self._cmp(o, lambda a, b: a == b, False)

    def i_v_readlane_b32(self, o, m): self.wr_s(o[0], int(self._vec(self.rd32(o[1]))[self.rd32(o[2]) & 63]))
    def i_v_readfirstlane_b32(self, o, m):
        first = (self.exec & -self.exec).bit_length() - 1 if self.exec else 0
        self.wr_s(o[0], int(self._vec(self.rd32(o[1]))[first]))
    def i_v_writelane_b32(self, o, m):
        k, a, n = self._tok(o[0])
        self.v[self.vsym[a] if k == "vsym" else a][self.rd32(o[2]) & 63] = self.rd32(o[1])

    # ------------------------------------------------------------------ LDS / global memory
    @staticmethod
    def _offset(m):
        mm = re.search(r"offset:(\d+)", m)
        return int(mm.group(1)) if mm else 0

    def _gather(self, mem, addr, ndw):
        out = []
        act = self._lanebits(self.exec)
        addr = np.where(act, addr, 0)                      # inactive lanes make no access
        for j in range(ndw):
            a = addr + 4 * j
            w = (mem[a].astype(np.uint32) | (mem[a + 1].astype(np.uint32) << 8) | (mem[a + 2].astype(np.uint32) << 16)
                 | (mem[a + 3].astype(np.uint32) << 24))
            out.append(w)
        return out

    def _scatter(self, mem, addr, words):
        mask = self._lanebits(self.exec)
        for j, w in enumerate(words):
            for byte in range(4):
                mem[(addr + 4 * j + byte)[mask]] = ((w >> np.uint32(8 * byte)) & np.uint32(0xFF)).astype(np.uint8)[mask]

    def i_ds_read_addtid_b32(self, o, m):
        addr = (self.m0 & 0xFFFF) + self._offset(m) + 4 * LANES
        self.wr_v(o[0], self._gather(self.lds, addr, 1)[0])
    def i_ds_write_b32(self, o, m):
        self._scatter(self.lds, self._vec(self.rd32(o[0])).astype(np.int64) + self._offset(m), [self._vec(self.rd32(o[1]))])
    def i_ds_write_b128(self, o, m):
        k, a, n = self._tok(o[1])
        self._scatter(self.lds, self._vec(self.rd32(o[0])).astype(np.int64) + self._offset(m), [self.v[a + j] for j in range(4)])

    def _gaddr(self, voff, sbase, m):
        return int(self.rd64(sbase)) + self._vec(self.rd32(voff)).astype(np.int64) + self._offset(m)
    def i_global_load_dword(self, o, m): self.wr_v(o[0], self._gather(self.heap, self._gaddr(o[1], o[2], m), 1)[0])
    def i_global_load_dwordx4(self, o, m): self.wr_v(o[0], self._gather(self.heap, self._gaddr(o[1], o[2], m), 4))
    def i_global_store_dword(self, o, m): self._scatter(self.heap, self._gaddr(o[0], o[2], m), [self._vec(self.rd32(o[1]))])
    def i_global_store_dwordx4(self, o, m):
        k, a, n = self._tok(o[1])
        self._scatter(self.heap, self._gaddr(o[0], o[2], m), [self.v[a + j] for j in range(4)])
