#!/usr/bin/env python3
"""A functional emulator for the subset of gfx950 assembly the generators in this directory emit (tools/gen_h3_*_asm.py).

Why: the glue between the hand-scheduled GEMM blocks (residual, LayerNorm, fp32 -> fp16 hi/lo split, transposer) is a few
thousand generated instructions per kernel family with a hand-made register map; a wrong index there is silent corruption,
and GPU time is the scarce resource.  This runs the generated text on the CPU - one wave = 64 lanes as numpy vectors, VGPRs,
AGPRs, SGPRs, VCC / SCC / M0, a byte-addressed LDS, an optional flat global buffer - so the glue can be held to a numpy
restatement of what it is meant to compute before it ever reaches the chip (tests/test_asm_glue_cpu.py).

What it is NOT: a timing model.  Wait states and MFMA forwarding rules are not checked; by default s_waitcnt / s_nop are
no-ops and memory operations complete in program order.  Several waves can be run in lock step over one LDS: `run_waves`
advances each wave to its next s_barrier in turn (in a chosen wave order).

Adversarial completion (r05; tests/test_asm_protocol_cpu.py): the stage hand-off of the GEMM statements is a protocol of
counted waits and barriers over an asynchronous LDS-DMA queue, and r04's stress runs found a hole in it that no functional
test saw.  Two switches make the emulator complete memory operations as LATE as the program's own waits allow:
  `late_vmem`  global_load_lds / global_load_dword* sit in a per-wave queue and land (in order) only when an
               `s_waitcnt vmcnt(n)` forces them - a tile read that the waits do not cover reads stale bytes;
  `late_lds`   ds_read results are sampled AND written back only when an `s_waitcnt lgkmcnt(n)` forces them (ds_writes keep
               their place in the queue but store at once) - a slot refilled, or a register consumed, before the read was
               waited for shows the wrong data.
With waves run to their barriers in both orders, a statement whose result survives all four combinations has no
read-before-landed, no refill-before-read and no use-before-wait on those schedules.  `Wave.in_flight()` lists what is still
queued at the end: LDS-DMA pieces are expected there (the ring runs ahead across statements), register loads are a bug.

Register layouts of the matrix instructions (gfx950):
  v_mfma_f32_16x16x16_f16  A: lane l holds A[l % 16][4 (l / 16) + e], e = 0..3 (two VGPRs);  B: B[4 (l / 16) + e][l % 16];
                           D: lane l holds D[4 (l / 16) + r][l % 16], r = 0..3
  v_mfma_f32_16x16x32_f16  the same with eight k values per lane: k = 8 (l / 16) + e (four VGPRs)
"""
import re

import numpy as np

LANES = 64


def _f32(u):
    return u.view(np.float32)


def _u32(f):
    return np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)


class Wave:
    def __init__(self, lds=None, gmem=None, gbase=0, wave_id=0):
        self.v = np.zeros((256, LANES), np.uint32)
        self.a = np.zeros((256, LANES), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.vcc = np.zeros(LANES, bool)
        self.scc = 0
        self.m0 = 0
        self.lds = lds if lds is not None else np.zeros(160 * 1024, np.uint8)
        self.gmem = gmem
        self.gbase = gbase
        self.wave_id = wave_id
        self.lane = np.arange(LANES, dtype=np.uint32)
        self.count = 0
        self.mfma_count = 0
        self.written_v = set()   # registers written by the program (for clobber checks)
        self.written_a = set()
        self.written_s = set()
        self.read_before_write_v = set()
        self.track_uninit = False
        self.init_v = set()
        self.init_a = set()
        self.late_vmem = False
        self.late_lds = False
        self.vm_q = []     # [(kind, closure)] in issue order; kind: "lds" (LDS-DMA) | "reg" (load into registers)
        self.lgkm_q = []

    def _retire(self, q, n):
        while len(q) > n:
            q.pop(0)[1]()

    def in_flight(self):
        return [k for k, _ in self.vm_q], [k for k, _ in self.lgkm_q]

    def drain(self):
        self._retire(self.vm_q, 0)
        self._retire(self.lgkm_q, 0)

    # ---------------------------------------------------------------- operand access
    _re_rng = re.compile(r"^([vas])\[(\d+):(\d+)\]$")
    _re_one = re.compile(r"^([vas])(\d+)$")

    _parsed = {}

    def _parse(self, op):
        try:
            return self._parsed[op]
        except KeyError:
            pass
        key, op = op, op.strip()
        m = self._re_rng.match(op)
        if m:
            r = m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        else:
            m = self._re_one.match(op)
            r = (m.group(1), int(m.group(2)), 1) if m else None
        self._parsed[key] = r
        return r

    @staticmethod
    def _literal(op):
        op = op.strip()
        if op.startswith("0x") or op.startswith("-0x"):
            return np.uint32(int(op, 16) & 0xFFFFFFFF)
        if re.match(r"^-?\d+$", op):
            return np.uint32(int(op) & 0xFFFFFFFF)
        if re.match(r"^-?\d+\.\d*$", op):
            return np.float32(float(op)).view(np.uint32)
        raise ValueError(f"operand {op!r}")

    def rd(self, op, idx=0):
        """32-bit read of register `idx` of operand `op` as a [64] uint32 vector (scalars broadcast)."""
        op = op.strip()
        if op == "vcc":
            raise ValueError("vcc as a 32-bit source")
        if op == "m0":
            return np.full(LANES, self.m0, np.uint32)
        if op.startswith("-") and self._parse(op[1:]) is not None:      # float source modifier: -v12
            return self.rd(op[1:], idx) ^ np.uint32(0x80000000)
        if op.startswith("|") and op.endswith("|"):                       # |v12|
            return self.rd(op[1:-1], idx) & np.uint32(0x7FFFFFFF)
        p = self._parse(op)
        if p is None:
            return np.full(LANES, self._literal(op), np.uint32)
        cls, base, n = p
        assert idx < n, (op, idx)
        if cls == "v":
            if self.track_uninit and (base + idx) not in self.init_v:
                self.read_before_write_v.add(base + idx)
            return self.v[base + idx].copy()
        if cls == "a":
            if self.track_uninit and (base + idx) not in self.init_a:
                self.read_before_write_v.add(1000 + base + idx)
            return self.a[base + idx].copy()
        return np.full(LANES, self.s[base + idx], np.uint32)

    def rd_s(self, op, idx=0):
        op = op.strip()
        if op == "m0":
            return np.uint32(self.m0)
        if op == "scc":
            return np.uint32(self.scc)
        p = self._parse(op)
        if p is None:
            return self._literal(op)
        cls, base, n = p
        assert cls == "s", op
        return self.s[base + idx]

    def rd64(self, op):
        p = self._parse(op)
        if p is None:
            return np.full(LANES, int(np.int32(self._literal(op))), np.int64).astype(np.uint64)
        cls, base, n = p
        assert n == 2, op
        lo, hi = self.rd(op, 0).astype(np.uint64), self.rd(op, 1).astype(np.uint64)
        return lo | (hi << np.uint64(32))

    def wr(self, op, val, idx=0, mask=None):
        cls, base, n = self._parse(op)
        val = np.asarray(val, dtype=np.uint32)
        if cls == "s":
            self.s[base + idx] = val if val.ndim == 0 else val[0]
            self.written_s.add(base + idx)
            return
        arr = self.v if cls == "v" else self.a
        (self.written_v if cls == "v" else self.written_a).add(base + idx)
        (self.init_v if cls == "v" else self.init_a).add(base + idx)
        if mask is None:
            arr[base + idx] = val
        else:
            arr[base + idx][mask] = np.broadcast_to(val, (LANES,))[mask]

    # ---------------------------------------------------------------- helpers
    def _halves(self, op, n_regs):
        """fp16 elements of an operand of n_regs registers: [64, 2 n_regs] float32."""
        p = self._parse(op)
        if p is not None and p[0] in "va" and not self.track_uninit:
            regs = (self.v if p[0] == "v" else self.a)[p[1]:p[1] + n_regs].T      # [64, n]
        else:
            regs = np.stack([self.rd(op, i) for i in range(n_regs)], axis=1)
        h = np.ascontiguousarray(regs).view(np.uint16).reshape(LANES, 2 * n_regs)  # little endian: low half first
        return h.view(np.float16).astype(np.float32)

    def _mfma(self, d, a, b, c, kper):
        n = kper // 2
        A = self._halves(a, n)   # lane l: row l % 16, k = kper (l / 16) + e
        B = self._halves(b, n)
        # [g][row][e] -> [row][g * kper + e]
        Am = A.reshape(4, 16, kper).transpose(1, 0, 2).reshape(16, 4 * kper)
        Bm = B.reshape(4, 16, kper).transpose(0, 2, 1).reshape(4 * kper, 16)
        with np.errstate(invalid="ignore", over="ignore"):   # (garbage operands of the negative controls)
            D = Am.astype(np.float64) @ Bm.astype(np.float64)
        if c.strip() != "0":
            # lane l, register r: C[4 (l / 16) + r][l % 16]
            cv = np.stack([_f32(self.rd(c, r)) for r in range(4)], axis=0)          # [r][lane]
            D = D + cv.reshape(4, 4, 16).transpose(1, 0, 2).reshape(16, 16)
        D = D.astype(np.float32)
        out = D.reshape(4, 4, 16).transpose(1, 0, 2).reshape(4, LANES)              # [r][lane]
        for r in range(4):
            self.wr(d, _u32(np.ascontiguousarray(out[r])), r)
        self.mfma_count += 1

    def _lds_rd(self, addr, nbytes):
        idx = addr[:, None].astype(np.int64) + np.arange(nbytes)[None, :]
        return self.lds[idx]  # [64, nbytes]

    def _lds_wr(self, addr, data):
        idx = addr[:, None].astype(np.int64) + np.arange(data.shape[1])[None, :]
        self.lds[idx] = data

    # ---------------------------------------------------------------- one instruction
    _decoded = {}   # instruction text -> (opcode, operands, modifiers): statements are loops, decode each line once

    @classmethod
    def _decode(cls, line):
        toks = line.split(None, 1)
        opc = toks[0]
        rest = toks[1] if len(toks) > 1 else ""
        mods = {}
        for m in re.finditer(r"\b(offset|op_sel|op_sel_hi|neg_lo|neg_hi):(\[[^\]]*\]|-?\d+)", rest):
            mods[m.group(1)] = m.group(2)
        rest_ops = re.sub(r"\b(offset|op_sel|op_sel_hi|neg_lo|neg_hi):(\[[^\]]*\]|-?\d+)", "", rest)
        # split operands on commas that are not inside brackets
        ops, depth, cur = [], 0, ""
        for ch in rest_ops:
            if ch == "[":
                depth += 1
            if ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur.strip())
        return opc, ops, mods

    def step(self, line, labels=None):
        """Execute one instruction.  Returns None, ("branch", label) or ("barrier",)."""
        line = line.strip()
        if not line or line.endswith(":"):
            return None
        self.count += 1
        d = self._decoded.get(line)
        if d is None:
            d = self._decoded[line] = self._decode(line)
        opc, ops, mods = d
        f = getattr(self, "i_" + opc, None)
        if f is None:
            raise NotImplementedError(f"asm_emu: {opc}  ({line})")
        return f(ops, mods)

    # ---- no-ops
    def i_s_nop(self, o, m): pass
    def i_s_waitcnt(self, o, m):
        txt = " ".join(o)
        mm = re.search(r"vmcnt\((\d+)\)", txt)
        if mm:
            self._retire(self.vm_q, int(mm.group(1)))
        mm = re.search(r"lgkmcnt\((\d+)\)", txt)
        if mm:
            self._retire(self.lgkm_q, int(mm.group(1)))
    def i_s_setprio(self, o, m): pass

    def i_s_barrier(self, o, m):
        return ("barrier",)

    # ---- scalar
    def i_s_mov_b32(self, o, m):
        v = self.rd_s(o[1])
        if o[0] == "m0":
            self.m0 = int(v)
        else:
            self.wr(o[0], v)

    def i_s_mov_b64(self, o, m):
        self.wr(o[0], self.rd_s(o[1], 0), 0)
        p = self._parse(o[1])
        self.wr(o[0], self.rd_s(o[1], 1) if p else np.uint32(0), 1)

    def _s_arith(self, o, fn, carry_in=False):
        a, b = int(self.rd_s(o[1])), int(self.rd_s(o[2]))
        r, scc = fn(a, b, self.scc)
        if o[0] == "m0":
            self.m0 = r & 0xFFFFFFFF
        else:
            self.wr(o[0], np.uint32(r & 0xFFFFFFFF))
        self.scc = scc

    def i_s_add_u32(self, o, m): self._s_arith(o, lambda a, b, c: (a + b, int(a + b > 0xFFFFFFFF)))
    def i_s_addc_u32(self, o, m): self._s_arith(o, lambda a, b, c: (a + b + c, int(a + b + c > 0xFFFFFFFF)))
    def i_s_sub_u32(self, o, m): self._s_arith(o, lambda a, b, c: (a - b, int(b > a)))
    def i_s_subb_u32(self, o, m): self._s_arith(o, lambda a, b, c: (a - b - c, int(b + c > a)))
    def i_s_mul_i32(self, o, m): self._s_arith(o, lambda a, b, c: (a * b, c))
    def i_s_lshl_b32(self, o, m): self._s_arith(o, lambda a, b, c: (a << (b & 31), int(((a << (b & 31)) & 0xFFFFFFFF) != 0)))
    def i_s_lshr_b32(self, o, m): self._s_arith(o, lambda a, b, c: (a >> (b & 31), int((a >> (b & 31)) != 0)))
    def i_s_and_b32(self, o, m): self._s_arith(o, lambda a, b, c: (a & b, int((a & b) != 0)))
    def i_s_or_b32(self, o, m): self._s_arith(o, lambda a, b, c: (a | b, int((a | b) != 0)))

    def i_s_cmp_eq_u32(self, o, m): self.scc = int(int(self.rd_s(o[0])) == int(self.rd_s(o[1])))
    def i_s_cmp_lg_u32(self, o, m): self.scc = int(int(self.rd_s(o[0])) != int(self.rd_s(o[1])))
    def i_s_cmp_gt_u32(self, o, m): self.scc = int(int(self.rd_s(o[0])) > int(self.rd_s(o[1])))
    def i_s_cmp_lt_u32(self, o, m): self.scc = int(int(self.rd_s(o[0])) < int(self.rd_s(o[1])))
    def i_s_bitcmp0_b32(self, o, m): self.scc = int(((int(self.rd_s(o[0])) >> (int(self.rd_s(o[1])) & 31)) & 1) == 0)
    def i_s_bitcmp1_b32(self, o, m): self.scc = int(((int(self.rd_s(o[0])) >> (int(self.rd_s(o[1])) & 31)) & 1) == 1)

    def i_s_cselect_b32(self, o, m):
        v = self.rd_s(o[1]) if self.scc else self.rd_s(o[2])
        self.wr(o[0], v)

    def i_s_branch(self, o, m): return ("branch", o[0])
    def i_s_cbranch_scc1(self, o, m): return ("branch", o[0]) if self.scc else None
    def i_s_cbranch_scc0(self, o, m): return ("branch", o[0]) if not self.scc else None

    def i_s_load_dword(self, o, m):
        base = int(self.rd_s(o[1], 0)) | (int(self.rd_s(o[1], 1)) << 32)
        off = int(self._literal(o[2]))
        a = base + off - self.gbase
        self.wr(o[0], self.gmem[a:a + 4].view(np.uint32)[0])

    def i_s_memtime(self, o, m):
        self.wr(o[0], np.uint32(self.count), 0)
        self.wr(o[0], np.uint32(0), 1)

    # ---- vector integer / moves
    def i_v_mov_b32(self, o, m): self.wr(o[0], self.rd(o[1]))

    def i_v_mov_b64(self, o, m):
        p = self._parse(o[1])
        self.wr(o[0], self.rd(o[1], 0), 0)
        self.wr(o[0], self.rd(o[1], 1) if p else np.zeros(LANES, np.uint32), 1)

    def i_v_accvgpr_read_b32(self, o, m): self.wr(o[0], self.rd(o[1]))
    def i_v_accvgpr_write_b32(self, o, m): self.wr(o[0], self.rd(o[1]))

    def i_v_mbcnt_lo_u32_b32(self, o, m):
        mask = int(np.int32(self.rd(o[0 + 1])[0]))  # -1
        assert mask == -1
        self.wr(o[0], np.minimum(self.lane, 32).astype(np.uint32) + self.rd(o[2]))

    def i_v_mbcnt_hi_u32_b32(self, o, m):
        self.wr(o[0], np.maximum(self.lane.astype(np.int64) - 32, 0).astype(np.uint32) + self.rd(o[2]))

    def i_v_lshlrev_b32(self, o, m): self.wr(o[0], self.rd(o[2]) << (self.rd(o[1]) & np.uint32(31)))
    def i_v_lshrrev_b32(self, o, m): self.wr(o[0], self.rd(o[2]) >> (self.rd(o[1]) & np.uint32(31)))
    def i_v_add_u32(self, o, m): self.wr(o[0], self.rd(o[1]) + self.rd(o[2]))
    def i_v_sub_u32(self, o, m): self.wr(o[0], self.rd(o[1]) - self.rd(o[2]))
    def i_v_subrev_u32(self, o, m): self.wr(o[0], self.rd(o[2]) - self.rd(o[1]))
    def i_v_and_b32(self, o, m): self.wr(o[0], self.rd(o[1]) & self.rd(o[2]))
    def i_v_or_b32(self, o, m): self.wr(o[0], self.rd(o[1]) | self.rd(o[2]))
    def i_v_add3_u32(self, o, m): self.wr(o[0], self.rd(o[1]) + self.rd(o[2]) + self.rd(o[3]))
    def i_v_mul_u32_u24(self, o, m): self.wr(o[0], (self.rd(o[1]) & np.uint32(0xFFFFFF)) * (self.rd(o[2]) & np.uint32(0xFFFFFF)))
    def i_v_lshl_or_b32(self, o, m): self.wr(o[0], (self.rd(o[1]) << (self.rd(o[2]) & np.uint32(31))) | self.rd(o[3]))
    def i_v_lshl_add_u32(self, o, m): self.wr(o[0], (self.rd(o[1]) << (self.rd(o[2]) & np.uint32(31))) + self.rd(o[3]))

    def i_v_lshl_add_u64(self, o, m):
        a = self.rd64(o[1])
        sh = np.uint64(int(self._literal(o[2])))
        c = self.rd64(o[3])
        r = (a << sh) + c
        self.wr(o[0], (r & np.uint64(0xFFFFFFFF)).astype(np.uint32), 0)
        self.wr(o[0], (r >> np.uint64(32)).astype(np.uint32), 1)

    def i_v_readfirstlane_b32(self, o, m): self.wr(o[0], self.rd(o[1])[0])

    def i_v_cmp_eq_u32(self, o, m):
        assert o[0] == "vcc"
        self.vcc = self.rd(o[1]) == self.rd(o[2])

    def i_v_cmp_lt_u32(self, o, m):
        assert o[0] == "vcc"
        self.vcc = self.rd(o[1]) < self.rd(o[2])

    def i_v_cndmask_b32(self, o, m):
        assert o[3] == "vcc"
        self.wr(o[0], np.where(self.vcc, self.rd(o[2]), self.rd(o[1])))

    def i_v_bfe_i32(self, o, m):
        x, off, w = self.rd(o[1]), self.rd(o[2]) & np.uint32(31), self.rd(o[3]) & np.uint32(31)
        r = ((x.astype(np.uint64) >> off.astype(np.uint64)) & ((np.uint64(1) << w.astype(np.uint64)) - np.uint64(1))).astype(np.int64)
        sign = (r >> (w.astype(np.int64) - 1)) & 1
        r = np.where((w > 0) & (sign == 1), r - (np.int64(1) << w.astype(np.int64)), r)
        self.wr(o[0], (r & 0xFFFFFFFF).astype(np.uint32))

    def i_v_bfi_b32(self, o, m):
        s0, s1, s2 = self.rd(o[1]), self.rd(o[2]), self.rd(o[3])
        self.wr(o[0], (s0 & s1) | (~s0 & s2))

    # ---- vector float
    def _fop(self, o, fn, n=2):
        srcs = [_f32(self.rd(x)) for x in o[1:1 + n]]
        with np.errstate(all="ignore"):
            self.wr(o[0], _u32(fn(*srcs)))

    def i_v_add_f32(self, o, m): self._fop(o, lambda a, b: a + b)
    def i_v_sub_f32(self, o, m): self._fop(o, lambda a, b: a - b)
    def i_v_mul_f32(self, o, m): self._fop(o, lambda a, b: a * b)
    def i_v_max_f32(self, o, m): self._fop(o, np.maximum)
    def i_v_min_f32(self, o, m): self._fop(o, np.minimum)
    def i_v_fma_f32(self, o, m): self._fop(o, lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32), 3)
    def i_v_rsq_f32(self, o, m): self._fop(o, lambda a: (1.0 / np.sqrt(a.astype(np.float64))).astype(np.float32), 1)
    def i_v_rcp_f32(self, o, m): self._fop(o, lambda a: (1.0 / a.astype(np.float64)).astype(np.float32), 1)
    def i_v_exp_f32(self, o, m): self._fop(o, lambda a: np.exp2(a.astype(np.float64)).astype(np.float32), 1)

    def i_v_fmamk_f32(self, o, m):   # d = s0 * K + s1
        a, k, b = _f32(self.rd(o[1])), _f32(self.rd(o[2])), _f32(self.rd(o[3]))
        self.wr(o[0], _u32((a.astype(np.float64) * k.astype(np.float64) + b.astype(np.float64)).astype(np.float32)))

    def i_v_fmaak_f32(self, o, m):   # d = s0 * s1 + K
        a, b, k = _f32(self.rd(o[1])), _f32(self.rd(o[2])), _f32(self.rd(o[3]))
        self.wr(o[0], _u32((a.astype(np.float64) * b.astype(np.float64) + k.astype(np.float64)).astype(np.float32)))

    def _pk_src(self, op, i):
        p = self._parse(op)
        if p is None or p[2] == 1:
            return _f32(self.rd(op, 0))   # scalar / constant / single register: both halves read it
        return _f32(self.rd(op, i))

    def i_v_pk_fma_f32(self, o, m):
        for i in range(2):
            a, b, c = self._pk_src(o[1], i), self._pk_src(o[2], i), self._pk_src(o[3], i)
            self.wr(o[0], _u32((a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)), i)

    def i_v_pk_mul_f32(self, o, m):
        for i in range(2):
            self.wr(o[0], _u32(self._pk_src(o[1], i) * self._pk_src(o[2], i)), i)

    def i_v_pk_add_f32(self, o, m):
        for i in range(2):
            self.wr(o[0], _u32(self._pk_src(o[1], i) + self._pk_src(o[2], i)), i)

    def i_v_cvt_pk_f16_f32(self, o, m):
        with np.errstate(all="ignore"):
            lo = _f32(self.rd(o[1])).astype(np.float16).view(np.uint16).astype(np.uint32)
            hi = _f32(self.rd(o[2])).astype(np.float16).view(np.uint16).astype(np.uint32)
        self.wr(o[0], lo | (hi << np.uint32(16)))

    def i_v_pk_max_f16(self, o, m):
        a = np.ascontiguousarray(self.rd(o[1])).view(np.float16).reshape(LANES, 2)
        b = np.ascontiguousarray(self.rd(o[2])).view(np.float16).reshape(LANES, 2)
        r = np.maximum(a, b)
        self.wr(o[0], np.ascontiguousarray(r).view(np.uint32).reshape(LANES))

    def i_v_fma_mix_f32(self, o, m):
        sel = [int(x) for x in mods_list(m.get("op_sel", "[0,0,0]"))]
        sel_hi = [int(x) for x in mods_list(m.get("op_sel_hi", "[0,0,0]"))]
        vals = []
        for i in range(3):
            op = o[1 + i]
            neg = False
            if op.startswith("-") and self._parse(op[1:]) is not None:
                neg, op = True, op[1:]
            raw = self.rd(op)
            if sel_hi[i]:
                h = (raw >> np.uint32(16)) if sel[i] else (raw & np.uint32(0xFFFF))
                val = h.astype(np.uint16).view(np.float16).astype(np.float64)
            else:
                val = _f32(raw).astype(np.float64)
            vals.append(-val if neg else val)
        self.wr(o[0], _u32((vals[0] * vals[1] + vals[2]).astype(np.float32)))

    # ---- cross-lane
    def i_v_permlane16_swap_b32(self, o, m):
        d, s = self.rd(o[0]), self.rd(o[1])
        d2, s2 = d.copy(), s.copy()
        for base in (0, 32):   # odd rows of vdst <-> even rows of src
            d2[base + 16:base + 32] = s[base:base + 16]
            s2[base:base + 16] = d[base + 16:base + 32]
        self.wr(o[0], d2)
        self.wr(o[1], s2)

    def i_v_permlane32_swap_b32(self, o, m):
        d, s = self.rd(o[0]), self.rd(o[1])
        d2, s2 = d.copy(), s.copy()
        d2[32:64] = s[0:32]
        s2[0:32] = d[32:64]
        self.wr(o[0], d2)
        self.wr(o[1], s2)

    # ---- matrix
    def i_v_mfma_f32_16x16x16_f16(self, o, m): self._mfma(o[0], o[1], o[2], o[3], 4)
    def i_v_mfma_f32_16x16x32_f16(self, o, m): self._mfma(o[0], o[1], o[2], o[3], 8)

    # ---- LDS
    def _ds_rd(self, o, m, nreg):
        addr = self.rd(o[1]) + np.uint32(int(m.get("offset", 0)))

        def land():
            data = self._lds_rd(addr, 4 * nreg)
            regs = np.ascontiguousarray(data).view(np.uint32).reshape(LANES, nreg)
            for i in range(nreg):
                self.wr(o[0], regs[:, i], i)
        if self.late_lds:
            self.lgkm_q.append(("read", land))
        else:
            land()

    def _ds_wr(self, o, m, nreg):
        addr = self.rd(o[0]) + np.uint32(int(m.get("offset", 0)))
        regs = np.stack([self.rd(o[1], i) for i in range(nreg)], axis=1)
        self._lds_wr(addr, np.ascontiguousarray(regs).view(np.uint8).reshape(LANES, 4 * nreg))
        if self.late_lds:
            self.lgkm_q.append(("write", lambda: None))   # (returns in order with the reads)

    def i_ds_read_b128(self, o, m): self._ds_rd(o, m, 4)
    def i_ds_read_b64(self, o, m): self._ds_rd(o, m, 2)
    def i_ds_read_b32(self, o, m): self._ds_rd(o, m, 1)
    def i_ds_write_b128(self, o, m): self._ds_wr(o, m, 4)
    def i_ds_write_b64(self, o, m): self._ds_wr(o, m, 2)
    def i_ds_write_b32(self, o, m): self._ds_wr(o, m, 1)

    # ---- global
    def _gaddr(self, o, m):
        """(vaddr, saddr | off) addressing: 64-bit VGPR address, or SGPR base + 32-bit VGPR offset."""
        off = int(m.get("offset", 0))
        if o[-1] == "off":
            return self.rd64(o[-2]).astype(np.int64) + off - self.gbase
        base = int(self.rd_s(o[-1], 0)) | (int(self.rd_s(o[-1], 1)) << 32)
        return self.rd(o[-2]).astype(np.int64) + base + off - self.gbase

    def i_global_load_dwordx4(self, o, m):
        a = self._gaddr(o[1:], m)
        idx = a[:, None] + np.arange(16)[None, :]
        regs = np.ascontiguousarray(self.gmem[idx]).view(np.uint32).reshape(LANES, 4)

        def land():
            for i in range(4):
                self.wr(o[0], regs[:, i], i)
        if self.late_vmem:
            self.vm_q.append(("reg", land))
        else:
            land()

    def i_global_load_lds_dwordx4(self, o, m):
        a = self._gaddr(o, m)
        idx = a[:, None] + np.arange(16)[None, :]
        lds_addr = (np.int64(self.m0) + int(m.get("offset", 0)) + 16 * self.lane.astype(np.int64))   # (M0 is read at issue)
        data = self.gmem[idx].copy()
        if self.late_vmem:
            self.vm_q.append(("lds", lambda: self._lds_wr(lds_addr, data)))
        else:
            self._lds_wr(lds_addr, data)

    def i_global_store_dwordx2(self, o, m):
        # (vaddr, vdata, saddr | off)
        off = int(m.get("offset", 0))
        if o[2] == "off":
            a = self.rd64(o[0]).astype(np.int64) + off - self.gbase
        else:
            base = int(self.rd_s(o[2], 0)) | (int(self.rd_s(o[2], 1)) << 32)
            a = self.rd(o[0]).astype(np.int64) + base + off - self.gbase
        regs = np.stack([self.rd(o[1], i) for i in range(2)], axis=1)
        data = np.ascontiguousarray(regs).view(np.uint8).reshape(LANES, 8)
        for l in range(LANES):   # (lanes usually share the address: last lane wins, as on the chip)
            self.gmem[a[l]:a[l] + 8] = data[l]


def mods_list(s):
    return [x for x in s.strip("[]").split(",") if x != ""]


def prepare(lines, operands):
    """Substitute the inline-asm operands (%[name] -> register text) and the %= label suffix; returns (lines, label -> index)."""
    out = []
    for l in lines:
        for k, v in operands.items():
            l = l.replace(f"%[{k}]", v)
        l = l.replace("%=", "0")
        if "%[" in l:
            raise KeyError(f"unbound operand in: {l}")
        out.append(l)
    labels = {l[:-1]: i for i, l in enumerate(out) if l.endswith(":")}
    return out, labels


def run(wave, lines, operands=None, max_steps=50_000_000):
    """Run one wave to the end of `lines` (barriers are ignored)."""
    lines, labels = prepare(lines, operands or {})
    pc = 0
    n = 0
    while pc < len(lines):
        r = wave.step(lines[pc], labels)
        pc += 1
        if r and r[0] == "branch":
            pc = labels[r[1]]
        n += 1
        assert n < max_steps, "runaway"
    return wave


def run_waves(waves, lines, operands_per_wave, max_steps=200_000_000, order=None):
    """Run several waves over one LDS in lock step: each wave runs to its next s_barrier (or the end), then the next -
    in index order, or in `order`."""
    progs = [prepare(lines, ops) for ops in operands_per_wave]
    pcs = [0] * len(waves)
    done = [False] * len(waves)
    n = 0
    while not all(done):
        at_barrier = []
        for w in (order if order is not None else range(len(waves))):
            wave = waves[w]
            if done[w]:
                continue
            code, labels = progs[w]
            while True:
                if pcs[w] >= len(code):
                    done[w] = True
                    break
                r = wave.step(code[pcs[w]], labels)
                pcs[w] += 1
                n += 1
                assert n < max_steps, "runaway"
                if r and r[0] == "branch":
                    pcs[w] = labels[r[1]]
                elif r and r[0] == "barrier":
                    at_barrier.append(w)
                    break
        live = [w for w in range(len(waves)) if not done[w]]
        assert not live or sorted(at_barrier) == live, f"waves {live} alive but only {at_barrier} at a barrier: deadlock"
    return waves
