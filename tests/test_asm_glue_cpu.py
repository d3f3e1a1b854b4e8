"""The generated glue of the encoder-stack asm statements (tools/gen_h3_enc_asm.py), executed on the CPU by the functional
emulator tools/asm_emu.py and held to a numpy restatement of what it is meant to compute: residual-seeded accumulators ->
LayerNorm -> padding select -> fp16 hi/lo split into the FFN's operand registers / the transposed copy of x for the next
attention block / the out-MLP's operand images, and the next block's accumulator seeds.

This is a check of the generators' register maps and data layouts (a wrong index there is silent corruption on the chip),
not of timing: wait states and s_waitcnt counts are outside the emulator.  The 48-token statement (known good on hardware
since r03) runs through the same checks, which pins the emulator's own instruction semantics.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_emu as E  # noqa: E402

SL, PRIV = 120 * 1024, 32 * 1024   # LDS addresses of the side block and of the wave-private block in these tests
XT, WAVE = 8 * 1024, 2              # wide layout: the workgroup's shared X^T tile; the wave these tests play (token slots 96..143)
OPERANDS = {"sl": "s10", "priv": "s11", "padm": "v250", "xt": "s12", "wave": "s13"}


def load_gen(name, argv):
    old = sys.argv
    sys.argv = [name] + list(argv)
    try:
        spec = importlib.util.spec_from_file_location(name + "_" + "_".join(a.strip("-").replace("=", "") for a in argv),
                                                      os.path.join(ROOT, "tools", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = old
    return m


def f32(x):
    return np.asarray(x, np.float32).view(np.uint32)


class Setup:
    """One wave with random accumulators, a random side block and the state generate() establishes in front of the glue."""

    def __init__(self, enc, seed, pad_tile=None):
        self.enc, self.NT = enc, enc.NT
        rng = np.random.default_rng(seed)
        self.lds = np.zeros(160 * 1024, np.uint8)
        self.w = w = E.Wave(lds=self.lds)
        self.side = rng.standard_normal(1280).astype(np.float32)   # (dense: in_proj / out_proj biases behind the 656)
        self.lds[SL:SL + 1280 * 4] = self.side.view(np.uint8)
        lane = np.arange(64)
        self.g, self.i16 = lane // 16, lane % 16
        w.s[10], w.s[11], w.s[12], w.s[13] = SL, PRIV, XT, WAVE
        if enc.DENSE:
            w.s[enc.S_SLCUR], w.s[enc.S_SLOTHER] = SL, SL + 5120
        self.pad = np.zeros((self.NT, 16), bool)
        if pad_tile is not None:
            self.pad[pad_tile, 11:] = True           # tokens 11..15 of that tile are padding
        padmask = np.zeros(64, np.uint32)
        for jt in range(self.NT):
            padmask |= (self.pad[jt][self.i16].astype(np.uint32) << jt)
        w.v[250] = padmask
        w.s[enc.S_PADT] = sum(1 << jt for jt in range(self.NT) if self.pad[jt].any())
        self.sc_a, self.sc_f, self.sc_next = np.float32(2.0 ** -6), np.float32(2.0 ** -4), np.float32(2.0 ** -7)
        w.s[enc.S_EPS] = f32(1e-5)
        w.s[enc.S_IA] = f32(1 / self.sc_a)
        w.s[enc.S_IF] = f32(1 / self.sc_f)
        if not enc.STATELESS:   # the 48-token statement keeps these in registers across the layer loop
            w.v[enc.V_SLG] = SL + 16 * self.g
            w.v[enc.V_PRIV16] = PRIV + 16 * lane
            w.v[enc.V_PRIV8] = PRIV + 8 * lane
            w.v[enc.V_PAD] = padmask
            w.v[enc.V_ONE] = w.v[enc.V_ONE + 1] = f32(1.0)
            idb = np.zeros((64, 4), np.float16)
            for e in range(4):
                idb[:, e] = (self.i16 == 4 * self.g + e)
            w.v[enc.V_IDB:enc.V_IDB + 2] = np.ascontiguousarray(idb).view(np.uint32).reshape(64, 2).T
        self.t = (rng.standard_normal((8, self.NT, 4, 64)) * 40).astype(np.float32)   # accumulators [ft][jt][r][lane]
        for ft in range(8):
            for jt in range(self.NT):
                for r in range(4):
                    w.a[enc.ACC(ft, jt) + r] = self.t[ft, jt, r].view(np.uint32)

    def tokens(self, regs):
        """[ft][jt][r][lane] register tiles -> [jt][token 16][feature 128]."""
        X = np.zeros((self.NT, 16, 128), np.float64)
        for ft in range(8):
            for jt in range(self.NT):
                for r in range(4):
                    X[jt, self.i16, 16 * ft + 4 * self.g + r] = regs[ft, jt, r]
        return X

    def layer_norm(self, scale, w_off, b_off, pre_bias=None):
        X = self.tokens(self.t) * np.float64(scale)
        if pre_bias is not None:
            X = X + self.side[pre_bias:pre_bias + 128]
        mean = X.mean(-1, keepdims=True)
        var = ((X - mean) ** 2).mean(-1, keepdims=True)
        Y = (X - mean) / np.sqrt(var + 1e-5) * self.side[w_off:w_off + 128] + self.side[b_off:b_off + 128]
        Y[self.pad] = 0
        return Y

    def acc_tokens(self):
        regs = np.zeros((8, self.NT, 4, 64), np.float32)
        for ft in range(8):
            for jt in range(self.NT):
                for r in range(4):
                    regs[ft, jt, r] = self.w.a[self.enc.ACC(ft, jt) + r].view(np.float32)
        return self.tokens(regs)


def split(v):
    hi = np.asarray(v, np.float32).astype(np.float16)
    lo = (np.asarray(v, np.float32) - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def halves(regs):
    """[n][64] uint32 registers -> [64][2 n] float64 (fp16 elements, low half first)."""
    return np.ascontiguousarray(np.asarray(regs).T).view(np.float16).reshape(64, -1).astype(np.float64)


def check_split(got_hi, got_lo, want, h1, what):
    """hi + lo must carry `want` to fp32-class accuracy (hi alone to fp16 accuracy in the single-MFMA build)."""
    scale = np.abs(want).max() + 1e-30
    if h1:
        assert np.abs(got_hi - want).max() / scale < 1.5e-3, what
    else:
        assert np.abs(got_hi + got_lo - want).max() / scale < 2e-6, (what, np.abs(got_hi + got_lo - want).max() / scale)
        assert np.abs(got_hi - want).max() / scale < 1.5e-3, what


VARIANTS = [(), ("--mode=windowed",), ("--h1",), ("--nt=4",), ("--nt=4", "--h1"), ("--wide",), ("--wide", "--h1"), ("--wide", "--ng=6"),
            ("--dense",), ("--dense", "--h1"), ("--dense", "--nt=4")]


@pytest.mark.parametrize("argv", VARIANTS, ids=lambda a: " ".join(a) or "nt3")
@pytest.mark.parametrize("pad_tile", [None, 2])
def test_g1_layernorm_ffn_operands_and_seeds(argv, pad_tile):
    enc = load_gen("gen_h3_enc_asm", argv)
    s = Setup(enc, 1, pad_tile)
    E.run(s.w, enc.g1(), OPERANDS)
    Y = s.layer_norm(s.sc_a, 0, 128, pre_bias=1040 if enc.DENSE else None)   # dense: + out_proj's bias (csrc H3D_OUTB)
    n_vg = 2 if enc.NT4 else 3
    for ks in range(4):
        for jt in range(s.NT):
            file = s.w.v if ks < n_vg else s.w.a
            hi = halves(file[enc.ffn.XB(ks, jt, "h"):enc.ffn.XB(ks, jt, "h") + 4])   # [lane][e]: feature 32 ks + 16 (e / 4) + 4 g + e % 4
            lo = halves(file[enc.ffn.XB(ks, jt, "l"):enc.ffn.XB(ks, jt, "l") + 4])
            want = np.zeros((64, 8))
            for e in range(8):
                want[:, e] = Y[jt, s.i16, 32 * ks + 16 * (e // 4) + 4 * s.g + e % 4]
            check_split(hi, lo, want, enc.H1, ("xb", ks, jt))
    seeds = s.acc_tokens()
    want = (Y + s.side[256:384]) / np.float64(s.sc_f)
    assert np.abs(seeds - want).max() / np.abs(want).max() < 1e-6


def xt_images(s, enc):
    """The transposed copy the attention block will read: [part][jt][token 16][feature 128] from the AGPR images (48-token
    statement) or the wave-private LDS block (64-token statement)."""
    out = np.zeros((2, s.NT, 16, 128))
    if enc.DENSE:
        # no transposed copy: the attention block's operands are the split activations in a96..a191 (always hi and lo)
        for ks in range(4):
            for jt in range(s.NT):
                for part, name in enumerate("hl"):
                    file = s.w.v if enc.attn.XB_CLS(ks, jt) == "v" else s.w.a   # (64-token build: one operand pair lives in VGPRs)
                    h = halves(file[enc.attn.XB(ks, jt, name):enc.attn.XB(ks, jt, name) + 4])
                    for e in range(8):
                        out[part, jt, s.i16, 32 * ks + 16 * (e // 4) + 4 * s.g + e % 4] = h[:, e]
        return out
    for ft in range(8):
        for part in range(1 if enc.H1 else 2):
            if enc.WIDE:
                # the shared tile [feature][192 tokens + pad]: rows of XT_ROW bytes, the lo half XT_LO behind the hi half
                for f in range(16):
                    a = XT + part * enc.attn.XT_LO + (16 * ft + f) * enc.attn.XT_ROW + 2 * 48 * WAVE
                    row = s.lds[a:a + 96].view(np.float16).astype(np.float64)   # this wave's 48 tokens
                    out[part, :, :, 16 * ft + f] = row.reshape(s.NT, 16)
            elif enc.NT4:
                for pair in range(2):
                    a = PRIV + 2 * enc.XT_IMG * ft + enc.XT_IMG * part + 1024 * pair
                    h = s.lds[a:a + 1024].view(np.float16).reshape(64, 8).astype(np.float64)   # [lane][e]
                    for e in range(8):
                        out[part, 2 * pair + e // 4, 4 * s.g + e % 4, 16 * ft + s.i16] = h[:, e]
            else:
                base = enc.attn.XT_AGPR + 12 * ft
                n01, n2 = (("a0h", "a1h"), ("a0l", "a1l"))[part]
                h = halves(s.w.a[base + enc.attn.XT_OFF[n01]:base + enc.attn.XT_OFF[n01] + 4])
                for e in range(8):
                    out[part, e // 4, 4 * s.g + e % 4, 16 * ft + s.i16] = h[:, e]
                h = halves(s.w.a[base + enc.attn.XT_OFF[n2]:base + enc.attn.XT_OFF[n2] + 2])
                for e in range(4):
                    out[part, 2, 4 * s.g + e, 16 * ft + s.i16] = h[:, e]
    return out


@pytest.mark.parametrize("argv", VARIANTS, ids=lambda a: " ".join(a) or "nt3")
def test_g2_layernorm_transposed_copy_and_seeds(argv):
    enc = load_gen("gen_h3_enc_asm", argv)
    s = Setup(enc, 2, pad_tile=s_pad(enc))
    s.w.s[enc.S_IA] = f32(1 / s.sc_next)      # (by then S_IA holds the NEXT layer's attention scale)
    if not enc.STATELESS:
        s.w.v[enc.V_C2] = s.w.v[enc.V_C2 + 1] = f32(1 / s.sc_next)
    E.run(s.w, enc.g2(False), OPERANDS)
    Y = s.layer_norm(s.sc_f, 384, 512)
    xt = xt_images(s, enc)
    for jt in range(s.NT):
        hi, lo = xt[0, jt], xt[1, jt]
        check_split(hi, lo, Y[jt], enc.H1A, ("xt", jt))
    seeds = s.acc_tokens()
    want = Y / np.float64(s.sc_next)
    assert np.abs(seeds - want).max() / np.abs(want).max() < 1e-6


def s_pad(enc):
    return enc.NT - 1


@pytest.mark.parametrize("argv", VARIANTS, ids=lambda a: " ".join(a) or "nt3")
def test_g2_last_layer_leaves_the_out_mlp_operand_images(argv):
    enc = load_gen("gen_h3_enc_asm", argv)
    s = Setup(enc, 3, pad_tile=0)
    E.run(s.w, enc.g2(True), OPERANDS)
    Y = s.layer_norm(s.sc_f, 384, 512)
    for ks in range(4):
        for jt in range(s.NT):
            img = {}
            for part in range(1 if enc.H1 else 2):
                a = PRIV + 1024 * ((ks * s.NT + jt) * 2 + part)
                img[part] = s.lds[a:a + 1024].view(np.float16).reshape(64, 8).astype(np.float64)
            want = np.zeros((64, 8))
            for e in range(8):
                want[:, e] = Y[jt, s.i16, 32 * ks + 16 * (e // 4) + 4 * s.g + e % 4]
            check_split(img[0], img.get(1, 0 * img[0]), want, enc.H1, ("out image", ks, jt))


@pytest.mark.parametrize("argv", VARIANTS, ids=lambda a: " ".join(a) or "nt3")
def test_entry_reads_x_and_builds_the_first_transposed_copy(argv):
    """generate() up to the layer loop: x in as register images from the wave-private block -> transposed copy + seeds."""
    enc = load_gen("gen_h3_enc_asm", argv)
    s = Setup(enc, 4)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((8, s.NT, 4, 64)) * 3).astype(np.float32)
    for ft in range(8):
        for jt in range(s.NT):
            a = PRIV + 1024 * (ft * s.NT + jt)
            s.lds[a:a + 1024] = np.ascontiguousarray(x[ft, jt].T).view(np.uint8).reshape(-1)   # lane-major: 16 bytes per lane
    # the operands of the statement
    gm = np.zeros(4096, np.uint8)
    scales = np.array([s.sc_a, 0, s.sc_f], np.float32)
    if enc.DENSE:   # [0] is the in_proj scale there; the residual scale (out_proj's) sits 2 + 3 L floats behind the pointer
        scales = np.array([77.0, 0, s.sc_f] + [0] * 8 + [s.sc_a], np.float32)
    gm[256:256 + 4 * len(scales)] = scales.view(np.uint8)
    s.w.gmem, s.w.gbase = gm, 0x1000
    s.w.s[20], s.w.s[21] = 0x1000 + 256, 0
    ops = dict(OPERANDS, layers="3", padt="0", scales="s[20:21]", eps="s22", sf="s[24:25]", side="s[26:27]", dump="s[28:29]",
               stampen="0", sidestride="5120")
    s.w.s[22] = f32(1e-5)
    lines = enc.generate()
    entry = lines[:lines.index(".Lenc_layer_%=:")]
    E.run(s.w, entry, ops)
    X = s.tokens(x)
    xt = xt_images(s, enc)
    for jt in range(s.NT):
        check_split(xt[0, jt], xt[1, jt], X[jt], enc.H1A, ("xt", jt))
    seeds = s.acc_tokens()
    want = X / np.float64(s.sc_a)
    assert np.abs(seeds - want).max() / np.abs(want).max() < 1e-6
    assert s.w.s[enc.S_IF].view(np.float32) == np.float32(1 / s.sc_f)


@pytest.mark.parametrize("argv", [("--nt=4",), ("--nt=4", "--h1"), ("--wide",), ("--wide", "--ng=6", "--h1"), ("--dense",), ("--dense", "--h1"),
                                  ("--dense", "--nt=4")],
                         ids=lambda a: " ".join(a))
def test_the_64_token_glue_keeps_nothing_in_registers_across_the_embedded_blocks(argv):
    """Every glue phase of the 64-token statement (and of the wide / dense ones, which take the same form) must work from a
    register file the embedded blocks have overwritten:
    poison all VGPRs and the AGPRs the blocks own for their operands, then run each phase."""
    enc = load_gen("gen_h3_enc_asm", argv)
    for phase, make in (("g1", enc.g1), ("g2", lambda: enc.g2(False)), ("g2 last", lambda: enc.g2(True))):
        s = Setup(enc, 7)
        s.w.v[:250] = 0x7FC12345   # NaN pattern
        s.w.a[128 if enc.NT4 else 96:] = 0x7FC12345
        E.run(s.w, make(), OPERANDS)
        assert np.isfinite(s.acc_tokens()).all() or phase == "g2 last", phase
        if phase == "g1":
            Y = s.layer_norm(s.sc_a, 0, 128, pre_bias=1040 if enc.DENSE else None)
            want = (Y + s.side[256:384]) / np.float64(s.sc_f)
            assert np.abs(s.acc_tokens() - want).max() / np.abs(want).max() < 1e-6
