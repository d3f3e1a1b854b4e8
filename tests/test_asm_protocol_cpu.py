"""The weight-stage hand-off of the generated FFN statements (tools/gen_h3_ffn_asm.py) under ADVERSARIAL completion of the
asynchronous memory operations, on the CPU (tools/asm_emu.py `late_vmem` / `late_lds`).

A statement streams its weights through a ring of LDS stage buffers: LDS-DMA pieces land whenever they land, every wave
reads every buffer, and the only things that order the two are counted `s_waitcnt`s and workgroup barriers.  r04's stress
runs found a hole in exactly this protocol after every functional test had passed.  Here the four waves of a workgroup run
the whole statement - prologue, chunk loop, both tails - over a real stage stream in (emulated) global memory, while
  * every LDS-DMA piece and register load lands only when the issuing wave's own `vmcnt` wait forces it, or
  * every ds_read is sampled only when the wave's own `lgkmcnt` wait forces it (so a refill that overtakes it shows),
with the waves advanced to each barrier in both orders (wave 0 far ahead / far behind).  The result must equal a numpy
restatement of the chunked MLP computed from the same packed stream.  Covered: the split-fp16 and the single-MFMA statements on
48- and 64-token waves, and the six-slot ring with its barrier-free stages (r05).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_emu as E  # noqa: E402
from test_asm_glue_cpu import load_gen  # noqa: E402

STAGE, TILES = 9216, 8192
RING_LDS, PRIV_LDS, PRIV_STRIDE = 0, 64 * 1024, 40 * 1024   # ring base; wave-private blocks (<= 32 images of 1 KiB each)
OPS = {"cur": "s40", "gn": "v[252:253]", "ring": "s41", "wave": "s42", "priv": "s43", "chunks": "s44"}
GBASE = 1 << 20


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16)


def split(x):
    hi = f16(x)
    lo = f16(np.asarray(x, np.float32) - hi.astype(np.float32))
    return hi, lo


def tile_bytes(W, row0, col0):
    """One 16 x 32 fp16 tile in the A-operand lane order of the packed stream (csrc h3_pack_block_kernel): element (lane, e)
    = W[row0 + lane % 16][col0 + 16 (e / 4) + 4 (lane / 16) + e % 4]."""
    out = np.zeros((64, 8), np.float16)
    for l in range(64):
        for e in range(8):
            out[l, e] = W[row0 + l % 16, col0 + 16 * (e // 4) + 4 * (l // 16) + e % 4]
    return out.view(np.uint8).reshape(-1)


def image_bytes(X, k0, t0):
    """One B-operand image (16 bytes per lane) of activations X[feature][token]: (lane, e) = X[k0 + 16 (e / 4) + 4 (lane / 16)
    + e % 4][t0 + lane % 16]."""
    out = np.zeros((64, 8), np.float16)
    for l in range(64):
        for e in range(8):
            out[l, e] = X[k0 + 16 * (e // 4) + 4 * (l // 16) + e % 4, t0 + l % 16]
    return out.view(np.uint8).reshape(-1)


class Case:
    """A chunked MLP  y = W2 . act(scale (W1q . x) + b1)  of `chunks` 32-unit chunks, packed as the stage stream the statement
    reads (csrc h3_pack_weights: A(0) | A(c+1) B(c) ... | B(n-1)).  Shapes (gen_h3_ffn_asm.py SHAPES): ffn 128 -> ReLU -> 128,
    in 64 -> SiLU -> 128, out 128 -> SiLU -> 16."""

    def __init__(self, gen, chunks, seed, shape="ffn"):
        self.gen, self.n, self.shape = gen, chunks, shape
        self.H1, self.NT = gen.H1, gen.NT
        self.ks_in, self.ot_out, self.silu = (gen.SHAPES[shape][k] for k in ("ks_in", "ot_out", "silu"))
        ks_in, ot_out = self.ks_in, self.ot_out
        rng = np.random.default_rng(seed)
        F = 32 * chunks
        self.scale = np.float32(1.0 if gen.H1 else 2.0 ** -3)
        W1 = rng.standard_normal((F, 32 * ks_in)).astype(np.float32) * (0.1 if gen.H1 else 0.8)
        W2 = rng.standard_normal((16 * ot_out, F)).astype(np.float32) * 0.5
        self.b1 = rng.standard_normal(F).astype(np.float32) * 0.3
        self.W1h, self.W1l = split(W1)
        self.W2h, self.W2l = split(W2)
        T = 16 * self.NT
        self.X = [rng.standard_normal((32 * ks_in, T)).astype(np.float32) for _ in range(4)]        # per wave
        # ---- the stream
        fa = 1 if (self.H1 or ks_in == 2) else 2
        stages = []

        def a_stage(ch, o):
            st = np.zeros(STAGE, np.uint8)
            if self.H1:     # tile ks_in o + ks, both o in one stage
                for oo in range(2):
                    for ks in range(ks_in):
                        st[1024 * (ks_in * oo + ks):][:1024] = tile_bytes(self.W1h, 32 * ch + 16 * oo, 32 * ks)
            elif ks_in == 2:   # in-MLP: one A stage, pair q = 2 o + ks
                for q in range(4):
                    st[2048 * q:][:1024] = tile_bytes(self.W1h, 32 * ch + 16 * (q // 2), 32 * (q % 2))
                    st[2048 * q + 1024:][:1024] = tile_bytes(self.W1l, 32 * ch + 16 * (q // 2), 32 * (q % 2))
            else:           # pair ks of rows 32 ch + 16 o: hi KiB, lo KiB
                for ks in range(4):
                    st[2048 * ks:][:1024] = tile_bytes(self.W1h, 32 * ch + 16 * o, 32 * ks)
                    st[2048 * ks + 1024:][:1024] = tile_bytes(self.W1l, 32 * ch + 16 * o, 32 * ks)
            if o == 0:
                st[TILES:TILES + 128] = self.b1[32 * ch:32 * ch + 32].view(np.uint8)
                st[TILES + 128:TILES + 132] = np.array([self.scale], np.float32).view(np.uint8)
            return st

        def b_stage(ch, hf):
            st = np.zeros(STAGE, np.uint8)
            if self.H1:
                for ot in range(ot_out):
                    st[1024 * ot:][:1024] = tile_bytes(self.W2h, 16 * ot, 32 * ch)
            else:
                for p in range(min(4, ot_out - 4 * hf)):
                    st[2048 * p:][:1024] = tile_bytes(self.W2h, 16 * (4 * hf + p), 32 * ch)
                    st[2048 * p + 1024:][:1024] = tile_bytes(self.W2l, 16 * (4 * hf + p), 32 * ch)
            return st

        nb = 1 if self.H1 else (ot_out + 3) // 4
        for o in range(fa):
            stages.append(a_stage(0, o))
        for ch in range(chunks - 1):
            for o in range(fa):
                stages.append(a_stage(ch + 1, o))
            for hf in range(nb):
                stages.append(b_stage(ch, hf))
        for hf in range(nb):
            stages.append(b_stage(chunks - 1, hf))
        self.n_stages = len(stages)
        for _ in range(gen.AHEAD + 2):   # what the hand-offs of the last stages fetch: the next statement's stages
            stages.append(rng.integers(0, 255, STAGE, dtype=np.uint8))
        self.stream = np.concatenate(stages)

    def reference(self, w):
        xh, xl = split(self.X[w])
        d = np.float64

        def act(v):
            if not self.silu:
                return np.maximum(v, 0).astype(np.float32)
            u = np.exp2((np.float32(-1.4426950408889634) * v).astype(np.float32).astype(d)).astype(np.float32)   # the statement's op sequence
            return (v * (1.0 / (u + np.float32(1.0)).astype(d)).astype(np.float32)).astype(np.float32)
        if self.H1:
            pre = (self.W1h.astype(d) @ xh.astype(d)).astype(np.float32) + self.b1[:, None]      # (bias = the chains' start value)
            hb = f16(act(pre))
            return self.W2h.astype(d) @ hb.astype(d)
        acc = self.W1h.astype(d) @ xh.astype(d) + self.W1h.astype(d) @ xl.astype(d) + self.W1l.astype(d) @ xh.astype(d)
        v = act((acc.astype(np.float32) * self.scale + self.b1[:, None]).astype(np.float32))
        hh, hl = split(v)
        return self.W2h.astype(d) @ hh.astype(d) + self.W2h.astype(d) @ hl.astype(d) + self.W2l.astype(d) @ hh.astype(d)

    def run(self, late_vmem, late_lds, order):
        gen, NT = self.gen, self.NT
        lds = np.zeros(256 * 1024, np.uint8)   # (roomier than the chip's: the layout here is the test's own)
        lds[RING_LDS:RING_LDS + gen.AHEAD * STAGE] = self.stream[:gen.AHEAD * STAGE]    # the stages the kernel's prologue requested
        gmem = self.stream
        waves = []
        for w in range(4):
            wv = E.Wave(lds=lds, gmem=gmem, gbase=GBASE, wave_id=w)
            wv.late_vmem, wv.late_lds = late_vmem, late_lds
            priv = PRIV_LDS + w * PRIV_STRIDE
            xh, xl = split(self.X[w])
            for ks in range(self.ks_in):
                for jt in range(NT):
                    for part, src in enumerate((xh, xl)):
                        i = 2 * (NT * ks + jt) + part
                        lds[priv + 1024 * i:priv + 1024 * (i + 1)] = image_bytes(src.astype(np.float32), 32 * ks, 16 * jt)
            lane = np.arange(64, dtype=np.uint64)
            gn = np.uint64(GBASE + gen.AHEAD * STAGE) + 16 * lane
            wv.v[252] = (gn & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            wv.v[253] = (gn >> np.uint64(32)).astype(np.uint32)
            wv.s[40], wv.s[41], wv.s[42], wv.s[43], wv.s[44] = 0, RING_LDS, w, priv, self.n
            waves.append(wv)
        gen.SHAPE = gen.SHAPES[self.shape]
        E.run_waves(waves, gen.generate(), [OPS] * 4, order=order)
        ys = []
        for w, wv in enumerate(waves):
            vm, lg = wv.in_flight()
            assert "reg" not in vm, "a register load is still in flight when the statement ends"
            assert not lg, "LDS operations not waited for when the statement ends"
            # stages consumed -> ring slot and stream pointer handed to the next statement
            assert int(wv.s[40]) == self.n_stages % gen.RING
            gn = int(wv.v[252][0]) | (int(wv.v[253][0]) << 32)
            assert gn == GBASE + (gen.AHEAD + self.n_stages) * STAGE
            priv = PRIV_LDS + w * PRIV_STRIDE
            y = np.zeros((16 * self.ot_out, 16 * NT), np.float32)
            for ot in range(self.ot_out):
                for jt in range(NT):
                    img = lds[priv + 1024 * (NT * ot + jt):][:1024].view(np.float32).reshape(64, 4)
                    for l in range(64):
                        y[16 * ot + 4 * (l // 16):16 * ot + 4 * (l // 16) + 4, 16 * jt + l % 16] = img[l]
            ys.append(y)
        return ys


VARIANTS = [(), ("--h1",), ("--h1", "--ring6"), ("--ring6",), ("--nt=4",), ("--nt=4", "--h1")]
MODES = [(False, False, None), (True, False, None), (True, False, [3, 2, 1, 0]), (False, True, None), (False, True, [3, 2, 1, 0])]


def check(gen, chunks, mode, seed=0, shape="ffn"):
    case = Case(gen, chunks, seed, shape)
    ys = case.run(*mode)
    for w, y in enumerate(ys):
        ref = case.reference(w)
        err = np.abs(y - ref).max() / np.abs(ref).max()
        # (single-MFMA form: a last-bit difference in a pre-activation can move an fp16 rounding of the hidden layer; a stale
        # weight tile is an error of order one either way)
        assert err < (3e-4 if gen.H1 else 1e-5), (w, err, mode)


@pytest.mark.parametrize("argv", VARIANTS, ids=lambda a: " ".join(a) or "split-fp16")
@pytest.mark.parametrize("mode", MODES, ids=["in-order", "dma-late", "dma-late-reversed", "reads-late", "reads-late-reversed"])
def test_ffn_statement_under_adversarial_completion(argv, mode):
    """Three chunks: the prologue, both halves of the loop body and the tail behind the first half."""
    check(load_gen("gen_h3_ffn_asm", argv), 3, mode)


@pytest.mark.parametrize("argv", [("--h1", "--ring6"), ("--h1",), ()], ids=lambda a: " ".join(a) or "split-fp16")
@pytest.mark.parametrize("chunks", [1, 2, 4])
def test_ffn_statement_chunk_counts(argv, chunks):
    """One chunk (no loop trip), two (the tail behind the first half), four (the loop's back edge) - with the DMA landing late."""
    check(load_gen("gen_h3_ffn_asm", argv), chunks, (True, False, [3, 2, 1, 0]), seed=chunks)


def test_the_checker_sees_a_missing_wait(monkeypatch):
    """The protocol test must be able to fail: loosen the DMA wait of the heavy stages by one stage and the result is wrong."""
    gen = load_gen("gen_h3_ffn_asm", ("--h1", "--ring6"))
    real = gen.sync

    def loose(light, before_light):
        return [l.replace("vmcnt(4)", "vmcnt(8)").replace("vmcnt(6)", "vmcnt(10)") for l in real(light, before_light)]
    monkeypatch.setattr(gen, "sync", loose)
    with pytest.raises(AssertionError):
        check(gen, 3, (True, False, None))


@pytest.mark.parametrize("argv", [(), ("--h1",), ("--h1", "--ring6"), ("--nt=4",)], ids=lambda a: " ".join(a) or "split-fp16")
@pytest.mark.parametrize("shape", ["in", "out"])
@pytest.mark.parametrize("mode", [MODES[2], MODES[3]], ids=["dma-late-reversed", "reads-late"])
def test_in_and_out_mlp_statements(argv, shape, mode):
    """The SiLU shapes (64 -> 256 -> 128 and 128 -> 256 -> 16 in the product; here three chunks)."""
    check(load_gen("gen_h3_ffn_asm", argv), 3, mode, seed=5, shape=shape)


# ---------------------------------------------------------------------------------------------------------------------------
# The kernel-attention statements (tools/gen_h3_attn_asm.py): score fragments arrive by ordinary register loads in the same
# queue as the LDS-DMA pieces (r04's hole was one of those loads landing after the statement's exit), the X^T operands come
# from the wave-private block.  Differential form: the in-order run of the emulator is the reference (its arithmetic is pinned
# on the GPU and by the glue tests), the adversarial runs must reproduce it bit for bit on arbitrary finite data.
# ---------------------------------------------------------------------------------------------------------------------------
ATT_OPS = dict(OPS, heads="s45", sf="v[250:251]", xt="s46", win="s47")
XT_LDS = 64 * 1024                       # wide layout: the workgroup's shared transposed tile (106 KiB), private blocks behind it


def run_attention(gen, heads, late_vmem, late_lds, order, seed=3):
    rng = np.random.default_rng(seed)
    NT = gen.NT
    wide = hasattr(gen, "NG")            # tools/gen_h3_attn_wide_asm.py
    ahead, ring = (5, 5) if wide else (gen.AHEAD, gen.RING)
    sf_head = gen.NG * NT * gen.FRAG_BLOCK if wide else NT * gen.SF_BYTES
    priv0, stride = (XT_LDS + 112 * 1024, 24 * 1024) if wide else (PRIV_LDS, PRIV_STRIDE)
    n_stages = heads * 4 * (1 if gen.H1 else 2)
    small = lambda n: (rng.integers(-512, 512, n).astype(np.float32) / 1024).astype(np.float16).view(np.uint8)   # finite halves
    stream = small((n_stages + ahead + 2) * STAGE // 2)
    sf = small((heads + 2) * sf_head * 4 // 2)       # per wave: heads + the head of slack the last loads touch
    gmem = np.concatenate([stream, sf])
    lds = np.zeros(320 * 1024, np.uint8)
    lds[RING_LDS:RING_LDS + ahead * STAGE] = stream[:ahead * STAGE]
    if wide:
        lds[XT_LDS:XT_LDS + 2 * gen.XT_LO] = small(gen.XT_LO)
    waves = []
    for w in range(4):
        wv = E.Wave(lds=lds, gmem=gmem, gbase=GBASE, wave_id=w)
        wv.late_vmem, wv.late_lds = late_vmem, late_lds
        priv = priv0 + w * stride
        if not wide:
            lds[priv:priv + 32 * 1024] = small(16 * 1024)       # the transposed copy of x: hi / lo images per feature tile
        lane = np.arange(64, dtype=np.uint64)
        gn = np.uint64(GBASE + ahead * STAGE) + 16 * lane
        wv.v[252], wv.v[253] = (gn & np.uint64(0xFFFFFFFF)).astype(np.uint32), (gn >> np.uint64(32)).astype(np.uint32)
        sfp = np.uint64(GBASE + len(stream) + w * (heads + 2) * sf_head)
        wv.v[250], wv.v[251] = np.uint32(sfp & np.uint64(0xFFFFFFFF)), np.uint32(sfp >> np.uint64(32))
        wv.s[40], wv.s[41], wv.s[42], wv.s[43], wv.s[45] = 0, RING_LDS, w, priv, heads
        # the wave's key window: byte offset of its first K = 32 group in a row of the tile (it never leaves the 192 tokens)
        wv.s[46], wv.s[47] = XT_LDS, (64 * min(w, 6 - gen.NG) if wide else 0)
        waves.append(wv)
    E.run_waves(waves, gen.generate(), [ATT_OPS] * 4, order=order)
    out = []
    for w, wv in enumerate(waves):
        vm, lg = wv.in_flight()
        assert "reg" not in vm, "a score-fragment load is still in flight when the statement ends"
        assert not lg
        assert int(wv.s[40]) == n_stages % ring
        priv = priv0 + w * stride
        out.append(lds[priv:priv + 8 * NT * 1024].copy())
    return out


ATTENTION = [("gen_h3_attn_asm", a) for a in ((), ("--mode=windowed",), ("--ring6",), ("--nt=4",), ("--nt=4", "--h1"))] + \
            [("gen_h3_attn_wide_asm", a) for a in ((), ("--h1",), ("--ng=3",), ("--ng=3", "--h1"), ("--ng=6",), ("--ng=6", "--h1"))]


@pytest.mark.parametrize("name,argv", ATTENTION, ids=lambda a: a if isinstance(a, str) else (" ".join(a) or "full"))
def test_attention_statement_under_adversarial_completion(name, argv):
    gen = load_gen(name, argv)
    ref = run_attention(gen, 2, False, False, None)
    assert all(np.isfinite(r.view(np.float32)).all() and np.abs(r.view(np.float32)).max() > 0 for r in ref)
    for mode in MODES[1:]:
        got = run_attention(gen, 2, *mode)
        for w in range(4):
            assert np.array_equal(got[w], ref[w]), (argv, mode, w)


def test_the_checker_sees_r04s_race(monkeypatch):
    """r04's hole, re-opened: without the wait in front of the statement's exit the fragment loads of the head behind the last
    are still in flight when the registers go back to the caller."""
    gen = load_gen("gen_h3_attn_wide_asm", ("--h1",))
    real = gen.generate

    def without_exit_wait():
        lines = real()
        i = max(k for k, l in enumerate(lines) if l.startswith("s_waitcnt vmcnt(") and "lgkmcnt" not in l)
        return lines[:i] + lines[i + 1:]
    monkeypatch.setattr(gen, "generate", without_exit_wait)
    with pytest.raises(AssertionError, match="still in flight"):
        run_attention(gen, 2, True, False, None)


def run_dense_attention(gen, late_vmem, late_lds, order, seed=4):
    """The dense softmax block (tools/gen_h3_dense_attn_asm.py; 8 heads: 24 in_proj stages (q, k, v per head) + 8 out_proj stages)."""
    rng = np.random.default_rng(seed)
    NT, ring, n_stages, side = gen.NT, gen.RING, 32, 300 * 1024
    priv_stride = 64 * 1024 if NT == 4 else PRIV_STRIDE
    small = lambda n: (rng.integers(-512, 512, n).astype(np.float32) / 1024).astype(np.float16).view(np.uint8)
    gmem = small((n_stages + ring + 2) * STAGE // 2)
    lds = np.zeros(332 * 1024, np.uint8)
    lds[RING_LDS:RING_LDS + ring * STAGE] = gmem[:ring * STAGE]
    sl = (rng.standard_normal(1280) * 0.1).astype(np.float32)
    sl[640] = sl[642] = 2.0 ** -2                       # in_proj / out_proj scales
    lds[side:side + 5120] = sl.view(np.uint8)
    ops = dict(OPS, sl="s46", **{f"m{jt}{p}": f"v{244 + 2 * jt + k}" for jt in range(3) for k, p in enumerate("lh")})
    waves = []
    for w in range(4):
        wv = E.Wave(lds=lds, gmem=gmem, gbase=GBASE, wave_id=w)
        wv.late_vmem, wv.late_lds = late_vmem, late_lds
        priv = PRIV_LDS + w * priv_stride
        lds[priv:priv + 8 * NT * 1024] = small(4 * NT * 1024)       # the split activations: 8 NT operand images
        lane = np.arange(64, dtype=np.uint64)
        gn = np.uint64(GBASE + ring * STAGE) + 16 * lane
        wv.v[252], wv.v[253] = (gn & np.uint64(0xFFFFFFFF)).astype(np.uint32), (gn >> np.uint64(32)).astype(np.uint32)
        for r in range(244, 250):
            wv.v[r] = np.uint32(0xFFFFFFFF if w % 2 == 0 else 0x0FFF0FFF)      # key masks (odd waves: the last four keys of a tile padded)
        wv.s[40], wv.s[41], wv.s[42], wv.s[43], wv.s[46] = 0, RING_LDS, w, priv, side
        waves.append(wv)
    E.run_waves(waves, gen.generate(), [ops] * 4, order=order)
    out = []
    for w, wv in enumerate(waves):
        vm, lg = wv.in_flight()
        assert "reg" not in vm and not lg
        assert int(wv.s[40]) == n_stages % ring
        priv = PRIV_LDS + w * priv_stride
        out.append(lds[priv:priv + 8 * NT * 1024].copy())
    return out


@pytest.mark.parametrize("argv", [(), ("--nt=4",)], ids=["48-token", "64-token"])
def test_dense_attention_statement_under_adversarial_completion(argv):
    gen = load_gen("gen_h3_dense_attn_asm", argv)
    ref = run_dense_attention(gen, False, False, None)
    assert all(np.isfinite(r.view(np.float32)).all() and np.abs(r.view(np.float32)).max() > 0 for r in ref)
    for mode in MODES[1:]:
        got = run_dense_attention(gen, *mode)
        for w in range(4):
            assert np.array_equal(got[w], ref[w]), (mode, w)


# ---------------------------------------------------------------------------------------------------------------------------
# The WHOLE encoder-stack statements (tools/gen_h3_enc_asm.py): attention block, LayerNorm glue, FFN, the layer loop, the
# per-layer side-block DMA (which nobody waits for explicitly: "the first hand-off inside the attention block covers it"),
# the barriers that stand between one wave's X^T writes and another's reads (wide / paired layouts), the dense model's
# double-buffered side block.  Differential again: two layers of every product statement in the kernel's own LDS layout,
# in-order run = reference, the adversarial runs must reproduce its operand images for the out-MLP bit for bit.
# ---------------------------------------------------------------------------------------------------------------------------
ENC_VARIANTS = [(), ("--mode=windowed",), ("--ring6",), ("--mode=windowed", "--ring6"), ("--h1",), ("--h1", "--ring6"), ("--mode=windowed", "--h1", "--ring6"),
                ("--nt=4",), ("--nt=4", "--h1"), ("--nt=4", "--pair"), ("--nt=4", "--pair", "--h1"),
                ("--wide",), ("--wide", "--h1"), ("--wide", "--ng=3"), ("--wide", "--ng=6"), ("--wide", "--ng=6", "--h1"),
                ("--dense",), ("--dense", "--h1"), ("--dense", "--nt=4")]


def run_enc(enc, late_vmem, late_lds, order, seed=7, layers=2, heads=2, chunks=2):
    rng = np.random.default_rng(seed)
    NT, ffn, attn = enc.NT, enc.ffn, enc.attn
    ring = ffn.RING
    wave_lds = 32768 if enc.NT4 else 28672 if enc.WIDE else 24576             # csrc H3N4_WAVE_LDS / H3W_WAVE_LDS / H3(D)_WAVE_LDS
    priv0 = ring * STAGE
    sl = priv0 + 4 * wave_lds
    side_stride = 1024 * enc.SIDE_CHUNKS
    assert sl + (2 if enc.DENSE and not enc.NT4 else 1) * side_stride <= 160 * 1024   # (dense on 64-token waves: single-buffered)
    att_stages = 32 if enc.DENSE else heads * 4 * (1 if enc.H1 else 2)
    n_stages = layers * (att_stages + chunks * (2 if enc.H1 else 4))
    small = lambda n: (rng.integers(-512, 512, n).astype(np.float32) / 1024).astype(np.float16).view(np.uint8)
    stream = small((n_stages + ffn.AHEAD + 2) * STAGE // 2)
    sf_head = 0 if enc.DENSE else (attn.NG * NT * attn.FRAG_BLOCK if enc.WIDE else NT * attn.SF_BYTES)
    sf_layer = (heads + 2) * sf_head
    sf = small(max(4 * (layers + 1) * sf_layer // 2, 8))
    side = (rng.standard_normal((layers + 1) * side_stride // 4) * 0.3).astype(np.float32)
    for l in range(layers + 1):
        side[l * side_stride // 4 + 640:l * side_stride // 4 + 643] = 0.25      # (dense: the attention block's scales)
    scales = np.full(64, 0.25, np.float32)
    dump = np.zeros(4096, np.uint8)
    parts = [stream, sf, side.view(np.uint8), scales.view(np.uint8), dump]
    offs = np.cumsum([0] + [len(p) for p in parts])
    gmem = np.concatenate(parts)
    lds = np.zeros(160 * 1024, np.uint8)
    lds[:ffn.AHEAD * STAGE] = stream[:ffn.AHEAD * STAGE]
    if enc.DENSE:   # layer 0's side block is in flight when the statement starts (the kernel's prologue): here it has landed
        lds[sl:sl + side_stride] = side[:side_stride // 4].view(np.uint8)
    ops = {"cur": "s40", "gn": "v[252:253]", "ring": "s41", "wave": "s42", "priv": "s43", "chunks": "s44", "heads": "s45",
           "layers": "s46", "sl": "s47", "eps": "s54", "padt": "s55", "padm": "v247", "stampen": "s39", "scales": "s[52:53]",
           "xt": "s30", "win": "s31"}
    if enc.STATELESS:
        ops.update(sf="s[36:37]", sfstride="s48", sfstridehi="s49", side="s[34:35]", sidestride="s50", dump="s[32:33]")
        ops.update({f"m{jt}{p}": f"v{248 + 2 * jt + k}" if jt < 2 else f"v{254 + k}" for jt in range(3) for k, p in enumerate("lh")})
    else:
        ops.update(sf="v[250:251]", sfstride="s[48:49]", side="v[248:249]", sidestride="s[50:51]", dump="v[254:255]")
    waves = []
    lane = np.arange(64, dtype=np.uint64)
    lo, hi = (lambda a: (a & np.uint64(0xFFFFFFFF)).astype(np.uint32)), (lambda a: (a >> np.uint64(32)).astype(np.uint32))
    for w in range(4):
        wv = E.Wave(lds=lds, gmem=gmem, gbase=GBASE, wave_id=w)
        wv.late_vmem, wv.late_lds = late_vmem, late_lds
        priv = priv0 + w * wave_lds
        lds[priv:priv + 8 * NT * 1024] = rng.standard_normal(8 * NT * 256).astype(np.float32).view(np.uint8)   # x: fp32 images
        gn = np.uint64(GBASE + ffn.AHEAD * STAGE) + 16 * lane
        wv.v[252], wv.v[253] = lo(gn), hi(gn)
        sfp = np.uint64(GBASE + offs[1] + w * (layers + 1) * sf_layer)
        sidep = np.uint64(GBASE + offs[2])
        scp, dmp = GBASE + int(offs[3]), GBASE + int(offs[4])
        wv.s[52], wv.s[53] = scp & 0xFFFFFFFF, scp >> 32
        if enc.STATELESS:
            wv.s[36], wv.s[37] = int(sfp) & 0xFFFFFFFF, int(sfp) >> 32
            wv.s[48], wv.s[49], wv.s[50] = sf_layer, 0, side_stride
            wv.s[34], wv.s[35] = int(sidep) & 0xFFFFFFFF, int(sidep) >> 32
            wv.s[32], wv.s[33] = dmp & 0xFFFFFFFF, dmp >> 32
            for r in (248, 249, 250, 251, 254, 255):
                wv.v[r] = np.uint32(0xFFFFFFFF if w % 2 == 0 else 0x0FFF0FFF)
        else:
            wv.v[250], wv.v[251] = np.uint32(int(sfp) & 0xFFFFFFFF), np.uint32(int(sfp) >> 32)
            wv.s[48], wv.s[49], wv.s[50], wv.s[51] = sf_layer, 0, side_stride, 0
            sl_lane = sidep + 16 * lane
            wv.v[248], wv.v[249] = lo(sl_lane), hi(sl_lane)
            wv.v[254], wv.v[255] = np.uint32(dmp & 0xFFFFFFFF), np.uint32(dmp >> 32)
        # padding: the last five tokens of the wave's last tile (one bit per token tile in the lane's mask word)
        pad = np.zeros(64, np.uint32)
        pad[(np.arange(64) % 16) >= 11] = 1 << (NT - 1)
        wv.v[247] = pad
        wv.s[55] = 1 << (NT - 1)
        wv.s[54] = np.float32(1e-5).view(np.uint32)
        wv.s[39], wv.s[40], wv.s[41], wv.s[42], wv.s[43] = 0, 0, 0, w, priv
        wv.s[44], wv.s[45], wv.s[46], wv.s[47] = chunks, heads, layers, sl
        if enc.WIDE and not enc.NT4:
            wv.s[30], wv.s[31] = priv0, 32 * min(3 * w, 12 - 2 * attn.NG)
        waves.append(wv)
    E.run_waves(waves, enc.generate(), [ops] * 4, order=order)
    out = []
    for w, wv in enumerate(waves):
        vm, lg = wv.in_flight()
        assert "reg" not in vm, "a register load is still in flight when the statement ends"
        assert not lg
        assert int(wv.s[40]) == n_stages % ring
        priv = priv0 + w * wave_lds
        out.append(lds[priv:priv + wave_lds].copy())    # the out-MLP's operand images (and whatever else the block holds)
    return out


@pytest.mark.parametrize("argv", ENC_VARIANTS, ids=lambda a: " ".join(a) or "48-token")
def test_encoder_stack_statement_under_adversarial_completion(argv):
    enc = load_gen("gen_h3_enc_asm", argv)
    ref = run_enc(enc, False, False, None)
    if not enc.H1:   # 4 k-steps x NT token tiles x (hi, lo) images of LayerNorm output: finite, not all zero
        imgs = [r[:8 * enc.NT * 1024].view(np.float16).astype(np.float32) for r in ref]
        assert all(np.isfinite(f).all() and 0.5 < np.abs(f).max() < 50 for f in imgs)
    for mode in (MODES[1], MODES[2], MODES[3]):   # DMA late (both wave orders), reads late
        got = run_enc(enc, *mode)
        for w in range(4):
            assert np.array_equal(got[w], ref[w]), (argv, mode, w)


def test_the_checker_sees_a_missing_barrier_between_partner_waves(monkeypatch):
    """Paired layout: a wave mixes against its partner's X^T images, which the partner writes in the glue of the layer before;
    the barrier at the top of a layer is what stands between the two.  Without it the runs depend on the wave order."""
    enc = load_gen("gen_h3_enc_asm", ("--nt=4", "--pair"))
    real = enc.layer_top
    monkeypatch.setattr(enc, "layer_top", lambda: [l for l in real() if l != "s_barrier"])
    a = run_enc(enc, False, False, None)
    b = run_enc(enc, False, False, [3, 2, 1, 0])
    assert any(not np.array_equal(a[w], b[w]) for w in range(4))


# ---------------------------------------------------------------------------------------------------------------------------
# r06: the dense softmax block NUMERICALLY - the differential test above says that the statement does not depend on when its
# memory operations land, not that it computes attention.  Here one layer's block runs on the emulator over a random packed
# stage stream and is held to a float64 restatement of nn.MultiheadAttention's arithmetic on the same operands (in_proj with
# scale / bias, 1 / sqrt(16) on q, key-masked softmax, P.V, out_proj k-steps per head pair; y unscaled, as the statement returns it).
# The 48-token statement (on hardware since r03) pins the restatement; the 64-token one (--nt=4: two query halves) is new.
# ---------------------------------------------------------------------------------------------------------------------------
def dense_attention_numeric(argv, seed=9):
    gen = load_gen("gen_h3_dense_attn_asm", argv)
    NT, ring = gen.NT, gen.RING
    H, n_stages, side = 8, 32, 300 * 1024    # (behind the four wave-private blocks: 64 KiB + 4 x 40 / 64 KiB)
    rng = np.random.default_rng(seed)
    # real weights, split into hi / lo tiles as h3_pack_weights does (a stream of unrelated random "lo" tiles would make the
    # dropped lo x lo term as large as 2^-12 of the result)
    W_in = (rng.standard_normal((384, 128)) * 0.2).astype(np.float32)     # rows: q (0..127), k, v
    W_out = (rng.standard_normal((128, 128)) * 0.2).astype(np.float32)
    wi_h, wi_l = split(W_in)
    wo_h, wo_l = split(W_out)
    stages = [("q", 0), ("k", 0)]
    for h in range(H):
        stages.append(("v", h))
        if h + 1 < H:
            stages += [("q", h + 1), ("k", h + 1)]
        if h % 2:
            stages += [("o", h // 2, 0), ("o", h // 2, 1)]
    assert len(stages) == n_stages
    gmem = np.zeros((n_stages + ring + 2) * STAGE, np.uint8)
    for i, st in enumerate(stages):
        for p in range(4):
            if st[0] == "o":
                r0, c0, (hi, lo) = 16 * (4 * st[2] + p), 32 * st[1], (wo_h, wo_l)
            else:
                r0, c0, (hi, lo) = 128 * "qkv".index(st[0]) + 16 * st[1], 32 * p, (wi_h, wi_l)
            gmem[i * STAGE + 2048 * p:][:1024] = tile_bytes(hi, r0, c0)
            gmem[i * STAGE + 2048 * p + 1024:][:1024] = tile_bytes(lo, r0, c0)
    Wi = wi_h.astype(np.float64) + wi_l.astype(np.float64)
    Wo = wo_h.astype(np.float64) + wo_l.astype(np.float64)
    lds = np.zeros(320 * 1024, np.uint8)
    lds[RING_LDS:RING_LDS + ring * STAGE] = gmem[:ring * STAGE]
    sl = (rng.standard_normal(1280) * 0.1).astype(np.float32)
    sl[640] = 2.0 ** -1                                  # in_proj scale
    lds[side:side + 5120] = sl.view(np.uint8)
    ops = dict(OPS, sl="s46", **{f"m{jt}{p}": f"v{244 + 2 * jt + k}" for jt in range(3) for k, p in enumerate("lh")})
    T = 16 * NT
    priv_stride = 2 * 8 * NT * 1024 if NT == 4 else PRIV_STRIDE
    waves, xs, masks = [], [], []
    for w in range(4):
        wv = E.Wave(lds=lds, gmem=gmem, gbase=GBASE, wave_id=w)
        priv = PRIV_LDS + w * priv_stride
        x = (rng.standard_normal((128, T)) * 0.7).astype(np.float32)       # [feature][token]
        xh, xl = split(x)
        for ks in range(4):
            for jt in range(NT):
                lds[priv + 1024 * ((ks * NT + jt) * 2):][:1024] = image_bytes(xh, 32 * ks, 16 * jt)
                lds[priv + 1024 * ((ks * NT + jt) * 2 + 1):][:1024] = image_bytes(xl, 32 * ks, 16 * jt)
        xs.append(xh.astype(np.float64) + xl.astype(np.float64))
        valid = np.ones(T, bool)
        if w % 2:
            valid[T - 5:] = False                        # odd waves: the last five keys are padding
        masks.append(valid)
        lane = np.arange(64, dtype=np.uint64)
        gn = np.uint64(GBASE + ring * STAGE) + 16 * lane
        wv.v[252], wv.v[253] = (gn & np.uint64(0xFFFFFFFF)).astype(np.uint32), (gn >> np.uint64(32)).astype(np.uint32)
        bits = sum(1 << t for t in range(T) if valid[t])
        g = (np.arange(64) // 16).astype(np.uint64)
        word = np.uint64(bits) >> (np.uint64(4) * g)     # csrc: kvalid[jt] = m >> (4 g); one molecule over the wave's tokens here
        for jt in range(3):
            wv.v[244 + 2 * jt], wv.v[245 + 2 * jt] = (word & np.uint64(0xFFFFFFFF)).astype(np.uint32), (word >> np.uint64(32)).astype(np.uint32)
        wv.s[40], wv.s[41], wv.s[42], wv.s[43], wv.s[46] = 0, RING_LDS, w, priv, side
        waves.append(wv)
    E.run_waves(waves, gen.generate(), [ops] * 4, order=None)
    sc_in = float(sl[640])
    worst = 0.0
    for w in range(4):
        x, valid = xs[w], masks[w]
        Os = []
        for h in range(H):
            bq, bk, bv = (sl[656 + 16 * h:][:16].astype(np.float64), sl[656 + 128 + 16 * h:][:16].astype(np.float64),
                          sl[656 + 256 + 16 * h:][:16].astype(np.float64))
            q = (Wi[16 * h:][:16] @ x * sc_in + bq[:, None]) * 0.25
            k = Wi[128 + 16 * h:][:16] @ x * sc_in + bk[:, None]
            v = Wi[256 + 16 * h:][:16] @ x * sc_in + bv[:, None]
            S = k.T @ q                                   # [key][query]
            S[~valid, :] = -3.0e4
            P = np.exp(S - S.max(0, keepdims=True))
            P /= P.sum(0, keepdims=True)
            Os.append(v @ P)                              # [feature 16][query]
        y = Wo @ np.concatenate(Os, axis=0)               # out_proj without its scale / bias (the caller's)
        priv = PRIV_LDS + w * priv_stride
        got = np.zeros((128, T))
        for ot in range(8):
            for jt in range(NT):
                img = lds[priv + 1024 * (ot * NT + jt):][:1024].view(np.float32).reshape(64, 4)
                for l in range(64):
                    got[16 * ot + 4 * (l // 16):16 * ot + 4 * (l // 16) + 4, 16 * jt + l % 16] = img[l]
        assert np.isfinite(got).all()
        worst = max(worst, np.abs(got - y).max() / np.abs(y).max())
    return worst


@pytest.mark.parametrize("argv", [(), ("--nt=4",)], ids=["48-token", "64-token"])
def test_dense_attention_statement_computes_attention(argv):
    err = dense_attention_numeric(argv)
    assert err < 2e-5, err     # three-term fp16 products, fp32 softmax (v_exp_f32 / v_rcp_f32) against float64
