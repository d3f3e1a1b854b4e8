#!/usr/bin/env python3
"""Generate timewarp_amd/csrc/tw_h3_enc_asm.inc (and tw_h3_encw_asm.inc, --mode=windowed): the WHOLE encoder stack of
the split-fp16 net-block kernel (gfx950) as the body of one `asm volatile` statement - per layer the kernel-attention
block (gen_h3_attn_asm.py) and the FFN (gen_h3_ffn_asm.py), embedded unchanged, and between them what r01-r03 left to
hipcc: residual add, LayerNorm, padding select, fp32 -> fp16 hi/lo split, the transposed copy of x.

Why: the section profile put 15.6 k of a layer's 240 k cycles into that compiled glue (1.4 k instructions at ~6 cycles,
96 AGPR <-> VGPR moves of the residual per section, two trips of y and of the split operands through the LDS).  Here
  * the residual never exists beside the accumulators: they START at x / scale (attention) or (x + b2) / scale (FFN) -
    scales are powers of two, so y * scale = x + output with nothing else to add - and the fp32 residual registers of the
    compiled version (96 per lane, kept in AGPRs across the GEMM blocks) are gone;
  * y is read once from a0..a95, normalised in registers, split straight into the FFN's operand registers / through the
    matrix pipe into the transposed tile of the next attention block, and the next accumulators are seeded in the same pass;
  * every phase runs over 24 independent (feature tile, token tile) register tiles, so VALU latencies overlap.

One LayerNorm (token = lane & 15, its 128 features spread over 8 tiles x 4 lane groups x 4 registers):
  P1  t = acc * scale                                   sums per token tile
  P2  mean  (v_permlane16/32_swap across the four lane groups)
  P3  d = t - mean, squares summed
  P4  rstd = rsq(var + eps)                              (v_rsq_f32, 1 ulp; the compiled version divided by an IEEE sqrt)
  P5  x' = d * rstd * w + b, padding tokens -> 0, then  G1: split -> FFN operands, acc <- (x' + b2) / s_w2
                                                        G2: split -> K=16 MFMAs against the identity -> X^T images,
                                                            acc <- x' / s_wc(next layer);  last layer: split -> LDS for
                                                            the out-MLP statement
Register map of the glue (the embedded blocks own v0..v211 / a0..a119 while they run):
  v72..v167   T[ft][jt]: y -> t -> d -> x'  (same index as the accumulators: T = 72 + a)
  v0..v71     G1: the FFN's xb operands (k-steps 0-2; k-step 3 goes to a96..a119).  G2: transposer - D tiles v0..v47
              (two buffers x 3 token tiles x hi/lo), split halves v48..v71 (two buffers)
  a96..a191   the transposed copy of x (fp16 hi / lo, 12 per feature tile) from the transposer to the end of the
              attention block: the mixing MFMAs take their A operands there, nothing goes through the LDS
  v168..v191  LayerNorm weight / bias / FFN output bias of a feature tile (two buffers; streamed from the side block)
  v192..v211  temporaries, X^T image staging (G2: v176..v179, v188..v191, v208..v211)
  v212..v223  sums, -mean / rstd pairs
  v224..v237  persistent: LDS addresses, padding mask, score-fragment base, side-block DMA source, identity operand,
              broadcast constants
  s74..s82, s98, s99  layer counter, scales and their inverses, eps
Operands: see the asm statement in csrc/tw_netblock_h3.hip (kernel, `ENC`).

--nt=4 (tw_h3n4_enc_asm.inc / tw_h1n4_enc_asm.inc): the 64-token build (one molecule of 49-64 atoms per wave; BASELINE configs[3]).
Same phases over 32 (feature tile, token tile) register tiles; what differs:
  * the embedded blocks own v0..v245 / a0..a191 (gen_h3_attn_asm.py / gen_h3_ffn_asm.py --nt=4), so the glue keeps NOTHING
    in registers across them: lane addresses, the identity operand and the broadcast constants are rebuilt at the top of each
    phase (a dozen VALU ops against ~1200), the layer's pointers live in SGPRs (score fragments, side block, stamp buffer);
  * the transposed copy of x goes to the wave-private LDS block as the per-section build's images (the attention block's
    AGPRs a128..a255 would hold it, but then the compiler's own live state would have nowhere to go but scratch - the point
    of this statement is ScratchSize 0);
  * register map:  v64..v191 T[ft][jt];  v0..v63 G1: the FFN's xb operands of k-steps 0, 1 (k-steps 2, 3 go to a128..a191),
    G2: transposer D tiles (two buffers x 4 token tiles x hi/lo), exit: operand images of the out-MLP (two k-steps);
    v192..v215 LayerNorm w / b / b2 of a feature tile (two buffers), G2: split halves H (buffer 0);  v216..v223, v224..v231
    temporaries (v224..v231 also -mean / rstd pairs inside a LayerNorm);  v232..v235 accumulator seed / the pair of ones;
    v236..v243 sums inside a LayerNorm, then broadcast constants and the identity operand;  v244 side-block lane address,
    v245 wave-private lane address.  G2 also uses T tiles that are dead by then: T[0] as split halves (buffer 1), T[1] as
    image staging;
  * SGPRs s56..s81 (the embedded blocks own s82..s99)."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


WINDOWED = "--mode=windowed" in sys.argv
# --h1: the single-MFMA "fast" variant (TW_PATH_FUSED_H1; gen_h3_ffn_asm.py / gen_h3_attn_asm.py H1).  The glue only ever
# produces the fp16 hi halves: a split is one pack per register pair, the transposer sends three tiles per feature tile
# through the matrix pipe instead of six.  The section stamps are always compiled in (one scalar compare per stamp when
# off): bench.py reads the attention block's cycles from them.
H1 = "--h1" in sys.argv
NT4 = "--nt=4" in sys.argv      # 64-token waves (the embedded generators read the same flag)
# --nt=4 --pair: one molecule of 97-128 atoms per pair of 64-token waves (gen_h3_attn_asm.py --pair: the attention block mixes
# against both waves' X^T images).  The glue is the 64-token statement's unchanged - every wave still writes its own images
# into its own block, and the barrier at the top of a layer (layer_top) stands between those writes and the partner's reads.
PAIR = "--pair" in sys.argv
assert not PAIR or NT4
# --wide [--ng=3|6]: the wide layout (molecules packed over the workgroup's 192 token slots, 25-192 atoms): the attention block
# is gen_h3_attn_wide_asm.py's, the transposed copy of x the workgroup's SHARED tile in the LDS
WIDE = "--wide" in sys.argv
# --dense: the dense softmax model (transformer_nvp; BASELINE configs[4]): the attention block is gen_h3_dense_attn_asm.py's,
# whose operands are the split activations themselves (B operand of q / k, A operand of v) in a96..a191 - no transposed copy.
# With --h1 the MLP sections are the single-MFMA ones, the attention block stays in split form (csrc: "MLP sections only").
# --dense --nt=4 (r06, tw_h3n4d_enc_asm.inc): the dense model on 64-token waves - one molecule of 49-64 atoms per wave; the
# softmax block in two query halves (gen_h3_dense_attn_asm.py --nt=4), y in a0..a127 and the split activations in a128..a255 (all
# 256 AGPRs), the side block SINGLE-buffered (160 KiB of LDS are full: three-slot ring + 4 x 32 KiB + 5 KiB) - fetched at the
# top of a layer behind a barrier and waited for there, ~2 us per layer.
DENSE = "--dense" in sys.argv
STATELESS = NT4 or WIDE or DENSE  # the glue keeps nothing in VGPRs across the embedded blocks; pointers in SGPRs
# --ring6 (48-token kernel-attention build only - the one layout with 9 KiB of LDS to spare): six stage buffers, the FFN's
# stages pair up under one barrier (gen_h3_ffn_asm.py R6; the embedded generators read the same flag)
R6 = "--ring6" in sys.argv
assert not (R6 and STATELESS)
EXPERIMENT = set(filter(None, os.environ.get("H3_ENC_EXPERIMENT", "").split(",")))
# The section stamps are compiled into every statement (r05; r04: the fast-mode ones only): five scalar compare-and-branch
# pairs per layer when off.  bench.py reads the attention block's share of a launch from the PRODUCT build that way
# (tw_debug_set_flags 16 | 8192) instead of from the per-section build.
EXPERIMENT.add("stamps")
attn = load("gen_h3_dense_attn_asm" if DENSE else ("gen_h3_attn_wide_asm" if WIDE else "gen_h3_attn_asm"))
ffn = load("gen_h3_ffn_asm")
ffn.H1 = H1
if not DENSE:
    attn.H1 = H1
H1A = H1 and not DENSE          # does the attention block take hi halves only?
attn.FUSED = True
if not (WIDE or DENSE):
    attn.WINDOWED = WINDOWED
if not WIDE:
    assert attn.NT4 == NT4
if not STATELESS:
    attn.XT_AGPR = 96   # the transposed copy of x lives in a96..a191 from the transposer to the end of the attention block
ffn.FUSED = True
ffn.SHAPE = ffn.SHAPES["ffn"]
assert ffn.NT4 == NT4 and not (NT4 and WINDOWED) and not (WIDE and (NT4 or WINDOWED)) and not (DENSE and (WINDOWED or WIDE))
assert not (DENSE and NT4 and H1), "dense model on 64-token waves: the split-fp16 statement only"
DENSE1 = DENSE and NT4          # single-buffered side block

NT = 4 if NT4 else 3
SIDE_CHUNKS = 5 if DENSE else 3
XT_IMG = None if (WIDE or DENSE) else attn.XT_IMG
SIDE_OUTB = 1040                # dense: out_proj bias [128] in the layer's side block (csrc H3D_OUTB)
SIDE_LDS_BYTES = 1024 * SIDE_CHUNKS
# side block of a layer (floats): LN1 w, LN1 b, FFN b2, LN2 w, LN2 b (128 each)
SIDE_LN1W, SIDE_LN1B, SIDE_B2, SIDE_LN2W, SIDE_LN2B = 0, 128, 256, 384, 512

T = lambda ft, jt: 72 + 4 * (3 * ft + jt)
ACC = lambda ft, jt: 4 * (3 * ft + jt)
PRM = lambda buf, which: 168 + 12 * buf + 4 * which        # which: 0 = w, 1 = b, 2 = b2 / s_w2
TMP = lambda k: 192 + 4 * (k % 2)
U = lambda k: 200 + 4 * (k % 2)
V_ACC0 = 208                                               # G1: seed of an accumulator tile
D = lambda buf, jt, part: 4 * (6 * buf + 2 * jt + part)    # part: 0 = hi, 1 = lo
H = lambda buf, jt, part: 48 + 12 * buf + 4 * jt + 2 * part
IMG01 = lambda part: (176, 188)[part]                      # G2 streams two parameters per buffer: 176..179, 188..191 are free
IMG2 = lambda part: 208 + 2 * part
S2 = lambda jt: 212 + 2 * jt
NM = lambda jt: 218 + 2 * jt
V_PRIV8, V_PRIV16, V_SLG, V_PAD, V_SFB, V_SIDE, V_IDB, V_C, V_C2 = 224, 225, 226, 227, 228, 230, 232, 234, 236
N_V = 244
S_LAYER, S_PADT, S_SCPTR, S_SCA, S_SCF, S_IA, S_IF, S_EPS, S_NA, S_NF = 76, 77, 78, 80, 81, 82, 98, 99, 74, 75
S_STAMP_T, S_STAMP_STEP = 70, 72
S_LO = 70                                                  # first SGPR of the clobber list
PADM = f"v{V_PAD}"
XBX = lambda ks, jt, part: (8 * (3 * ks + jt) if ks < 3 else 72 + 8 * jt) + (0 if part == 0 else 4)   # exit: all VGPRs
V_ONE = 240   # pair [1.0, 1.0]: a + b as fma(a, 1, b) - v_pk_add_f32 costs twice a v_pk_fma_f32 (tools/probe/valu_cost_probe.hip)
V_STAMP = 242   # stamps build only: dump address of the current layer's stamp block
if NT4:
    T = lambda ft, jt: 64 + 4 * (4 * ft + jt)
    ACC = lambda ft, jt: 4 * (4 * ft + jt)
    PRM = lambda buf, which: 192 + 12 * buf + 4 * which
    TMP = lambda k: 216 + 4 * (k % 2)
    U = lambda k: 224 + 4 * (k % 2)
    V_ACC0 = 232
    D = lambda buf, jt, part: 4 * (8 * buf + 2 * jt + part)
    H = lambda buf, jt, part: (192, T(0, 0))[buf] + 4 * jt + 2 * part     # buffer 1: T[0], dead once feature tile 0 is split
    IMG = lambda part, pair: T(1, 0) + 8 * part + 4 * pair               # T[1], dead once feature tile 1 is split
    S2 = lambda jt: 236 + 2 * jt
    NM = lambda jt: 224 + 2 * jt
    V_ONE = 232
    V_C, V_C2, V_IDB, V_SLG, V_PRIV16 = 236, 238, 240, 244, 245
    V_ZERO = 242                                           # stamps: zero offset of the saddr-form store
    V_PRIV8 = V_PAD = V_SFB = V_SIDE = V_STAMP = None      # (not kept in registers in this build)
    N_V = 246
    S_SFB, S_SIDE, S_DUMP = 56, 58, 60
    S_IF, S_EPS, S_IA, S_SCF, S_SCA, S_PADT, S_LAYER, S_NA = 62, 63, 64, 65, 66, 67, 68, 69
    S_STAMP_T, S_STAMP_STEP, S_NF, S_SCPTR = 70, 72, 74, 78
    S_LO = 56
    PADM = "%[padm]"
    XBX = lambda ks, jt, part: 32 * (ks % 2) + 8 * jt + 4 * part        # exit staging: two k-steps of images in v0..v63
    if DENSE:
        S_SLCUR, S_SLOTHER, S_OSC, S_SWAP = 56, 57, 76, 75     # (S_SFB's pair: this model has no score fragments)
        attn.SL = f"s{S_SLCUR}"
    else:
        attn.SF_BASE = f"s[{S_SFB}:{S_SFB + 1}]"
elif WIDE or DENSE:
    # the 48-token map; v224.. are scratch here, rebuilt per phase (the six-group attention statement owns v0..v239, the
    # dense one v0..v219)
    V_XTW, V_XTWL, V_ZERO = 228, 229, 242                  # this lane's 8 bytes of a shared-tile row, hi / lo half
    V_PRIV8 = V_PAD = V_SFB = V_SIDE = V_STAMP = None
    S_SFB, S_SIDE, S_DUMP = 56, 58, 60
    S_IF, S_EPS, S_IA, S_SCF, S_SCA, S_PADT, S_LAYER, S_NA = 62, 63, 64, 65, 66, 67, 68, 69
    S_STAMP_T, S_STAMP_STEP, S_NF, S_SCPTR = 70, 72, 74, 78
    S_LO = 56
    PADM = "%[padm]"
    if DENSE:
        # the side block is double-buffered in the LDS (layer l in buffer l % 2, fetched a whole layer ahead): the attention
        # block reads its biases and the in_proj scale at its very entry, which one DMA latency behind layer_top() would race
        S_SLCUR, S_SLOTHER, S_OSC, S_SWAP = 56, 57, 76, 75     # (S_SFB's pair: this model has no score fragments)
        attn.SL = f"s{S_SLCUR}"
    else:
        attn.SF_BASE = f"s[{S_SFB}:{S_SFB + 1}]"
else:
    attn.SF_BASE = f"v[{V_SFB}:{V_SFB + 1}]"


def vr(base, n=2):
    return f"v[{base}:{base + n - 1}]"


def sr(base, n=2):
    return f"s[{base}:{base + n - 1}]"


def _check_register_map():
    """The glue's register sets must not overlap where they are live together (a typo here is silent corruption)."""
    def span(lo, n):
        return set(range(lo, lo + n))
    if NT4:
        t = span(T(0, 0), 128)
        xb = span(0, 64)
        prm = span(PRM(0, 0), 24)
        tmp = span(TMP(0), 8)
        u = span(U(0), 8)
        seed = span(V_ACC0, 4)
        stats = span(S2(0), 8) | span(NM(0), 8) | span(V_ONE, 2)
        consts = span(V_C, 2) | span(V_C2, 2) | span(V_IDB, 2)
        addr = span(V_SLG, 1) | span(V_PRIV16, 1)
        # inside a LayerNorm: T, parameters, one temporary set, sums / means, the ones, the side-block address
        for a, b in ((t, prm), (t, tmp), (t, stats), (prm, tmp), (prm, stats), (tmp, span(S2(0), 8)), (addr, t | prm | tmp | stats)):
            assert not (a & b), sorted(a & b)
        assert not (span(NM(0), 8) & (span(S2(0), 8) | span(V_ONE, 2) | tmp))
        # G1 behind the LayerNorm: T, xb, b2 columns, both temporary sets, U, the seed, V_C, the side-block address
        g1_sets = (t, xb, prm, tmp, u, seed, span(V_C, 2), span(V_SLG, 1))
        for i, a in enumerate(g1_sets):
            for b in g1_sets[i + 1:]:
                assert not (a & b), sorted(a & b)
        # G2 behind the LayerNorm: T (tiles 0 / 1 reused once dead), D = xb, H buffer 0 in the parameter columns, temporaries, U
        d = set()
        for buf in range(2):
            for jt in range(NT):
                for part in range(2):
                    d |= span(D(buf, jt, part), 4)
        assert d == xb
        h0 = set().union(*(span(H(0, jt, part), 2) for jt in range(NT) for part in range(2)))
        h1 = set().union(*(span(H(1, jt, part), 2) for jt in range(NT) for part in range(2)))
        img = set().union(*(span(IMG(part, pair), 4) for part in range(2) for pair in range(2)))
        assert h0 <= prm and h1 == span(T(0, 0), 16) and img == span(T(1, 0), 16)
        g2_sets = (d, h0, tmp, u, span(V_C2, 2), span(V_IDB, 2), span(V_PRIV16, 1), t)
        for i, a in enumerate(g2_sets):
            for b in g2_sets[i + 1:]:
                assert not (a & b), sorted(a & b)
        xbx = set().union(*(span(XBX(ks, jt, part), 4) for ks in range(2) for jt in range(NT) for part in range(2)))
        assert xbx == xb
        assert max(t | prm | tmp | u | seed | stats | consts | addr) < N_V
        return
    t = span(T(0, 0), 96)
    xb = span(0, 72)
    transposer = span(D(0, 0, 0), 48) | span(H(0, 0, 0), 24)
    prm = span(PRM(0, 0), 24)
    tmp = span(TMP(0), 8) | span(U(0), 8) | span(V_ACC0, 4)
    img = span(IMG01(0), 4) | span(IMG01(1), 4) | span(IMG2(0), 2) | span(IMG2(1), 2)
    stats = span(S2(0), 6) | span(NM(0), 6)
    if WIDE or DENSE:
        persistent = span(V_PRIV16, 1) | span(V_SLG, 1) | span(V_XTW, 2) | span(V_IDB, 2) | span(V_C, 2) | span(V_C2, 2) | \
            span(V_ONE, 2) | span(V_ZERO, 1)
    else:
        persistent = span(V_PRIV8, 1) | span(V_PRIV16, 1) | span(V_SLG, 1) | span(V_PAD, 1) | span(V_SFB, 2) | span(V_SIDE, 2) | \
            span(V_IDB, 2) | span(V_C, 2) | span(V_C2, 2) | span(V_ONE, 2) | span(V_STAMP, 2)
    assert transposer == xb                                    # G2 reuses the FFN operand registers of G1
    for a, b in ((t, xb), (t, prm), (t, tmp), (t, stats), (t, persistent), (xb, prm), (xb, tmp), (xb, stats), (prm, tmp),
                 (prm, stats), (tmp, stats), (stats, persistent), (tmp, persistent), (prm, persistent), (xb, persistent)):
        assert not (a & b), sorted(a & b)
    # the image staging registers of G2 sit in the b2 columns of the parameter buffers (G2 streams w and b only) and
    # behind the temporaries G1's seeds use
    assert img <= span(PRM(0, 2), 4) | span(PRM(1, 2), 4) | span(V_ACC0, 4)
    assert max(persistent) < N_V and min(persistent) >= (220 if DENSE else 212)  # the embedded blocks own v0..v211 (dense: v219)
    xbx = set()
    for ks in range(4):
        for jt in range(NT):
            for part in range(2):
                xbx |= span(XBX(ks, jt, part), 4)
    assert len(xbx) == 96 and max(xbx) < T(2, 0)               # exit images: only T tiles that are dead by then


def interleave(*streams):
    """Round-robin merge of instruction streams (independent register tiles): hides VALU latencies."""
    out, streams = [], [list(s) for s in streams]
    while any(streams):
        for s in streams:
            if s:
                out.append(s.pop(0))
    return out


def quad_sums(regs):
    """regs: list of (value VGPR, scratch VGPR).  value <- sum over the four lanes sharing (lane & 15); same association
    as csrc h3_quad_sum."""
    out = []
    out += [f"v_mov_b32 v{b}, v{a}" for a, b in regs]
    out += ["s_nop 1"]
    out += [f"v_permlane16_swap_b32 v{a}, v{b}" for a, b in regs]
    out += [f"v_add_f32 v{a}, v{a}, v{b}" for a, b in regs]
    out += [f"v_mov_b32 v{b}, v{a}" for a, b in regs]
    out += ["s_nop 1"]
    out += [f"v_permlane32_swap_b32 v{a}, v{b}" for a, b in regs]
    out += [f"v_add_f32 v{a}, v{a}, v{b}" for a, b in regs]
    return out


def lane_addr(dst, base, shift, scratch, group=False):
    """64-token build: v{dst} <- %[base] + (lane << shift)   (group: the lane group lane >> 4 instead of the lane)."""
    L = [f"v_mbcnt_lo_u32_b32 v{scratch}, -1, 0", f"v_mbcnt_hi_u32_b32 v{scratch}, -1, v{scratch}"]
    if group:
        L.append(f"v_lshrrev_b32 v{scratch}, 4, v{scratch}")
    src = base if base[0] in "sv" and base[1:].isdigit() else f"%[{base}]"   # a register of the statement's own, or an operand
    L += [f"v_lshlrev_b32 v{dst}, {shift}, v{scratch}", f"v_add_u32 v{dst}, {src}, v{dst}"]
    return L


def identity_operand(t0, t1):
    """The B operand of the transposing K = 16 MFMA: element e of lane (i16, g) = (i16 == 4 g + e).  t0, t1: four scratch
    VGPRs each."""
    L = [f"v_mbcnt_lo_u32_b32 v{t0}, -1, 0", f"v_mbcnt_hi_u32_b32 v{t0}, -1, v{t0}",   # lane
         f"v_lshrrev_b32 v{t0 + 1}, 4, v{t0}",                                          # g
         f"v_and_b32 v{t0 + 2}, 15, v{t0}",                                             # i16
         f"v_lshlrev_b32 v{t0 + 3}, 2, v{t0 + 1}",                                      # 4 g
         f"v_sub_u32 v{t0 + 2}, v{t0 + 2}, v{t0 + 3}",                                  # i16 - 4 g  (0..3 -> a one in that element)
         f"v_mov_b32 v{V_IDB}, 0", f"v_mov_b32 v{V_IDB + 1}, 0",
         f"v_mov_b32 v{t1 + 2}, 0x3c00", f"v_mov_b32 v{t1 + 3}, 0x3c000000"]
    for e, (reg, src) in enumerate(((V_IDB, t1 + 2), (V_IDB, t1 + 3), (V_IDB + 1, t1 + 2), (V_IDB + 1, t1 + 3))):
        L += [f"v_cmp_eq_u32 vcc, {e}, v{t0 + 2}", f"v_cndmask_b32 v{reg}, v{reg}, v{src}, vcc"]
    return L


def layer_norm(inv_sgpr, w_off, b_off, label, pre_bias=None):
    """T <- LayerNorm(acc * scale) * w + b, padding tokens -> 0.  The scale (a power of two) never touches the data:
    with t = acc, x = scale * t:  (x - mean_x) * rsq(var_x + eps) = (t - mean_t) * rsq(var_t + eps / scale^2).
    pre_bias (dense, LayerNorm 1): t = acc + bias / scale first - out_proj's bias, from the side block at that float offset
    (the accumulators were seeded a layer earlier, before this layer's side block was in reach)."""
    L = []
    A = L.append
    if STATELESS:   # nothing survives the embedded blocks: the side-block lane address and the pair of ones are rebuilt here
        L += lane_addr(V_SLG, f"s{S_SLCUR}" if DENSE else "sl", 4, TMP(0), group=True)
        A(f"v_mov_b32 v{V_ONE}, 1.0")
        A(f"v_mov_b32 v{V_ONE + 1}, 1.0")
    for jt in range(NT):
        A(f"v_mov_b32 v{S2(jt)}, 0")
        A(f"v_mov_b32 v{S2(jt) + 1}, 0")
    # (64-token map: V_C2 shares registers with the sums S2 - the -mean pair of tile 0, written only in P2, carries the constant)
    V_PB = NM(0) if NT4 else V_C2
    if pre_bias is not None:
        A(f"v_mov_b32 v{V_PB}, s{inv_sgpr}")
        A(f"v_mov_b32 v{V_PB + 1}, s{inv_sgpr}")
        A(f"ds_read_b128 {vr(PRM(0, 0), 4)}, v{V_SLG} offset:{4 * pre_bias}")
    # P1: t = acc, sums
    for ft in range(8):
        for jt in range(NT):
            for r in range(4):
                A(f"v_accvgpr_read_b32 v{T(ft, jt) + r}, a{ACC(ft, jt) + r}")
        if pre_bias is not None:
            buf = ft % 2
            if ft + 1 < 8:
                A(f"ds_read_b128 {vr(PRM(1 - buf, 0), 4)}, v{V_SLG} offset:{4 * pre_bias + 64 * (ft + 1)}")
                A("s_waitcnt lgkmcnt(1)")
            else:
                A("s_waitcnt lgkmcnt(0)")
            for h in range(2):
                A(f"v_pk_mul_f32 {vr(PRM(buf, 0) + 2 * h)}, {vr(PRM(buf, 0) + 2 * h)}, {vr(V_PB)}")
            for h in range(2):
                for jt in range(NT):
                    A(f"v_pk_fma_f32 {vr(T(ft, jt) + 2 * h)}, {vr(PRM(buf, 0) + 2 * h)}, {vr(V_ONE)}, {vr(T(ft, jt) + 2 * h)}")
        for h in range(2):
            for jt in range(NT):
                A(f"v_pk_fma_f32 {vr(S2(jt))}, {vr(T(ft, jt) + 2 * h)}, {vr(V_ONE)}, {vr(S2(jt))}")
    # P2: -mean
    for jt in range(NT):
        A(f"v_add_f32 v{S2(jt)}, v{S2(jt)}, v{S2(jt) + 1}")
    L += quad_sums([(S2(jt), S2(jt) + 1) for jt in range(NT)])
    for jt in range(NT):
        A(f"v_mul_f32 v{NM(jt)}, 0xbc000000, v{S2(jt)}")
    for jt in range(NT):
        A(f"v_mov_b32 v{NM(jt) + 1}, v{NM(jt)}")
        A(f"v_mov_b32 v{S2(jt)}, 0")
        A(f"v_mov_b32 v{S2(jt) + 1}, 0")
    # P3: d = t - mean, squares
    for ft in range(8):
        for h in range(2):
            for jt in range(NT):
                A(f"v_pk_fma_f32 {vr(T(ft, jt) + 2 * h)}, {vr(T(ft, jt) + 2 * h)}, {vr(V_ONE)}, {vr(NM(jt))}")
        for h in range(2):
            for jt in range(NT):
                A(f"v_pk_fma_f32 {vr(S2(jt))}, {vr(T(ft, jt) + 2 * h)}, {vr(T(ft, jt) + 2 * h)}, {vr(S2(jt))}")
    # P4: rstd = rsq(var + eps / scale^2)
    for jt in range(NT):
        A(f"v_add_f32 v{S2(jt)}, v{S2(jt)}, v{S2(jt) + 1}")
    A(f"v_mov_b32 v{TMP(0)}, s{S_EPS}")
    A(f"v_mul_f32 v{TMP(0)}, s{inv_sgpr}, v{TMP(0)}")
    A(f"v_mul_f32 v{TMP(0)}, s{inv_sgpr}, v{TMP(0)}")
    L += quad_sums([(S2(jt), S2(jt) + 1) for jt in range(NT)])
    for jt in range(NT):
        A(f"v_fmamk_f32 v{S2(jt)}, v{S2(jt)}, 0x3c000000, v{TMP(0)}")
    for jt in range(NT):
        A(f"v_rsq_f32 v{NM(jt)}, v{S2(jt)}")
    A("s_nop 1")
    for jt in range(NT):
        A(f"v_mov_b32 v{NM(jt) + 1}, v{NM(jt)}")
    # P5a: x' = d * rstd * w + b  (w, b of a feature tile streamed from the side block, two buffers)
    L += [f"ds_read_b128 {vr(PRM(0, 0), 4)}, v{V_SLG} offset:{4 * w_off}", f"ds_read_b128 {vr(PRM(0, 1), 4)}, v{V_SLG} offset:{4 * b_off}"]
    for ft in range(8):
        buf = ft % 2
        if ft + 1 < 8:
            L += [f"ds_read_b128 {vr(PRM(1 - buf, 0), 4)}, v{V_SLG} offset:{4 * w_off + 64 * (ft + 1)}",
                  f"ds_read_b128 {vr(PRM(1 - buf, 1), 4)}, v{V_SLG} offset:{4 * b_off + 64 * (ft + 1)}"]
            A("s_waitcnt lgkmcnt(2)")
        else:
            A("s_waitcnt lgkmcnt(0)")
        for h in range(2):
            for jt in range(NT):
                A(f"v_pk_mul_f32 {vr(T(ft, jt) + 2 * h)}, {vr(T(ft, jt) + 2 * h)}, {vr(NM(jt))}")
        for h in range(2):
            for jt in range(NT):
                A(f"v_pk_fma_f32 {vr(T(ft, jt) + 2 * h)}, {vr(T(ft, jt) + 2 * h)}, {vr(PRM(buf, 0) + 2 * h)}, {vr(PRM(buf, 1) + 2 * h)}")
    # padding tokens -> 0: per token tile, only tiles that hold any (S_PADT) - three branches per LayerNorm
    for jt in range(NT):
        A(f"s_bitcmp0_b32 s{S_PADT}, {jt}")
        A(f"s_cbranch_scc1 .Lenc_nopad_{label}_{jt}_%=")
        A(f"v_and_b32 v{TMP(0)}, {1 << jt}, {PADM}")
        A(f"v_cmp_eq_u32 vcc, 0, v{TMP(0)}")
        for ft in range(8):
            for r in range(4):
                A(f"v_cndmask_b32 v{T(ft, jt) + r}, 0, v{T(ft, jt) + r}, vcc")
        A(f".Lenc_nopad_{label}_{jt}_%=:")
    return L


def split_tile(t, hi, lo, tmp, h1=None):
    """T tile (4 fp32) -> hi pair regs (2), lo pair regs (2): the eight VALU ops of csrc split_pair x 2."""
    ops = [f"v_cvt_pk_f16_f32 v{hi}, v{t}, v{t + 1}", f"v_cvt_pk_f16_f32 v{hi + 1}, v{t + 2}, v{t + 3}"]
    if H1 if h1 is None else h1:
        return ops
    for r in range(4):
        sel = "op_sel:[1,0,0] " if r % 2 else ""
        ops.append(f"v_fma_mix_f32 v{tmp + r}, v{hi + r // 2}, -1.0, v{t + r} {sel}op_sel_hi:[1,0,0]")
    ops += [f"v_cvt_pk_f16_f32 v{lo}, v{tmp}, v{tmp + 1}", f"v_cvt_pk_f16_f32 v{lo + 1}, v{tmp + 2}, v{tmp + 3}"]
    return ops


def by_pairs(streams):
    """Token tiles share the two temporary sets: tiles (0, 1) interleaved, then (2, 3) / tile 2."""
    out = []
    for k in range(0, len(streams), 2):
        out += interleave(*streams[k:k + 2])
    return out


def g1():
    """After the attention block: LayerNorm 1, FFN operands, FFN accumulator seeds (x' + b2) / s_w2."""
    L = layer_norm(S_IA, SIDE_LN1W, SIDE_LN1B, "g1", pre_bias=SIDE_OUTB if DENSE else None)
    A = L.append
    A(f"v_mov_b32 v{V_C}, s{S_IF}")
    A(f"v_mov_b32 v{V_C + 1}, s{S_IF}")
    b2 = lambda buf: PRM(buf, 2)
    n_vg = 2 if NT4 else 3     # k-steps of xb that live in VGPRs (gen_h3_ffn_asm.py XB)
    A(f"ds_read_b128 {vr(b2(0), 4)}, v{V_SLG} offset:{4 * SIDE_B2}")
    for ft in range(8):
        buf = ft % 2
        if ft + 1 < 8:
            A(f"ds_read_b128 {vr(b2(1 - buf), 4)}, v{V_SLG} offset:{4 * SIDE_B2 + 64 * (ft + 1)}")
            A("s_waitcnt lgkmcnt(1)")
        else:
            A("s_waitcnt lgkmcnt(0)")
        for h in range(2):   # b2 / s_w2
            A(f"v_pk_mul_f32 {vr(b2(buf) + 2 * h)}, {vr(b2(buf) + 2 * h)}, {vr(V_C)}")
        ks, odd = ft // 2, ft % 2
        streams = []
        for jt in range(NT):
            s = []
            if ks < n_vg:
                hi, lo = ffn.XB(ks, jt, "h") + 2 * odd, ffn.XB(ks, jt, "l") + 2 * odd
                s += split_tile(T(ft, jt), hi, lo, TMP(jt))
            else:
                # these k-steps live in AGPRs: split into temporaries, then move
                hi, lo = U(jt), U(jt) + 2
                s += split_tile(T(ft, jt), hi, lo, TMP(jt))
                s += [f"v_accvgpr_write_b32 a{ffn.XB(ks, jt, 'h') + 2 * odd + k}, v{hi + k}" for k in range(2)]
                if not H1:
                    s += [f"v_accvgpr_write_b32 a{ffn.XB(ks, jt, 'l') + 2 * odd + k}, v{lo + k}" for k in range(2)]
            streams.append(s)
        # the tiles share TMP / U (two sets): tiles 0 and 1 interleaved, the rest behind them
        L += by_pairs(streams) if NT4 else interleave(streams[0], streams[1]) + streams[2]
        for jt in range(NT):
            t = T(ft, jt)
            for h in range(2):
                A(f"v_pk_fma_f32 {vr(V_ACC0 + 2 * h)}, {vr(t + 2 * h)}, {vr(V_C)}, {vr(b2(buf) + 2 * h)}")
            for r in range(4):
                A(f"v_accvgpr_write_b32 a{ACC(ft, jt) + r}, v{V_ACC0 + r}")
    return L


def transposer_issue(ft, buf):
    """Split the three tiles of feature tile ft and send them through the matrix pipe (K = 16 against the identity)."""
    L = []
    if NT4:
        L += by_pairs([split_tile(T(ft, jt), H(buf, jt, 0), H(buf, jt, 1), TMP(jt)) for jt in range(NT)])
    else:
        L += interleave(split_tile(T(ft, 0), H(buf, 0, 0), H(buf, 0, 1), TMP(0)),
                        split_tile(T(ft, 1), H(buf, 1, 0), H(buf, 1, 1), TMP(1)))
        L += split_tile(T(ft, 2), H(buf, 2, 0), H(buf, 2, 1), TMP(0))
    L.append("s_nop 1")
    for jt in range(NT):
        for part in range(1 if H1 else 2):
            L.append(f"v_mfma_f32_16x16x16_f16 {vr(D(buf, jt, part), 4)}, {vr(H(buf, jt, part))}, {vr(V_IDB)}, 0")
    return L


def transposer_drain(ft, buf):
    """D tiles of feature tile ft (issued a whole tile of work ago) -> packed fp16 -> the attention block's operand AGPRs
    (no LDS in between: MFMA A operands may be AGPRs, and 192 ds_reads per layer disappear with the 32 ds_writes).
    64-token build: -> the images of the wave-private block ([T0 | T1], [T2 | T3] 16 bytes per lane each, hi then lo:
    csrc H3N4_XT_IMG), where the attention block's ds_reads expect them."""
    L = []
    parts = 1 if H1 else 2
    if WIDE:
        # the workgroup's shared tile [feature][192 tokens]: this lane's feature row 16 ft + i16, tokens 4 g .. 4 g + 3 of the
        # wave's token tile jt = 8 contiguous bytes (csrc: "Transposed through the matrix pipe like the 48-token kernel's tile")
        for part in range(parts):
            for jt in range(NT):
                d = D(buf, jt, part)
                img = IMG01(part) + 2 * jt if jt < 2 else IMG2(part)
                L += [f"v_cvt_pk_f16_f32 v{img}, v{d}, v{d + 1}", f"v_cvt_pk_f16_f32 v{img + 1}, v{d + 2}, v{d + 3}"]
        for part in range(parts):
            for jt in range(NT):
                img = IMG01(part) + 2 * jt if jt < 2 else IMG2(part)
                L.append(f"ds_write_b64 v{(V_XTW, V_XTWL)[part]}, {vr(img)} offset:{16 * ft * attn.XT_ROW + 32 * jt}")
        return L
    if NT4:
        for part in range(parts):
            for jt in range(NT):
                d = D(buf, jt, part)
                img = IMG(part, jt // 2) + 2 * (jt % 2)
                L += [f"v_cvt_pk_f16_f32 v{img}, v{d}, v{d + 1}", f"v_cvt_pk_f16_f32 v{img + 1}, v{d + 2}, v{d + 3}"]
        for part in range(parts):
            for pair in range(2):
                L.append(f"ds_write_b128 v{V_PRIV16}, {vr(IMG(part, pair), 4)} offset:{2 * XT_IMG * ft + XT_IMG * part + 1024 * pair}")
        return L
    base = attn.XT_AGPR + 12 * ft
    for part in range(parts):
        for jt in range(2):
            d = D(buf, jt, part)
            L += [f"v_cvt_pk_f16_f32 v{IMG01(part) + 2 * jt}, v{d}, v{d + 1}",
                  f"v_cvt_pk_f16_f32 v{IMG01(part) + 2 * jt + 1}, v{d + 2}, v{d + 3}"]
        d = D(buf, 2, part)
        L += [f"v_cvt_pk_f16_f32 v{IMG2(part)}, v{d}, v{d + 1}", f"v_cvt_pk_f16_f32 v{IMG2(part) + 1}, v{d + 2}, v{d + 3}"]
    for part, (n01, n2) in enumerate((("a0h", "a1h"), ("a0l", "a1l"))[:parts]):
        L += [f"v_accvgpr_write_b32 a{base + attn.XT_OFF[n01] + k}, v{IMG01(part) + k}" for k in range(4)]
        L += [f"v_accvgpr_write_b32 a{base + attn.XT_OFF[n2] + k}, v{IMG2(part) + k}" for k in range(2)]
    return L


def seed_attention(ft):
    """acc <- x' / s_wc for the three tiles of feature tile ft (V_C2 = 1 / s_wc of the layer that follows)."""
    L = []
    for jt in range(NT):
        t = T(ft, jt)
        L += [f"v_pk_mul_f32 {vr(U(jt) + 2 * h)}, {vr(t + 2 * h)}, {vr(V_C2)}" for h in range(2)]
        L += [f"v_accvgpr_write_b32 a{ACC(ft, jt) + r}, v{U(jt) + r}" for r in range(4)]
    return L


def dense_operands(seed_sgpr):
    """Dense model: x' (in T) -> the attention block's split operands xb[ks][jt] = {hi, lo} in a96..a191 (element e of k-step
    ks <-> feature 32 ks + 16 (e / 4) + 4 g + e % 4: the same images G1 builds for the FFN) + accumulator seeds x' / s_o."""
    L = [f"v_mov_b32 v{V_C2}, s{seed_sgpr}", f"v_mov_b32 v{V_C2 + 1}, s{seed_sgpr}"]
    for ft in range(8):
        ks, odd = ft // 2, ft % 2
        streams = []
        for jt in range(NT):
            hi, lo = U(jt), U(jt) + 2
            s = split_tile(T(ft, jt), hi, lo, TMP(jt), h1=False)
            mv = "v_mov_b32 v" if attn.XB_CLS(ks, jt) == "v" else "v_accvgpr_write_b32 a"   # (64-token build: one operand pair lives in VGPRs)
            s += [f"{mv}{attn.XB(ks, jt, 'h') + 2 * odd + k}, v{hi + k}" for k in range(2)]
            s += [f"{mv}{attn.XB(ks, jt, 'l') + 2 * odd + k}, v{lo + k}" for k in range(2)]
            streams.append(s)
        L += by_pairs(streams) if NT4 else interleave(streams[0], streams[1]) + streams[2]
        L += seed_attention(ft)
    return L


def transposer(seed_sgpr=None):
    """x' (in T) -> the attention block's transposed copy + accumulator seeds: issue feature tile ft, drain ft - 1.
    64-token build: what the phase needs in registers is rebuilt first (`seed_sgpr`: 1 / s_wc of the layer that follows)."""
    L = []
    if STATELESS:
        L += [f"v_mov_b32 v{V_C2}, s{seed_sgpr}", f"v_mov_b32 v{V_C2 + 1}, s{seed_sgpr}"]
        L += identity_operand(TMP(0), TMP(1))
    if NT4:
        L += lane_addr(V_PRIV16, "priv", 4, TMP(0))
    if WIDE:
        t0 = TMP(0)
        L += [f"v_mbcnt_lo_u32_b32 v{t0}, -1, 0", f"v_mbcnt_hi_u32_b32 v{t0}, -1, v{t0}",
              f"v_and_b32 v{t0 + 1}, 15, v{t0}", f"v_mul_u32_u24 v{t0 + 1}, {attn.XT_ROW}, v{t0 + 1}",     # row i16 of a feature tile
              f"v_lshrrev_b32 v{t0 + 2}, 4, v{t0}", f"v_lshlrev_b32 v{t0 + 2}, 3, v{t0 + 2}",              # tokens 4 g ..: 8 bytes
              f"v_add3_u32 v{V_XTW}, v{t0 + 1}, v{t0 + 2}, %[xt]",
              f"s_mul_i32 s{S_STAMP_T}, %[wave], {2 * 16 * NT}",                                           # the wave's first token slot
              f"v_add_u32 v{V_XTW}, s{S_STAMP_T}, v{V_XTW}",
              f"v_add_u32 v{V_XTWL}, {attn.XT_LO}, v{V_XTW}"]
    for ft in range(8):
        L += transposer_issue(ft, ft % 2)
        L += seed_attention(ft)
        if ft > 0:
            L += transposer_drain(ft - 1, (ft - 1) % 2)
    L += ["s_nop 7", "s_nop 7"]
    L += transposer_drain(7, 1)
    return L


def entry_transposer():
    """Layer 0: x (in T) -> X^T images + accumulator seeds."""
    return dense_operands(S_IA) if DENSE else transposer(S_IA)


def g2(last):
    """After the FFN: LayerNorm 2, then the next layer's transposed tile and accumulator seeds (or, last layer, the
    split operand images of the out-MLP)."""
    L = layer_norm(S_IF, SIDE_LN2W, SIDE_LN2B, "g2x" if last else "g2")
    A = L.append
    if last and STATELESS:
        L += lane_addr(V_PRIV16, "priv", 4, TMP(0))
    if not last and DENSE:
        return L + dense_operands(S_IA)
    if not last and STATELESS:
        return L + transposer(S_IA)
    for ft in range(8):
        buf = ft % 2
        if last:
            ks, odd = ft // 2, ft % 2
            if NT4:
                L += by_pairs([split_tile(T(ft, jt), XBX(ks, jt, 0) + 2 * odd, XBX(ks, jt, 1) + 2 * odd, TMP(jt)) for jt in range(NT)])
            else:
                L += interleave(split_tile(T(ft, 0), XBX(ks, 0, 0) + 2 * odd, XBX(ks, 0, 1) + 2 * odd, TMP(0)),
                                split_tile(T(ft, 1), XBX(ks, 1, 0) + 2 * odd, XBX(ks, 1, 1) + 2 * odd, TMP(1)))
                L += split_tile(T(ft, 2), XBX(ks, 2, 0) + 2 * odd, XBX(ks, 2, 1) + 2 * odd, TMP(0))
            if odd:
                for jt in range(NT):
                    for part in range(1 if H1 else 2):
                        A(f"ds_write_b128 v{V_PRIV16}, {vr(XBX(ks, jt, part), 4)} offset:{1024 * ((ks * NT + jt) * 2 + part)}")
        else:
            L += transposer_issue(ft, buf)
            L += seed_attention(ft)
            if ft > 0:
                L += transposer_drain(ft - 1, 1 - buf)
    if not last:
        L += ["s_nop 7", "s_nop 7"]
        L += transposer_drain(7, 1)
    return L


N_STAMP = [0]


def stamp(k):
    """H3_ENC_EXPERIMENT=stamps: s_memtime -> dump[40 + 4 l + k] (k = 0..3: attention start / end, FFN start / end) or,
    k = 4, dump[2 + 4 l + 3] (end of the layer), by the wave the caller enabled (%[stampen])."""
    if "stamps" not in EXPERIMENT:
        return []
    N_STAMP[0] += 1
    off = 8 * (40 + k) if k < 4 else 8 * (2 + 3)
    t = S_STAMP_T
    if STATELESS:   # the stamp block's address lives in an SGPR pair (no VGPR survives the embedded blocks)
        return ["s_cmp_eq_u32 %[stampen], 0", f"s_cbranch_scc1 .Lenc_nostamp_{N_STAMP[0]}_%=",
                f"s_memtime {sr(t)}", "s_waitcnt lgkmcnt(0)",
                f"v_mov_b32 v{TMP(0)}, s{t}", f"v_mov_b32 v{TMP(0) + 1}, s{t + 1}", f"v_mov_b32 v{TMP(0) + 2}, 0",
                f"global_store_dwordx2 v{TMP(0) + 2}, {vr(TMP(0))}, {sr(S_DUMP)} offset:{off}",
                f".Lenc_nostamp_{N_STAMP[0]}_%=:"]
    return ["s_cmp_eq_u32 %[stampen], 0", f"s_cbranch_scc1 .Lenc_nostamp_{N_STAMP[0]}_%=",
            "s_memtime s[70:71]", "s_waitcnt lgkmcnt(0)",
            f"v_mov_b32 v{NM(2)}, s70", f"v_mov_b32 v{NM(2) + 1}, s71",
            f"global_store_dwordx2 {vr(V_STAMP)}, {vr(NM(2))}, off offset:{off}",
            f".Lenc_nostamp_{N_STAMP[0]}_%=:"]


def layer_top():
    """All waves are done with the previous layer's side block: wave 0 fetches this layer's (LDS-DMA, nobody waits here:
    it is older than every weight stage of the layer, the first hand-off inside the attention block covers it)."""
    L = ["s_waitcnt lgkmcnt(0)", "s_barrier",
         "s_cmp_lg_u32 %[wave], 0", "s_cbranch_scc1 .Lenc_noside_%="]
    if DENSE1:
        # single buffer: THIS layer's block, now that every wave is done with the previous one's (the barrier above); wave 0 waits
        # for it and a second barrier lets the others in - the attention block reads biases and the in_proj scale at its very entry
        L += [f"s_mov_b32 m0, s{S_SLCUR}", "s_nop 0",
              f"v_mbcnt_lo_u32_b32 v{TMP(0)}, -1, 0", f"v_mbcnt_hi_u32_b32 v{TMP(0)}, -1, v{TMP(0)}", f"v_lshlrev_b32 v{TMP(0)}, 4, v{TMP(0)}"]
        for i in range(4):
            L.append(f"global_load_lds_dwordx4 v{TMP(0)}, {sr(S_SIDE)}" + (f" offset:{1024 * i}" if i else ""))
        L += [f"s_add_u32 m0, s{S_SLCUR}, 4096", f"v_add_u32 v{TMP(0) + 1}, 4096, v{TMP(0)}",
              f"global_load_lds_dwordx4 v{TMP(0) + 1}, {sr(S_SIDE)}",
              f"s_add_u32 s{S_SIDE}, s{S_SIDE}, %[sidestride]", f"s_addc_u32 s{S_SIDE + 1}, s{S_SIDE + 1}, 0",
              "s_waitcnt vmcnt(0)", ".Lenc_noside_%=:", "s_barrier"]
        return L
    if DENSE:
        # double-buffered: this layer's block arrived a layer ago (layer 0's: the kernel's prologue); fetch the NEXT layer's
        # into the other buffer, which every wave has finished reading (LayerNorm 2 of the previous layer; the barrier above)
        L += [f"s_cmp_eq_u32 s{S_LAYER}, 1", "s_cbranch_scc1 .Lenc_noside_%=", f"s_mov_b32 m0, s{S_SLOTHER}", "s_nop 0"]
    else:
        L += ["s_mov_b32 m0, %[sl]", "s_nop 0"]
    if STATELESS:   # SGPR base + lane offset
        L += [f"v_mbcnt_lo_u32_b32 v{TMP(0)}, -1, 0", f"v_mbcnt_hi_u32_b32 v{TMP(0)}, -1, v{TMP(0)}", f"v_lshlrev_b32 v{TMP(0)}, 4, v{TMP(0)}"]
        for i in range(min(SIDE_CHUNKS, 4)):
            L.append(f"global_load_lds_dwordx4 v{TMP(0)}, {sr(S_SIDE)}" + (f" offset:{1024 * i}" if i else ""))
        if SIDE_CHUNKS == 5:   # the instruction offset is 13 bits signed: the fifth KiB through the lane offset and m0
            L += [f"s_add_u32 m0, s{S_SLOTHER}, 4096", f"v_add_u32 v{TMP(0) + 1}, 4096, v{TMP(0)}",
                  f"global_load_lds_dwordx4 v{TMP(0) + 1}, {sr(S_SIDE)}"]
        L += [f"s_add_u32 s{S_SIDE}, s{S_SIDE}, %[sidestride]", f"s_addc_u32 s{S_SIDE + 1}, s{S_SIDE + 1}, 0", ".Lenc_noside_%=:"]
        return L
    for i in range(SIDE_CHUNKS):
        L.append(f"global_load_lds_dwordx4 {vr(V_SIDE)}, off" + (f" offset:{1024 * i}" if i else ""))
    L += [f"v_lshl_add_u64 {vr(V_SIDE)}, {vr(V_SIDE)}, 0, %[sidestride]", ".Lenc_noside_%=:"]
    return L


def generate():
    L = []
    A = L.append
    # ---- persistent registers
    if DENSE1:
        A(f"s_mov_b32 s{S_SLCUR}, %[sl]")
        A(f"s_mov_b64 {sr(S_SIDE)}, %[side]")                              # layer_top() fetches every layer's block itself
        L += lane_addr(V_PRIV16, "priv", 4, TMP(0))
    elif DENSE:
        A(f"s_mov_b32 s{S_SLCUR}, %[sl]")
        A(f"s_add_u32 s{S_SLOTHER}, %[sl], {SIDE_LDS_BYTES}")
        A(f"s_mov_b64 {sr(S_SIDE)}, %[side]")                              # layer 0's block is in flight already:
        A(f"s_add_u32 s{S_SIDE}, s{S_SIDE}, %[sidestride]")                # layer_top() fetches from layer 1 on
        A(f"s_addc_u32 s{S_SIDE + 1}, s{S_SIDE + 1}, 0")
        L += lane_addr(V_PRIV16, "priv", 4, TMP(0))
    elif STATELESS:
        A(f"s_mov_b64 {sr(S_SFB)}, %[sf]")
        A(f"s_mov_b64 {sr(S_SIDE)}, %[side]")
        L += lane_addr(V_PRIV16, "priv", 4, TMP(0))
    else:
        A(f"v_mbcnt_lo_u32_b32 v{TMP(0)}, -1, 0")
        A(f"v_mbcnt_hi_u32_b32 v{TMP(0)}, -1, v{TMP(0)}")                 # lane
        A(f"v_lshlrev_b32 v{V_PRIV8}, 3, v{TMP(0)}")
        A(f"v_add_u32 v{V_PRIV8}, %[priv], v{V_PRIV8}")
        A(f"v_lshlrev_b32 v{V_PRIV16}, 4, v{TMP(0)}")
        A(f"v_add_u32 v{V_PRIV16}, %[priv], v{V_PRIV16}")
        A(f"v_lshrrev_b32 v{TMP(0) + 1}, 4, v{TMP(0)}")                   # g
        A(f"v_lshlrev_b32 v{V_SLG}, 4, v{TMP(0) + 1}")
        A(f"v_add_u32 v{V_SLG}, %[sl], v{V_SLG}")                        # side block + 16 g bytes
        A(f"v_mov_b32 v{V_PAD}, %[padm]")
        A(f"v_mov_b64 {vr(V_SFB)}, %[sf]")
        A(f"v_mov_b64 {vr(V_SIDE)}, %[side]")
        # identity operand of the transposing MFMA: element e of lane (i16, g) = (i16 == 4 g + e)
        A(f"v_and_b32 v{TMP(0) + 2}, 15, v{TMP(0)}")                      # i16
        A(f"v_lshlrev_b32 v{TMP(0) + 3}, 2, v{TMP(0) + 1}")               # 4 g
        A(f"v_sub_u32 v{TMP(0) + 2}, v{TMP(0) + 2}, v{TMP(0) + 3}")       # i16 - 4 g  (0..3 -> a one in that element)
        A(f"v_mov_b32 v{V_IDB}, 0")
        A(f"v_mov_b32 v{V_IDB + 1}, 0")
        A(f"v_mov_b32 v{TMP(1) + 2}, 0x3c00")
        A(f"v_mov_b32 v{TMP(1) + 3}, 0x3c000000")
        for e, (reg, src) in enumerate(((V_IDB, TMP(1) + 2), (V_IDB, TMP(1) + 3), (V_IDB + 1, TMP(1) + 2), (V_IDB + 1, TMP(1) + 3))):
            A(f"v_cmp_eq_u32 vcc, {e}, v{TMP(0) + 2}")
            A(f"v_cndmask_b32 v{reg}, v{reg}, v{src}, vcc")
        A(f"v_mov_b32 v{V_ONE}, 1.0")
        A(f"v_mov_b32 v{V_ONE + 1}, 1.0")
    A(f"s_mov_b32 s{S_LAYER}, %[layers]")
    A(f"s_mov_b32 s{S_PADT}, %[padt]")
    A(f"s_mov_b64 s[{S_SCPTR}:{S_SCPTR + 1}], %[scales]")
    A(f"s_mov_b32 s{S_EPS}, %[eps]")
    if DENSE:
        # the residual scale of the attention block is out_proj's: floats 2 + 3 L + l behind %[scales] (h3_pack_weights)
        A(f"s_mul_i32 s{S_OSC}, %[layers], 12")
        A(f"s_add_u32 s{S_OSC}, s{S_OSC}, 8")
        A(f"s_add_u32 s{S_OSC}, s{S_SCPTR}, s{S_OSC}")
        A(f"s_addc_u32 s{S_OSC + 1}, s{S_SCPTR + 1}, 0")
        A(f"s_load_dword s{S_SCA}, s[{S_OSC}:{S_OSC + 1}], 0x0")
    else:
        A(f"s_load_dword s{S_SCA}, s[{S_SCPTR}:{S_SCPTR + 1}], 0x0")
    A(f"s_load_dword s{S_SCF}, s[{S_SCPTR}:{S_SCPTR + 1}], 0x8")
    # ---- x in: 24 register images from the wave-private block
    for ft in range(8):
        for jt in range(NT):
            A(f"ds_read_b128 {vr(T(ft, jt), 4)}, v{V_PRIV16} offset:{1024 * (ft * NT + jt)}")
    A("s_waitcnt lgkmcnt(0)")
    if WIDE:
        # the shared tile lies over all four wave-private blocks: nobody writes it before every wave has read its x images
        A("s_barrier")
    A(f"s_sub_u32 s{S_IA}, 0x7f000000, s{S_SCA}")                    # scales are powers of two: 1 / s by exponent
    A(f"s_sub_u32 s{S_IF}, 0x7f000000, s{S_SCF}")
    if not STATELESS:
        A(f"v_mov_b32 v{V_C2}, s{S_IA}")
        A(f"v_mov_b32 v{V_C2 + 1}, s{S_IA}")
    L += entry_transposer()
    # ---- layer loop
    if "stamps" in EXPERIMENT:
        if STATELESS:
            A(f"s_mov_b64 {sr(S_DUMP)}, %[dump]")
        else:
            A(f"v_mov_b64 {vr(V_STAMP)}, %[dump]")
        A(f"s_mov_b32 s{S_STAMP_STEP}, 32")
        A(f"s_mov_b32 s{S_STAMP_STEP + 1}, 0")
    A(".Lenc_layer_%=:")
    L += stamp(0)
    L += layer_top()
    L += attn.generate()
    L += stamp(1)
    L += g1()
    L += stamp(2)
    L += ffn.generate()
    L += stamp(3)
    A(f"s_sub_u32 s{S_LAYER}, s{S_LAYER}, 1")
    A(f"s_cmp_eq_u32 s{S_LAYER}, 0")
    A("s_cbranch_scc1 .Lenc_last_%=")
    # next layer's scales (12 bytes further), fragments
    A(f"s_add_u32 s{S_SCPTR}, s{S_SCPTR}, 12")
    A(f"s_addc_u32 s{S_SCPTR + 1}, s{S_SCPTR + 1}, 0")
    if DENSE:
        A(f"s_add_u32 s{S_OSC}, s{S_OSC}, 4")
        A(f"s_addc_u32 s{S_OSC + 1}, s{S_OSC + 1}, 0")
        A(f"s_load_dword s{S_NA}, s[{S_OSC}:{S_OSC + 1}], 0x0")
    else:
        A(f"s_load_dword s{S_NA}, s[{S_SCPTR}:{S_SCPTR + 1}], 0x0")
    A(f"s_load_dword s{S_NF}, s[{S_SCPTR}:{S_SCPTR + 1}], 0x8")
    if DENSE:
        pass
    elif STATELESS:
        A(f"s_add_u32 s{S_SFB}, s{S_SFB}, %[sfstride]")
        A(f"s_addc_u32 s{S_SFB + 1}, s{S_SFB + 1}, %[sfstridehi]")
    else:
        A(f"v_lshl_add_u64 {vr(V_SFB)}, {vr(V_SFB)}, 0, %[sfstride]")
    A("s_waitcnt lgkmcnt(0)")
    A(f"s_sub_u32 s{S_IA}, 0x7f000000, s{S_NA}")
    if not STATELESS:
        A(f"v_mov_b32 v{V_C2}, s{S_IA}")
        A(f"v_mov_b32 v{V_C2 + 1}, s{S_IA}")
    L += g2(False)
    L += stamp(4)
    if "stamps" in EXPERIMENT:
        if STATELESS:
            A(f"s_add_u32 s{S_DUMP}, s{S_DUMP}, s{S_STAMP_STEP}")
            A(f"s_addc_u32 s{S_DUMP + 1}, s{S_DUMP + 1}, 0")
        else:
            A(f"v_lshl_add_u64 {vr(V_STAMP)}, {vr(V_STAMP)}, 0, s[72:73]")
    A(f"s_mov_b32 s{S_SCA}, s{S_NA}")
    A(f"s_mov_b32 s{S_SCF}, s{S_NF}")
    A(f"s_sub_u32 s{S_IF}, 0x7f000000, s{S_SCF}")
    if DENSE and not DENSE1:   # the next layer's side block is the other buffer
        A(f"s_mov_b32 s{S_SWAP}, s{S_SLCUR}")
        A(f"s_mov_b32 s{S_SLCUR}, s{S_SLOTHER}")
        A(f"s_mov_b32 s{S_SLOTHER}, s{S_SWAP}")
    A("s_branch .Lenc_layer_%=")
    A(".Lenc_last_%=:")
    L += g2(True)
    A("s_waitcnt lgkmcnt(0)")
    L += stamp(4)
    return L


def main():
    _check_register_map()
    lines = generate()
    out_dir = "timewarp_amd/csrc"
    for a in sys.argv[1:]:
        if a.startswith("--out-dir="):
            out_dir = a.split("=", 1)[1]
    ng = getattr(attn, "NG", 5)
    mode = (" --mode=windowed" if WINDOWED else "") + (" --nt=4" if NT4 else "") + (" --pair" if PAIR else "") + \
        ((" --wide" + (f" --ng={ng}" if ng != 5 else "")) if WIDE else "") + (" --dense" if DENSE else "") + (" --h1" if H1 else "") + \
        (" --ring6" if R6 else "")
    fam = ("h1" if H1 else "h3") + ("r" if R6 else "") + ("n4" if NT4 else "") + ("p" if PAIR else "") + ((f"w{ng}" if ng != 5 else "w") if WIDE else "") + ("d" if DENSE else "")
    base = os.path.join(out_dir, f"tw_{fam}_encw_asm.inc" if WINDOWED else f"tw_{fam}_enc_asm.inc")
    out = [f"// GENERATED by tools/gen_h3_enc_asm.py{mode} - do not edit.  Body of the encoder-stack asm statement."]
    out += ['"' + l + '\\n\\t"' for l in lines]
    open(base, "w").write("\n".join(out) + "\n")
    if not WINDOWED:
        # (wide: the attention block's fragments reach a159, the FFN's operands a119, nothing of the glue lives in AGPRs)
        clob = [f'"v{i}"' for i in range(N_V)] + [f'"a{i}"' for i in range(160 if WIDE else 248 if DENSE1 else 192)] + [f'"s{i}"' for i in range(S_LO, 100)] + \
               ['"vcc"', '"scc"', '"memory"']
        cl = [f"// GENERATED by tools/gen_h3_enc_asm.py{mode} - clobber list of the encoder-stack asm statement."]
        for i in range(0, len(clob), 12):
            cl.append(", ".join(clob[i:i + 12]) + ("," if i + 12 < len(clob) else ""))
        open(os.path.join(out_dir, f"tw_{fam}_enc_clobbers.inc"), "w").write("\n".join(cl) + "\n")
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    print(f"enc{mode}: {len(lines)} instructions, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
