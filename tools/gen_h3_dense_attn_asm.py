#!/usr/bin/env python3
"""Generate timewarp_amd/csrc/tw_h3_attnd_asm.inc: the hand-scheduled dense-softmax attention block of one encoder layer of
the split-fp16 net-block kernel (transformer_nvp: nn.MultiheadAttention with 8 heads of 16, 48 tokens per wave, gfx950).

Dataflow (everything in registers):
  q_h, k_h = W_{q,k}[16 h ..] . x^T   standard orientation, D tile [feature][token] = K = 16 MFMA operand layout
  v_h      = x . W_v[16 h ..]^T       operands swapped: D tile [token][feature], lane = feature = A operand of P.V
  S^T[key][query] = k_h q_h^T / 4 (K = 16 MFMAs, 3-term split), masked softmax in the accumulator registers,
  O^T = V^T P^T (K = 32 for key tiles 0, 1 + K = 16 for tile 2), two heads' O = one k-step of out_proj.

Schedule - one wave per SIMD, so VALU work only hides in the shadow of MFMAs; per head h:
  A  stage V(h)               36 MFMAs  |  scale / bias / split of q_h, k_h               72 VALU
  B  S(h)                     27 MFMAs  |  scale / bias / split of v_h                    36 VALU
  C  stages Q(h+1), K(h+1)    72 MFMAs  |  mask, max, exp, sum, split of P(h)            ~330 VALU   (bare after the last head)
  D  P.V(h)                   18 MFMAs  |  1/sum, split of O(h)                           48 VALU
  E  (odd h) out_proj stages  72 MFMAs
The accumulators of q / k / v are single-buffered: their consumers (the splits) are done before the next head's stage
writes them.  Weight-stage order in the stream (h3_pack_weights, dense branch):  q0 k0 | v_h  q_{h+1} k_{h+1}  [out_proj of
the pair after an odd h] ...  Stage hand-off, ring and DMA protocol as in gen_h3_attn_asm.py.

Register map (private to the asm statement):
  v0..v31    weight tile slots p = 0..3: hi v[8p..], lo v[8p+4..]
  v32..v67   accumulators QA[jt] KA[jt] VA[jt]
  v68..v91   q / k operands (K = 16): QH QL KH KL [jt] x 2 registers
  v92..v103  v operands: VH01 VL01 (K = 32, key tiles 0 and 1), VH2 VL2 (K = 16, tile 2)
  v104..v139 S^T accumulators SC[jt][mt]; P operands are written over them: PL01 (+0), PH01 (+4), PL2 (+8), PH2 (+10)
  v140..v142 1 / sum per query tile; v144..v167 O32[jt], O16[jt]; v168..v191 out_proj operand OB[jt] = {h 4, l 4}
  v192..v203 temporaries; v204..v211 bias registers; v212.. addresses
  a0..a95    y[ot][jt];  a96..a191 xb[ks][jt] = {h, l} (B operand of q / k stages, A operand of the v stage)"""
import os
import sys

# --nt=4 (r06): 64-token waves - ONE molecule of 49-64 atoms per wave (csrc H3N4_*), a ring of three stage buffers.  Keys = 64 = two
# K = 32 groups (no K = 16 tail).  What does not fit as it stands is the score tile: 4 query tiles x 4 key tiles x 4 registers beside
# y (a0..a127) and the split activations (a128..a255) - so a head runs in two QUERY HALVES (tiles 0, 1 then 2, 3) over 32 score
# registers:   V(h) | S(A) | Q(h+1) + softmax(A) | P.V(A) | S(B) | K(h+1) + softmax(B) | P.V(B) | [out_proj after an odd h]
# Register map:  v0..v31 tile slots;  v32..v79 QA KA VA[jt] (the P.V accumulators take VA's registers: v is split into operands
# before the first P.V of the head);  v80..v111 QH QL KH KL[jt];  v112..v127 VH01 VL01 VH23 VL23;  v128..v159 SC[jl][mt] of the half in
# flight (P operands written over them: PL01 PL23 PH01 PH23);  v160..v164 1 / sum, reduction copy;  v168..v199 OB[jt];
# v200..v211 temporaries;  v212..v220 biases;  v221..v229 addresses;  a0..a127 y;  a128..a255 xb.  Key masks: ONE pair of words
# (%[m0l], %[m0h]: keys 0-31, 32-63 of the wave's molecule, shifted by the lane group) serves all four query tiles.
NT4 = "--nt=4" in sys.argv
NT = 4 if NT4 else 3
AHEAD = 3 if NT4 else 5
RING = AHEAD
STAGE, TILES = 9216, 8192
H3D_INB = 656              # float offset of in_proj bias [384] in the layer's side block; sc_in at 640
SLOT = lambda p, part: 8 * p + (0 if part == "h" else 4)
QA = lambda jt: 32 + 4 * jt
KA = lambda jt: 44 + 4 * jt
VA = lambda jt: 56 + 4 * jt
QH = lambda jt: 68 + 2 * jt
QL = lambda jt: 74 + 2 * jt
KH = lambda jt: 80 + 2 * jt
KL = lambda jt: 86 + 2 * jt
VH01, VL01, VH2, VL2 = 92, 96, 100, 102
SC = lambda jt, mt: 104 + 12 * jt + 4 * mt
PL01 = lambda jt: 104 + 12 * jt
PH01 = lambda jt: 108 + 12 * jt
PL2 = lambda jt: 112 + 12 * jt
PH2 = lambda jt: 114 + 12 * jt
RS = lambda jt: 140 + jt
V_RED = 143                # second copy for the cross-lane reductions
O32 = lambda jt: 144 + 4 * jt
O16 = lambda jt: 156 + 4 * jt
OB = lambda jt, part: 168 + 8 * jt + (0 if part == "h" else 4)
V_T = 192                  # v192..v203
V_BQ, V_BK, V_BV = 204, 208, 203   # BV shares the last temporary slot's neighbour (v203 is never used as a temporary)
V_TILE, V_LANE16, V_SLQ, V_SLV, V_GN, V_TMP = 212, 213, 214, 215, 216, 218
N_V, N_A = 220, 192
YACC = lambda ot, jt: 4 * (3 * ot + jt)
XB = lambda ks, jt, part: 96 + 8 * (3 * ks + jt) + (0 if part == "h" else 4)
if NT4:
    QA = lambda jt: 32 + 4 * jt
    KA = lambda jt: 48 + 4 * jt
    VA = lambda jt: 64 + 4 * jt
    QH = lambda jt: 80 + 2 * jt
    QL = lambda jt: 88 + 2 * jt
    KH = lambda jt: 96 + 2 * jt
    KL = lambda jt: 104 + 2 * jt
    VH01, VL01, VH23, VL23 = 112, 116, 120, 124
    SC = lambda jt, mt: 128 + 16 * (jt % 2) + 4 * mt          # the half in flight: query tiles jt % 2
    PL01 = lambda jt: 128 + 16 * (jt % 2)
    PL23 = lambda jt: 132 + 16 * (jt % 2)
    PH01 = lambda jt: 136 + 16 * (jt % 2)
    PH23 = lambda jt: 140 + 16 * (jt % 2)
    RS = lambda jt: 160 + jt
    V_RED = 164
    O32 = lambda jt: VA(jt)                                   # P.V accumulators: v's registers (dead after prep_v)
    OB = lambda jt, part: 168 + 8 * jt + (0 if part == "h" else 4)
    V_T = 200
    V_BQ, V_BK, V_BV = 212, 216, 220
    V_TILE, V_LANE16, V_SLQ, V_SLV, V_GN, V_TMP = 221, 222, 223, 224, 226, 228
    N_V, N_A = 230, 256   # (N_V: this block's own registers; the operand pair in v232.. / v240.. belongs to the caller's map)
    YACC = lambda ot, jt: 4 * (4 * ot + jt)
    # the split activations: a128..a247, and the LAST operand pair (k-step 3, token tile 3) in VGPRs v232..v235 / v240..v243 - registers
    # nothing touches between the glue's operand pass and the end of this block - so that a248..a255 stay free in every statement
    # of the kernel: hipcc parks what it carries through the prologue there; with all 256 AGPRs taken it went to scratch
    XB = lambda ks, jt, part: ((232 if part == "h" else 240) if (ks, jt) == (3, 3) else 128 + 8 * (4 * ks + jt) + (0 if part == "h" else 4))
    XB_CLS = lambda ks, jt: "v" if (ks, jt) == (3, 3) else "a"
    N_A = 248
if not NT4:
    XB_CLS = lambda ks, jt: "a"
S_OFF, S_REL, S_W2048, S_STRIDE, S_AUXOFF, S_END, S_PAIR = 84, 85, 86, 88, 90, 92, 93
S_SCIN, S_SCQ, S_LOG2E, S_MASKED = 94, 95, 96, 97
EXPERIMENT = set(filter(None, os.environ.get("H3_ATTN_EXPERIMENT", "").split(",")))
# Set by tools/gen_h3_enc_asm.py --dense, which embeds this block in the asm statement of the whole encoder stack: the split
# activations are already in a96..a191 (written there by the glue, no trip through the LDS), the accumulators arrive holding
# the residual (x / s_o) instead of zeros, and y stays in a0..a95.
FUSED = False
SL = "%[sl]"               # LDS address of the layer's side block (the encoder-stack statement double-buffers it: an SGPR there)


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(base, n=4):
    return f"a[{base}:{base + n - 1}]"


def mfma32(d, a, b, zero=False, dcls="v", acls="v", bcls="v"):
    R = lambda c, x: ar(x) if c == "a" else vr(x)
    return f"v_mfma_f32_16x16x32_f16 {R(dcls, d)}, {R(acls, a)}, {R(bcls, b)}, {'0' if zero else R(dcls, d)}"


def mfma16(d, a, b, zero=False):
    return f"v_mfma_f32_16x16x16_f16 {vr(d)}, {vr(a, 2)}, {vr(b, 2)}, {'0' if zero else vr(d)}"


def tile_reads(pair):
    return [f"ds_read_b128 {vr(SLOT(pair, 'h'))}, v{V_TILE} offset:{2048 * pair}",
            f"ds_read_b128 {vr(SLOT(pair, 'l'))}, v{V_TILE} offset:{2048 * pair + 1024}"]


def handoff(next_reads, label, aux):
    h = [
        f"s_mov_b32 s{S_REL}, s{S_OFF}",
        f"s_add_u32 s{S_OFF}, s{S_OFF}, {STAGE}",
        [f"s_cmp_eq_u32 s{S_OFF}, s{S_END}", f"s_cselect_b32 s{S_OFF}, %[ring], s{S_OFF}"],
    ]
    if next_reads:
        h += [f"v_add_u32 v{V_TILE}, s{S_OFF}, v{V_LANE16}"] + tile_reads(0) + tile_reads(1)
    h += [
        f"s_add_u32 m0, s{S_REL}, s{S_W2048}",
        "s_nop 0",
        f"global_load_lds_dwordx4 {vr(V_GN, 2)}, off",
        f"global_load_lds_dwordx4 {vr(V_GN, 2)}, off offset:1024",
    ]
    if aux:
        # the last five hand-offs of the section fetch the FFN's first stages, whose A0 stages carry a bias / scale block:
        # wave 0 moves it in every hand-off of the last head pair (harmless where the stage has none)
        h += [["s_cmp_lg_u32 %[wave], 0",
               f"s_cbranch_scc1 .Lh3atd_noaux_{label}_%=",
               f"s_cmp_lg_u32 s{S_PAIR}, 1",
               f"s_cbranch_scc1 .Lh3atd_noaux_{label}_%=",
               f"v_lshl_add_u64 {vr(V_TMP, 2)}, {vr(V_GN, 2)}, 0, s[{S_AUXOFF}:{S_AUXOFF + 1}]",
               f"s_add_u32 m0, s{S_REL}, {TILES}",
               "s_nop 0",
               f"global_load_lds_dwordx4 {vr(V_TMP, 2)}, off",
               f".Lh3atd_noaux_{label}_%=:"]]
    h += [f"v_lshl_add_u64 {vr(V_GN, 2)}, {vr(V_GN, 2)}, 0, s[{S_STRIDE}:{S_STRIDE + 1}]"]
    if "nodma" in EXPERIMENT:  # timing experiment (results wrong): no weight DMA inside the section
        h = [x for x in h if isinstance(x, str) and not x.startswith("global_load_lds")]
    return h


def weave(mfmas, valu, misc, valu_per=2, misc_per=3, skip=0):
    out = []
    valu, misc = list(valu), list(misc)
    n = len(mfmas)

    def emit(item):
        out.extend(item if isinstance(item, list) else [item])

    for i, m in enumerate(mfmas):
        out.append(m)
        if i < skip:
            continue
        left = n - i
        for _ in range(min(valu_per, -(-len(valu) // left)) if valu else 0):
            emit(valu.pop(0))
        for _ in range(min(misc_per, -(-len(misc) // left)) if misc else 0):
            emit(misc.pop(0))
    for item in valu + misc:
        emit(item)
    return out


def stage(groups, valu, label, next_reads=True, aux=True, skip=0, pre_barrier=(), valu_per=2):
    """One 4-pair weight stage: `groups[p]` = the MFMAs of tile pair p; `valu` is woven under them; `pre_barrier`: LDS reads
    issued in the second pair's shadow (they have returned by the lgkmcnt(0) in front of the barrier)."""
    valu = list(valu)
    share = -(-len(valu) // 4)
    parts = [valu[i * share:(i + 1) * share] for i in range(4)]
    out = ["s_waitcnt lgkmcnt(2)"]
    out += weave(groups[0], parts[0], tile_reads(2), skip=skip, valu_per=valu_per)
    out.append("s_waitcnt lgkmcnt(2)")
    out += weave(groups[1], parts[1], tile_reads(3) + list(pre_barrier), valu_per=valu_per)
    out.append(f"s_waitcnt vmcnt({2 * (AHEAD - 2)}) lgkmcnt(0)")   # stages s + 2 .. s + AHEAD - 1 may stay in flight, two DMAs each
    if "nobarrier" not in EXPERIMENT:
        out.append("s_barrier")
    out += weave(groups[2], parts[2], handoff(next_reads, label, aux), valu_per=valu_per)
    out += weave(groups[3], parts[3], [], valu_per=valu_per)
    return out


def qk_groups(acc):
    """q_h / k_h stage: pair p = k-step; acc[jt] += tile(p) . xb[p][jt]  (3-term split)."""
    groups = []
    for ks in range(4):
        g = []
        for k, (ap, bp) in enumerate((("h", "h"), ("h", "l"), ("l", "h"))):
            for jt in range(NT):
                g.append(mfma32(acc(jt), SLOT(ks, ap), XB(ks, jt, bp), zero=(ks == 0 and k == 0), bcls=XB_CLS(ks, jt)))
        groups.append(g)
    return groups


def v_groups():
    """v_h stage, operands swapped: VA[jt] += xb[p][jt] . tile(p)  ->  [token][feature]."""
    groups = []
    for ks in range(4):
        g = []
        for k, (xp, wp) in enumerate((("h", "h"), ("l", "h"), ("h", "l"))):
            for jt in range(NT):
                g.append(mfma32(VA(jt), XB(ks, jt, xp), SLOT(ks, wp), zero=(ks == 0 and k == 0), acls=XB_CLS(ks, jt)))
        groups.append(g)
    return groups


def out_groups(half):
    groups = []
    for p in range(4):
        g = []
        for ap, bp in (("h", "h"), ("h", "l"), ("l", "h")):
            for jt in range(NT):
                g.append(mfma32(YACC(4 * half + p, jt), SLOT(p, ap), OB(jt, bp), dcls="a"))
        groups.append(g)
    return groups


def split_pair(src, dst_h, dst_l, t):
    """4 fp32 values src..src+3 -> 2 registers of fp16 hi at dst_h, 2 of lo at dst_l (t: 4 temporaries): 8 VALU ops."""
    ops = [f"v_cvt_pk_f16_f32 v{dst_h}, v{src}, v{src + 1}", f"v_cvt_pk_f16_f32 v{dst_h + 1}, v{src + 2}, v{src + 3}"]
    for r in range(4):
        sel = "op_sel:[1,0,0] " if r % 2 else ""
        ops.append(f"v_fma_mix_f32 v{t + r}, v{dst_h + r // 2}, -1.0, v{src + r} {sel}op_sel_hi:[1,0,0]")
    ops += [f"v_cvt_pk_f16_f32 v{dst_l}, v{t}, v{t + 1}", f"v_cvt_pk_f16_f32 v{dst_l + 1}, v{t + 2}, v{t + 3}"]
    return ops


def prep_qk():
    """(QA * sc/4 + bq/4) -> QH / QL, (KA * sc + bk) -> KH / KL."""
    ops = []
    for jt in range(NT):
        t = V_T + 4 * (jt % 2)
        ops += [f"v_fma_f32 v{QA(jt) + r}, v{QA(jt) + r}, s{S_SCQ}, v{V_BQ + r}" for r in range(4)]
        ops += split_pair(QA(jt), QH(jt), QL(jt), t)
    for jt in range(NT):
        t = V_T + 4 * (jt % 2)
        ops += [f"v_fma_f32 v{KA(jt) + r}, v{KA(jt) + r}, s{S_SCIN}, v{V_BK + r}" for r in range(4)]
        ops += split_pair(KA(jt), KH(jt), KL(jt), t)
    return ops


def prep_v():
    ops = []
    for jt in range(NT):
        ops += [f"v_fma_f32 v{VA(jt) + r}, v{VA(jt) + r}, s{S_SCIN}, v{V_BV}" for r in range(4)]
    ops += split_pair(VA(0), VH01, VL01, V_T) + split_pair(VA(1), VH01 + 2, VL01 + 2, V_T + 4)
    if NT4:
        return ops + split_pair(VA(2), VH23, VL23, V_T) + split_pair(VA(3), VH23 + 2, VL23 + 2, V_T + 4)
    ops += split_pair(VA(2), VH2, VL2, V_T)
    return ops


def s_mfmas(half=None):
    """S^T[key tile mt][query tile jt] = k_h[mt] q_h[jt]^T (K = 16, three terms).  half (64-token build): query tiles 2 half, 2 half + 1."""
    out = []
    for jt in (range(NT) if half is None else (2 * half, 2 * half + 1)):
        for k, (kp, qp) in enumerate(((KH, QH), (KH, QL), (KL, QH))):
            for mt in range(NT):
                out.append(mfma16(SC(jt, mt), kp(mt), qp(jt), zero=(k == 0)))
    return out


def reduce4(op, v):
    """v <- op over the four lanes sharing (lane & 15), through V_RED (v_permlane16_swap / v_permlane32_swap)."""
    return [f"v_mov_b32 v{V_RED}, v{v}", "s_nop 1", f"v_permlane16_swap_b32 v{v}, v{V_RED}", f"{op} v{v}, v{v}, v{V_RED}",
            f"v_mov_b32 v{V_RED}, v{v}", "s_nop 1", f"v_permlane32_swap_b32 v{v}, v{V_RED}", f"{op} v{v}, v{v}, v{V_RED}"]


def soft(jt):
    """Masked softmax of query tile jt over its 12 accumulator elements (keys 16 mt + 4 g + r) and the split of P,
    unnormalised (1 / sum goes to RS[jt]).  Mask bits: %[mNl] bit r (mt 0), bit 16 + r (mt 1); %[mNh] bit r (mt 2)."""
    el = [(mt, r) for mt in range(NT) for r in range(4)]
    m, mx, sm = V_T + 8, V_T + 9, V_T + 10
    ops = []
    for i, (mt, r) in enumerate(el):
        word, bit = (f"%[m{jt}l]", 16 * mt + r) if mt < 2 else (f"%[m{jt}h]", r)
        ops += [f"v_bfe_i32 v{m}, {word}, {bit}, 1", f"v_bfi_b32 v{SC(jt, mt) + r}, v{m}, v{SC(jt, mt) + r}, s{S_MASKED}"]
        ops.append(f"v_mov_b32 v{mx}, v{SC(jt, mt) + r}" if i == 0 else f"v_max_f32 v{mx}, v{mx}, v{SC(jt, mt) + r}")
    ops += reduce4("v_max_f32", mx)
    ops.append(f"v_mul_f32 v{mx}, s{S_LOG2E}, v{mx}")
    for i, (mt, r) in enumerate(el):
        x = SC(jt, mt) + r
        ops += [f"v_fma_f32 v{x}, v{x}, s{S_LOG2E}, -v{mx}", f"v_exp_f32 v{x}, v{x}"]
    for i, (mt, r) in enumerate(el):   # sums after all exps: a transcendental's result is not read by the very next op
        x = SC(jt, mt) + r
        ops.append(f"v_mov_b32 v{sm}, v{x}" if i == 0 else f"v_add_f32 v{sm}, v{sm}, v{x}")
    ops += reduce4("v_add_f32", sm)
    ops.append(f"v_rcp_f32 v{RS(jt)}, v{sm}")
    # split, in place: hi packs to temporaries, residuals in place, lo packs over the low registers, hi packs moved behind
    t = V_T
    base = SC(jt, 0)
    for k in range(4):
        ops.append(f"v_cvt_pk_f16_f32 v{t + k}, v{base + 2 * k}, v{base + 2 * k + 1}")
    for k in range(8):
        sel = "op_sel:[1,0,0] " if k % 2 else ""
        ops.append(f"v_fma_mix_f32 v{base + k}, v{t + k // 2}, -1.0, v{base + k} {sel}op_sel_hi:[1,0,0]")
    for k in range(4):
        ops.append(f"v_cvt_pk_f16_f32 v{PL01(jt) + k}, v{base + 2 * k}, v{base + 2 * k + 1}")
    for k in range(4):
        ops.append(f"v_mov_b32 v{PH01(jt) + k}, v{t + k}")
    b2 = SC(jt, 2)
    for k in range(2):
        ops.append(f"v_cvt_pk_f16_f32 v{t + 4 + k}, v{b2 + 2 * k}, v{b2 + 2 * k + 1}")
    for k in range(4):
        sel = "op_sel:[1,0,0] " if k % 2 else ""
        ops.append(f"v_fma_mix_f32 v{b2 + k}, v{t + 4 + k // 2}, -1.0, v{b2 + k} {sel}op_sel_hi:[1,0,0]")
    for k in range(2):
        ops.append(f"v_cvt_pk_f16_f32 v{PL2(jt) + k}, v{b2 + 2 * k}, v{b2 + 2 * k + 1}")
    for k in range(2):
        ops.append(f"v_mov_b32 v{PH2(jt) + k}, v{t + 4 + k}")
    return ops


def pv_mfmas(jt):
    return [mfma32(O32(jt), VH01, PH01(jt), zero=True), mfma16(O16(jt), VH2, PH2(jt), zero=True),
            mfma32(O32(jt), VH01, PL01(jt)), mfma16(O16(jt), VH2, PL2(jt)),
            mfma32(O32(jt), VL01, PH01(jt)), mfma16(O16(jt), VL2, PH2(jt))]


def osplit(jt, hh):
    """(O32 + O16) / sum -> elements 4 hh .. 4 hh + 3 of OB[jt].h / .l (registers + 2 hh, + 2 hh + 1)."""
    t = V_T + 4 * (jt % 2)
    ops = [f"v_add_f32 v{O32(jt) + r}, v{O32(jt) + r}, v{O16(jt) + r}" for r in range(4)]
    ops += [f"v_mul_f32 v{O32(jt) + r}, v{O32(jt) + r}, v{RS(jt)}" for r in range(4)]
    ops += split_pair(O32(jt), OB(jt, "h") + 2 * hh, OB(jt, "l") + 2 * hh, t)
    return ops


def soft4(jt):
    """64-token build: masked softmax of query tile jt over its 16 accumulator elements (keys 16 mt + 4 g + r) and the split of P,
    unnormalised (1 / sum goes to RS[jt]).  Mask bits: %[m0l] bit 16 (mt % 2) + r for mt 0, 1; %[m0h] the same for mt 2, 3.
    P operands, in place over the 16 score registers: PL01 (+0) PL23 (+4) PH01 (+8) PH23 (+12)."""
    el = [(mt, r) for mt in range(NT) for r in range(4)]
    m, mx, sm = V_T + 8, V_T + 9, V_T + 10
    ops = []
    for i, (mt, r) in enumerate(el):
        word, bit = ("%[m0l]" if mt < 2 else "%[m0h]"), 16 * (mt % 2) + r
        ops += [f"v_bfe_i32 v{m}, {word}, {bit}, 1", f"v_bfi_b32 v{SC(jt, mt) + r}, v{m}, v{SC(jt, mt) + r}, s{S_MASKED}"]
        ops.append(f"v_mov_b32 v{mx}, v{SC(jt, mt) + r}" if i == 0 else f"v_max_f32 v{mx}, v{mx}, v{SC(jt, mt) + r}")
    ops += reduce4("v_max_f32", mx)
    ops.append(f"v_mul_f32 v{mx}, s{S_LOG2E}, v{mx}")
    for mt, r in el:
        x = SC(jt, mt) + r
        ops += [f"v_fma_f32 v{x}, v{x}, s{S_LOG2E}, -v{mx}", f"v_exp_f32 v{x}, v{x}"]
    for i, (mt, r) in enumerate(el):
        x = SC(jt, mt) + r
        ops.append(f"v_mov_b32 v{sm}, v{x}" if i == 0 else f"v_add_f32 v{sm}, v{sm}, v{x}")
    ops += reduce4("v_add_f32", sm)
    ops.append(f"v_rcp_f32 v{RS(jt)}, v{sm}")
    # split, in place: the eight hi packs to temporaries, the sixteen residuals in place, lo packs over registers 0..7, hi packs to 8..15
    t, base = V_T, SC(jt, 0)
    for k in range(8):
        ops.append(f"v_cvt_pk_f16_f32 v{t + k}, v{base + 2 * k}, v{base + 2 * k + 1}")
    for k in range(16):
        sel = "op_sel:[1,0,0] " if k % 2 else ""
        ops.append(f"v_fma_mix_f32 v{base + k}, v{t + k // 2}, -1.0, v{base + k} {sel}op_sel_hi:[1,0,0]")
    for k in range(8):
        ops.append(f"v_cvt_pk_f16_f32 v{PL01(jt) + k}, v{base + 2 * k}, v{base + 2 * k + 1}")
    for k in range(8):
        ops.append(f"v_mov_b32 v{PH01(jt) + k}, v{t + k}")
    return ops


def pv4_mfmas(half):
    """O^T[feature][query] of query tiles 2 half, 2 half + 1: one chain of six K = 32 MFMAs per tile (keys 0-31, 32-63 x three
    terms), the two chains interleaved."""
    chains = []
    for jt in (2 * half, 2 * half + 1):
        chains.append([mfma32(O32(jt), VH01, PH01(jt), zero=True), mfma32(O32(jt), VH23, PH23(jt)),
                       mfma32(O32(jt), VH01, PL01(jt)), mfma32(O32(jt), VH23, PL23(jt)),
                       mfma32(O32(jt), VL01, PH01(jt)), mfma32(O32(jt), VL23, PH23(jt))])
    return [m for pair in zip(*chains) for m in pair]


def osplit4(jt, hh):
    """O / sum -> elements 4 hh .. 4 hh + 3 of OB[jt].h / .l."""
    t = V_T + 4 * (jt % 2)
    ops = [f"v_mul_f32 v{O32(jt) + r}, v{O32(jt) + r}, v{RS(jt)}" for r in range(4)]
    return ops + split_pair(O32(jt), OB(jt, "h") + 2 * hh, OB(jt, "l") + 2 * hh, t)


def head_slot4(hh, L):
    """One head of the 64-token build (see the --nt=4 note at the top)."""
    A = L.append
    tag = f"h{hh}"
    bq4 = [f"v_mul_f32 v{V_BQ + r}, 0.25, v{V_BQ + r}" for r in range(4)]
    # ---- stage V(h) | q, k of this head: scale, bias, split (their biases were read in the K stage before)
    L += stage(v_groups(), bq4 + prep_qk(), f"{tag}v", skip=1)
    for half, nxt in ((0, QA), (1, KA)):
        # ---- S(half) | half 0: v of this head (scale, bias, split); half 1: 1 / sum and split of O for the first two query tiles
        A("s_nop 3")
        L += weave(s_mfmas(half), prep_v() if half == 0 else osplit4(0, hh) + osplit4(1, hh), [], skip=3 if half == 0 else 8, valu_per=2)
        # ---- stage Q(h+1) / K(h+1) | masked softmax + split of P of this half (bare after the last head)
        sv = soft4(2 * half) + soft4(2 * half + 1)
        if hh == 1:
            A(f"s_cmp_eq_u32 s{S_PAIR}, 1")
            A(f"s_cbranch_scc1 .Lh3atd_last{half}_{tag}_%=")
        L += stage(qk_groups(nxt), sv, f"{tag}{'qk'[half]}", skip=3, pre_barrier=bias_reads() if half == 1 else (), valu_per=3)
        if hh == 1:
            A(f"s_branch .Lh3atd_pv{half}_{tag}_%=")
            A(f".Lh3atd_last{half}_{tag}_%=:")
            A("s_nop 7")
            L += sv
            A(f".Lh3atd_pv{half}_{tag}_%=:")
        # ---- P.V of this half (the accumulators are v's: dead since prep_v)
        A("s_nop 3")
        L += pv4_mfmas(half)
    A("s_nop 15")    # (the last P.V MFMAs are eight passes each: their results are read next)
    A("s_nop 15")
    A("s_nop 15")
    L += osplit4(2, hh)
    L += osplit4(3, hh)
    # ---- out_proj k-step of the pair
    if hh == 1:
        A("s_nop 7")
        L += stage(out_groups(0), [], f"{tag}oa")
        L += stage(out_groups(1), [], f"{tag}ob")


def bias_reads():
    """q / k / v bias of the head at V_SLQ / V_SLV (advanced afterwards); q's is pre-multiplied by 1/4 by the caller."""
    return [f"ds_read_b128 {vr(V_BQ)}, v{V_SLQ} offset:{4 * H3D_INB}",
            f"ds_read_b128 {vr(V_BK)}, v{V_SLQ} offset:{4 * (H3D_INB + 128)}",
            f"ds_read_b32 v{V_BV}, v{V_SLV} offset:{4 * (H3D_INB + 256)}",
            f"v_add_u32 v{V_SLQ}, 64, v{V_SLQ}",
            f"v_add_u32 v{V_SLV}, 64, v{V_SLV}"]


def head_slot(hh, L):
    A = L.append
    tag = f"h{hh}"
    # ---- A: stage V(h) | q, k of this head: scale, bias, split.  The head's biases were read in the K stage before.
    bq4 = [f"v_mul_f32 v{V_BQ + r}, 0.25, v{V_BQ + r}" for r in range(4)]
    noprep = "noprep" in EXPERIMENT
    L += stage(v_groups(), [] if noprep else bq4 + prep_qk(), f"{tag}v", skip=1)
    # ---- B: S(h) | v of this head
    A("s_nop 3")
    L += weave([] if "nos" in EXPERIMENT else s_mfmas(), [] if noprep else prep_v(), [], skip=3)
    # ---- C: stages Q(h+1), K(h+1) | softmax + split of P(h)
    sv = [] if "nosoft" in EXPERIMENT else soft(0) + soft(1) + soft(2)   # (timing experiments: results wrong)
    half = len(sv) // 2
    if hh == 1:
        A(f"s_cmp_eq_u32 s{S_PAIR}, 1")
        A(f"s_cbranch_scc1 .Lh3atd_last_{tag}_%=")
    L += stage(qk_groups(QA), sv[:half], f"{tag}q", skip=3, valu_per=3)
    L += stage(qk_groups(KA), sv[half:], f"{tag}k", skip=0, pre_barrier=bias_reads(), valu_per=3)
    if hh == 1:
        A(f"s_branch .Lh3atd_pv_{tag}_%=")
        A(f".Lh3atd_last_{tag}_%=:")
        A("s_nop 7")
        L += sv                      # after the last head there is nothing left to hide it under
        A(f".Lh3atd_pv_{tag}_%=:")
    # ---- D: P.V(h) | 1 / sum and split of O for the tiles already done
    A("s_nop 3")
    if "nopv" not in EXPERIMENT:
        L += pv_mfmas(0) + pv_mfmas(1)
        L += weave(pv_mfmas(2), osplit(0, hh), [], skip=1)
        A("s_nop 7")
        L += osplit(1, hh)
        A("s_nop 7")
        L += osplit(2, hh)
    # ---- E: out_proj k-step of the pair
    if hh == 1:
        A("s_nop 7")
        L += stage(out_groups(0), [], f"{tag}oa")
        L += stage(out_groups(1), [], f"{tag}ob")


def generate():
    L = []
    A = L.append
    A(f"v_mbcnt_lo_u32_b32 v{V_LANE16}, -1, 0")
    A(f"v_mbcnt_hi_u32_b32 v{V_LANE16}, -1, v{V_LANE16}")
    # side-block lane addresses: f4 at feature 4 g (q, k biases), float at feature (lane & 15) (v bias)
    A(f"v_lshrrev_b32 v{V_T}, 4, v{V_LANE16}")
    A(f"v_lshlrev_b32 v{V_T}, 4, v{V_T}")
    A(f"v_add_u32 v{V_SLQ}, {SL}, v{V_T}")
    A(f"v_and_b32 v{V_T}, 15, v{V_LANE16}")
    A(f"v_lshlrev_b32 v{V_T}, 2, v{V_T}")
    A(f"v_add_u32 v{V_SLV}, {SL}, v{V_T}")
    A(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_LANE16}")
    A(f"s_lshl_b32 s{S_W2048}, %[wave], 11")
    A(f"s_mov_b32 s{S_W2048 + 1}, 0")
    A(f"s_mov_b32 s{S_STRIDE}, {STAGE}")
    A(f"s_mov_b32 s{S_STRIDE + 1}, 0")
    A(f"s_mov_b32 s{S_AUXOFF}, {TILES}")
    A(f"s_mov_b32 s{S_AUXOFF + 1}, 0")
    A(f"s_mov_b32 s{S_LOG2E}, 0x3fb8aa3b")
    A(f"s_mov_b32 s{S_MASKED}, 0xc6ea6000")          # -3e4
    A(f"s_add_u32 s{S_END}, %[ring], {RING * STAGE}")
    A(f"v_lshl_add_u64 {vr(V_GN, 2)}, %[gn], 0, s[{S_W2048}:{S_W2048 + 1}]")
    A(f"s_mul_i32 s{S_OFF}, %[cur], {STAGE}")
    A(f"s_add_u32 s{S_OFF}, s{S_OFF}, %[ring]")
    A(f"v_add_u32 v{V_TILE}, s{S_OFF}, v{V_LANE16}")
    # in_proj scale (side block slot 640) -> SGPRs (and a quarter of it for q).  Per-section build: the layer's side block was
    # requested by wave 0 only a few hundred cycles ago and nothing has waited for that DMA yet - the read sits behind the two
    # bare prologue stages below (two hand-offs: vmcnt wait + barrier; the first use of the scale is in the stage after them).
    # Encoder-stack build: the block arrived a whole layer ago (double-buffered), so the read can be the first thing.
    def read_scale():
        A(f"v_mov_b32 v{V_T}, {SL}")
        A(f"ds_read_b32 v{V_T + 1}, v{V_T} offset:{4 * 640}")

    def scale_to_sgprs():
        A(f"v_mul_f32 v{V_T + 2}, 0.25, v{V_T + 1}")
        A("s_nop 1")
        A(f"v_readfirstlane_b32 s{S_SCIN}, v{V_T + 1}")
        A(f"v_readfirstlane_b32 s{S_SCQ}, v{V_T + 2}")

    if FUSED:
        read_scale()
    if not FUSED:
        # split activations from the wave-private block: 24 images -> a96..a191
        A(f"v_add_u32 v{V_TMP}, %[priv], v{V_LANE16}")
        for i in range(8 * NT):
            ks, jt, part = i // (2 * NT), (i // 2) % NT, "hl"[i % 2]
            dst = (vr if XB_CLS(ks, jt) == "v" else ar)(XB(ks, jt, part))
            A(f"ds_read_b128 {dst}, v{V_TMP} offset:{1024 * i}")
        for i in range(32 * NT):
            A(f"v_accvgpr_write_b32 a{i}, 0")
    A("s_waitcnt lgkmcnt(0)")
    if FUSED:
        scale_to_sgprs()
    for r in tile_reads(0) + tile_reads(1):
        A(r)
    # ---- prologue: stages Q(0), K(0) bare; biases of head 0
    L += stage(qk_groups(QA), [], "pq", aux=False)
    if FUSED:
        L += stage(qk_groups(KA), [], "pk", aux=False, pre_barrier=bias_reads())
    else:
        # (the head's biases come from the same side block: read behind both hand-offs as well)
        L += stage(qk_groups(KA), [], "pk", aux=False)
        read_scale()
        L += bias_reads()
        A("s_waitcnt lgkmcnt(0)")   # LDS reads return in order: the next stage's four tile reads (older) land with them
        scale_to_sgprs()
    A(f"s_mov_b32 s{S_PAIR}, 4")
    A(".Lh3atd_pair_%=:")
    (head_slot4 if NT4 else head_slot)(0, L)
    (head_slot4 if NT4 else head_slot)(1, L)
    A(f"s_sub_u32 s{S_PAIR}, s{S_PAIR}, 1")
    A(f"s_cmp_eq_u32 s{S_PAIR}, 0")
    A("s_cbranch_scc0 .Lh3atd_pair_%=")
    # ---- out: ring slot index, y through the wave-private block, DMA pointer
    A(f"s_sub_u32 s{S_REL}, s{S_OFF}, %[ring]")
    A("s_mov_b32 %[cur], 0")
    for k in range(1, RING):
        A(f"s_cmp_eq_u32 s{S_REL}, {k * STAGE}")
        A(f"s_cselect_b32 %[cur], {k}, %[cur]")
    A("s_waitcnt lgkmcnt(0)")
    A("s_nop 15")
    A("s_nop 15")
    if not FUSED:
        A(f"v_add_u32 v{V_TMP}, %[priv], v{V_LANE16}")
        for i in range(8 * NT):
            for r in range(4):
                A(f"v_accvgpr_read_b32 v{V_T + (i % 2) * 4 + r}, a{4 * i + r}")
            A(f"ds_write_b128 v{V_TMP}, {vr(V_T + (i % 2) * 4)} offset:{1024 * i}")
        A("s_waitcnt lgkmcnt(0)")
    A(f"s_sub_u32 s{S_AUXOFF}, 0, s{S_W2048}")
    A(f"s_subb_u32 s{S_AUXOFF + 1}, 0, 0")
    A(f"v_lshl_add_u64 %[gn], {vr(V_GN, 2)}, 0, s[{S_AUXOFF}:{S_AUXOFF + 1}]")
    return L


def main():
    lines = generate()
    out_dir = "timewarp_amd/csrc"
    for a in sys.argv[1:]:
        if a.startswith("--out-dir="):
            out_dir = a.split("=", 1)[1]
    assert not NT4 or FUSED, "--nt=4 exists inside the encoder-stack statement only (tools/gen_h3_enc_asm.py --dense --nt=4)"
    base = os.path.join(out_dir, "tw_h3_attnd_asm.inc")
    out = ["// GENERATED by tools/gen_h3_dense_attn_asm.py - do not edit.  Body of the dense-softmax attention asm statement."]
    out += ['"' + l + '\\n\\t"' for l in lines]
    open(base, "w").write("\n".join(out) + "\n")
    clob = [f'"v{i}"' for i in range(N_V)] + [f'"a{i}"' for i in range(N_A)] + [f'"s{i}"' for i in range(84, 98)] + \
           ['"vcc"', '"scc"', '"memory"']
    cl = ["// GENERATED by tools/gen_h3_dense_attn_asm.py - clobber list of the dense-softmax attention asm statement."]
    for i in range(0, len(clob), 12):
        cl.append(", ".join(clob[i:i + 12]) + ("," if i + 12 < len(clob) else ""))
    open(base.replace("_asm.inc", "_clobbers.inc"), "w").write("\n".join(cl) + "\n")
    print(f"dense attention: {len(lines)} instructions, {sum(1 for l in lines if l.startswith('v_mfma'))} MFMAs")


if __name__ == "__main__":
    main()
