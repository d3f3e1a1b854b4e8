#!/usr/bin/env python3
"""Generate timewarp_amd/csrc/tw_h3_attn_asm.inc: the hand-scheduled kernel-attention block of one encoder
layer of the split-fp16 net-block kernel (gfx950):   y = sum_h Wc_h . (A_h X)^T   for 48 tokens per wave.

Per head h and k-step ks (32 features):
  mixing   xm[ks] = split( X^T[32 ks .. +31][tokens] . A_h[tokens][queries] )      36 MFMAs (18 K=32 + 18 K=16,
           separate accumulators - mixed-shape chains need wait states hipcc does not know about), operands:
           A = rows of the transposed fp16 hi/lo copy of X in the wave-private LDS block, B = score fragments
  GEMM     y[ot] += Wc_h[ot][ks] . xm[ks]       two weight stages (ot 0-3, 4-7) of 36 MFMAs each
Software pipeline (one step = one ks):  the mixing MFMAs of the NEXT k-step run first, then the two GEMM stages
of the current k-step with the fp32 -> fp16 hi/lo split of the new mixing result (72 VALU ops) issued in the
shadow of their MFMAs.  The score fragments of head h+1 are loaded (9 global loads, windowed 7) right after the last mixing of
head h and waited for one k-step later.  Stage hand-off as in the FFN block (gen_h3_ffn_asm.py).

Register map (private to the asm statement):
  v0..v35     score fragments  SF[jt] = {s0h 4, s0l 4, s1h 2, s1l 2}
  v36..v59    X^T operands, two buffers (t = 0, 1) of {a0h 4, a1h 2, a0l 4, a1l 2}
  v60..v83    acc[t][jt] (one chain per accumulator, K=32 and K=16 MFMAs alternating five MFMAs apart)
  v108..v155  xm[buf][jt] = {h 4, l 4}
  v156..v187  weight tile slots p=0..3: hi v[156+8p..], lo v[160+8p..]
  v188..v195  temporaries;  v196 tile address; v197/v198 X^T image addresses (16 B / 8 B per lane)
  v200:201 DMA source; v202:203 temp; v204 lane*16; v206:207 / v208:209 score-fragment addresses (16 B / 8 B lanes)
  v210:211 temp
  a0..a95     y[ot][jt]"""
import os
import sys

# --nt=4: 64-token waves (one molecule of 49-64 atoms per wave; csrc H3N4_*, gen_h3_ffn_asm.py --nt=4): keys = two K = 32 groups
# (one accumulation chain of six same-shape MFMAs per tile: no K = 16 tail, no second accumulator), 48 mixing MFMAs per k-step,
# 48 MFMAs per GEMM stage, a ring of three stage buffers.  Register map:
#   v0..v63 score fragments SF[jt] = {s0h, s0l, s1h, s1l} x 4;  v64..v95 X^T operands, two buffers of {a0h, a1h, a0l, a1l} x 4;
#   v96..v127 acc[t][jt];  v128..v191 xm[buf][jt] = {h 4, l 4};  v192..v223 weight tile slots;  v224..v231 temporaries;
#   v232.. addresses;  a0..a127 y[ot][jt]
NT4 = "--nt=4" in sys.argv
# --nt=4 --pair (embedded in the encoder-stack statement only, gen_h3_enc_asm.py): ONE molecule of 97-128 atoms per PAIR of 64-token
# waves, two molecules per workgroup (r04 ran these on the wide layout's 48-token waves: one molecule on three of four waves,
# 52-67 % of the token slots).  A wave's queries are its own 64 tokens, its keys the molecule's 128: four K = 32 groups - the two
# of its own X^T images and the two of its partner's, which are the lane-order images the 64-token statement already writes
# (conflict-free as they are, no shared row-major tile, no padding: 4 x 32 KiB of images + the three-slot ring + the side block =
# 158 KiB).  Per chain twelve same-shape MFMAs; the mixing of a k-step runs in four phases (feature tile t, key half) over the two
# 16-register operand buffers, the second half's operands read while the other tile's phase runs.  Score fragments: groups 0, 1
# in v0..v63 as before, groups 2, 3 in a128..a191 (MFMA B operands may be AGPRs) - 8 KiB per (head, query tile).
PAIR = "--pair" in sys.argv
assert not PAIR or NT4
NT = 4 if NT4 else 3
AHEAD = 3 if NT4 else 5    # stages requested ahead of the one being read
# --ring6: six stage buffers, a hand-off refills the slot of the stage BEFORE the current one (gen_h3_ffn_asm.py R6; every stage
# of this block keeps its barrier - the FFN is where stages pair up)
R6 = "--ring6" in sys.argv
assert not (R6 and NT4)
RING = AHEAD + (1 if R6 else 0)
STAGE, TILES = 9216, 8192
# Transposed copy of x in the wave-private block (csrc: "x -> transposed", H3_XT_IMG): per feature tile ft and part
# (hi, lo) one 1536-byte image = [T0 | T1] 16 B per lane (K = 32 operand) + T2 8 B per lane (K = 16 operand); every lane
# reads its own bytes (conflict-free).  r03 before: rows of 48 halfs per feature (24-dword stride; 28 in r01 / r02 was 2-way).
XT_IMG = 1536
SF_BYTES = 3072
SF = lambda jt, name: 12 * jt + {"s0h": 0, "s0l": 4, "s1h": 8, "s1l": 10}[name]
# a1 (token tile 2) directly behind a0 (tiles 0 | 1): registers +2..+5 of a part are the (T1 | T2) operand of windowed tile 2
XA = lambda buf, name: 36 + 12 * buf + {"a0h": 0, "a1h": 4, "a0l": 6, "a1l": 10}[name]
ACC = lambda t, jt: 60 + 4 * (3 * t + jt)
TAIL = lambda t, jt: 84 + 4 * (3 * t + jt)
XM = lambda buf, jt, part: 108 + 24 * buf + 8 * jt + (0 if part == "h" else 4)
SLOT = lambda p, part: 156 + 8 * p + (0 if part == "h" else 4)
V_T, V_TILE, V_XT0, V_XT1, V_GN, V_TMP, V_LANE16, V_SF16, V_SF8, V_TMP2 = 188, 196, 197, 198, 200, 202, 204, 206, 208, 210
YACC = lambda ot, jt: 4 * (3 * ot + jt)
S_OFF, S_REL, S_W2048, S_STRIDE, S_AUXOFF, S_END, S_CNT, S_K3072, S_K6144 = 84, 85, 86, 88, 90, 92, 93, 94, 96
N_V, N_A = 212, 96
N_S = 98
if NT4:
    XT_IMG, SF_BYTES = 2048, 4096
    SF = lambda jt, name: 16 * jt + {"s0h": 0, "s0l": 4, "s1h": 8, "s1l": 12}[name]
    XA = lambda buf, name: 64 + 16 * buf + {"a0h": 0, "a1h": 4, "a0l": 8, "a1l": 12}[name]
    ACC = lambda t, jt: 96 + 4 * (4 * t + jt)
    XM = lambda buf, jt, part: 128 + 32 * buf + 8 * jt + (0 if part == "h" else 4)
    SLOT = lambda p, part: 192 + 8 * p + (0 if part == "h" else 4)
    V_T, V_TILE, V_XT0, V_XT1, V_GN, V_TMP, V_LANE16, V_SF16, V_SF8, V_TMP2 = 224, 232, 233, 233, 234, 236, 238, 240, 242, 244
    YACC = lambda ot, jt: 4 * (4 * ot + jt)
    S_K3 = 98        # 3 * SF_BYTES; s82:83 = NT * SF_BYTES (fragment stride per head) - not s100:101, which hipcc reserves
    S_SFHEAD = 82
    N_V, N_A, N_S = 246, 128, 100
    if PAIR:
        SF_BYTES = 8192
        PAIR_HALF = 32768           # bytes from a wave's X^T images to its partner's (csrc H3N4_WAVE_LDS)
        SF_AGPR = 128               # groups 2, 3
        N_A = 192
EXPERIMENT = set(filter(None, os.environ.get("H3_ATTN_EXPERIMENT", "").split(",")))
# --mode=windowed: waves that hold two or more molecules.  The score matrix is block diagonal, so query tile 0 only has
# keys in [0, 32) and query tile 2 only in [16, 48): each takes ONE K=32 mixing MFMA per term (tile 2 with the
# (T1 | T2) registers of the operand buffer) and only tile 1 keeps the K=16 tail - 24 instead of 36 mixing
# MFMAs per k-step (the K=16 shape costs a full slot).  The fragment producer lays tile 2's K=32 block over keys 16..47
# for this mode (csrc: h3_score_frag_kernel `windowed`).  Single-molecule waves keep the full 48 keys (--mode=full).
WINDOWED = "--mode=windowed" in sys.argv
# Set by tools/gen_h3_enc_asm.py, which embeds this block in the asm statement of the whole encoder stack: the accumulators
# arrive holding the residual (x / scale) instead of zeros, y stays in a0..a95 (no trip through the LDS), and the score
# fragments of the layer are addressed through SF_BASE (a VGPR pair that statement advances per layer).
FUSED = False
# --h1 (or set by tools/gen_h3_enc_asm.py): the single-MFMA "fast" variant (TW_PATH_FUSED_H1, see gen_h3_ffn_asm.py).  One MFMA
# per product on the fp16 hi halves: 8 (windowed) / 12 mixing MFMAs per k-step, ONE weight stage of 8 hi tiles per k-step
# (pair p = output tiles 2 p, 2 p + 1, the second in the place of the lo tile), the split of a mixing result is a pack.
H1 = "--h1" in sys.argv
SF_BASE = "%[sf]"
# FUSED only: the transposed copy of x does not go through the LDS at all - the statement's transposer leaves it in AGPRs
# (MFMA A operands may be AGPRs), 12 per feature tile: [T0 | T1] hi 4, T2 hi 2, [T0 | T1] lo 4, T2 lo 2
XT_AGPR = None   # first AGPR (gen_h3_enc_asm.py: 96), or None: operand buffers v36..v59 filled by ds_reads
A2 = lambda t, part: XA(t, "a0h" if part == "h" else "a0l") + 2


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(base, n=4):
    return f"a[{base}:{base + n - 1}]"


def mfma32(d, a, b, zero=False, dreg="v", areg="v", breg="v"):
    dd = vr(d) if dreg == "v" else ar(d)
    aa = vr(a) if areg == "v" else ar(a)
    bb = vr(b) if breg == "v" else ar(b)
    return f"v_mfma_f32_16x16x32_f16 {dd}, {aa}, {bb}, {'0' if zero else dd}"


def mfma16(d, a, b, zero=False, areg="v"):
    aa = vr(a, 2) if areg == "v" else ar(a, 2)
    return f"v_mfma_f32_16x16x16_f16 {vr(d)}, {aa}, {vr(b, 2)}, {'0' if zero else vr(d)}"


XT_OFF = {"a0h": 0, "a1h": 4, "a0l": 6, "a1l": 10}


def xt_operand(ks, t, name):
    """(register, file) of a mixing A operand of feature tile 2 ks + t."""
    if XT_AGPR is not None:
        return XT_AGPR + 12 * (2 * ks + t) + XT_OFF[name], "a"
    return XA(t, name), "v"


def xt_reads(ks, t, buf, half=0):
    off = 2 * XT_IMG * (2 * ks + t)
    if PAIR:
        off += PAIR_HALF * half     # V_XT0 is the molecule's first wave's block: half 1 = the second wave's images
    if NT4:   # four 16-byte operands per (feature tile, part pair): [T0 | T1], [T2 | T3], hi and lo
        r = [f"ds_read_b128 {vr(XA(buf, 'a0h'))}, v{V_XT0} offset:{off}",
             f"ds_read_b128 {vr(XA(buf, 'a1h'))}, v{V_XT0} offset:{off + 1024}",
             f"ds_read_b128 {vr(XA(buf, 'a0l'))}, v{V_XT0} offset:{off + XT_IMG}",
             f"ds_read_b128 {vr(XA(buf, 'a1l'))}, v{V_XT0} offset:{off + XT_IMG + 1024}"]
        return r[:2] if H1 else r
    r = [f"ds_read_b128 {vr(XA(buf, 'a0h'))}, v{V_XT0} offset:{off}",
         f"ds_read_b128 {vr(XA(buf, 'a0l'))}, v{V_XT0} offset:{off + XT_IMG}",
         f"ds_read_b64 {vr(XA(buf, 'a1h'), 2)}, v{V_XT1} offset:{off + 1024}",
         f"ds_read_b64 {vr(XA(buf, 'a1l'), 2)}, v{V_XT1} offset:{off + XT_IMG + 1024}"]
    return r[0::2] if H1 else r


def mixing_mfmas(ks):
    """36 MFMAs of one k-step: six accumulators acc[t][jt], each a chain  K32 hh -> K16 hh -> K32 hl -> K16 hl ->
    K32 lh -> K16 lh.  Alternating the two MFMA shapes on ONE accumulator needs >= 5 wait states between them
    (tools/probe/mfma_chain_probe.hip); the six chains are issued round-robin, so two MFMAs of one chain are always
    five MFMAs (>= 40 cycles) apart."""
    out = []
    first = True
    terms = (("a0h", "s0h", "a1h", "s1h"), ("a0h", "s0l", "a1h", "s1l"), ("a0l", "s0h", "a1l", "s1h"))
    if PAIR:
        raise AssertionError("pair mode: mixing_part() issues the four phases itself")
    if NT4:
        # eight chains acc[t][jt] of six K = 32 MFMAs, issued round-robin: two MFMAs of a chain are eight apart
        for a32, b32, a16, b16 in (terms[:1] if H1 else terms):
            for a, b in ((a32, b32), (a16, b16)):
                for t in range(2):
                    for jt in range(NT):
                        reg, file = xt_operand(ks, t, a)
                        out.append(mfma32(ACC(t, jt), reg, SF(jt, b), zero=first, areg=file))
                first = False
        return out
    # H1: the hi x hi term only.  Windowed: a chain's K = 16 member follows its K = 32 member with >= 2 other MFMAs between
    for a32, b32, a16, b16 in (terms[:1] if H1 else terms):
        order = [(t, jt) for t in range(2) for jt in range(NT)]
        if H1 and WINDOWED:
            # the two chains that get a K = 16 member first: five MFMAs between the two shapes on one accumulator, as in the
            # three-term schedule (tools/probe/mfma_chain_probe.hip)
            order = [(0, 1), (1, 1), (0, 0), (0, 2), (1, 0), (1, 2)]
        for t, jt in order:
            reg, file = xt_operand(ks, t, a32)
            if WINDOWED and jt == 2:
                reg += 2   # the (T1 | T2) registers
            out.append(mfma32(ACC(t, jt), reg, SF(jt, b32), zero=first, areg=file))
        for t in range(2):
            for jt in ((1,) if WINDOWED else range(NT)):
                # windowed: the chain of (t, 1) still has four other MFMAs between its K=32 and its K=16 member
                reg, file = xt_operand(ks, t, a16)
                out.append(mfma16(ACC(t, jt), reg, SF(jt, b16), areg=file))
        first = False
    return out


def xt_reads_step(ks):
    return [] if "noxt" in EXPERIMENT or XT_AGPR is not None else xt_reads(ks, 0, 0) + xt_reads(ks, 1, 1)


def pair_phase(t, half):
    """Pair mode: the 24 (H1: 8) MFMAs of feature tile t against key groups 2 half, 2 half + 1 - four chains acc[t][jt], issued
    round-robin (two MFMAs of a chain are four apart: 64 issue cycles against ~40 of latency)."""
    out = []
    terms = (("a0h", "s0h", "a1h", "s1h"), ("a0h", "s0l", "a1h", "s1l"), ("a0l", "s0h", "a1l", "s1h"))
    first = half == 0
    for a32, b32, a16, b16 in (terms[:1] if H1 else terms):
        for a, b in ((a32, b32), (a16, b16)):
            for jt in range(NT):
                breg, base = ("v", 0) if half == 0 else ("a", SF_AGPR)
                out.append(mfma32(ACC(t, jt), XA(t, a), base + SF(jt, b), zero=first, breg=breg))
            first = False
    return out


def mixing_part(ks, reads_issued):
    """The 36 mixing MFMAs of k-step ks; the eight X^T operand reads are either issued here or were woven into
    the previous GEMM stage."""
    out = [] if reads_issued else xt_reads_step(ks)
    out.append("s_waitcnt lgkmcnt(0)")
    if "nomix" in EXPERIMENT:
        return out
    if PAIR:
        # four phases over the two operand buffers (buffer t = feature tile t): the second key half of a tile is read into its
        # buffer once every MFMA of that tile's first half has been issued AND the other tile's phase has started - a full
        # phase (>= 130 issue cycles) lies between the last reader's issue and the overwrite, another before the first use
        out += pair_phase(0, 0)
        out += weave(pair_phase(1, 0), [], xt_reads(ks, 0, 0, half=1), misc_per=1, skip=2)
        out.append("s_waitcnt lgkmcnt(0)")
        out += weave(pair_phase(0, 1), [], xt_reads(ks, 1, 1, half=1), misc_per=1, skip=2)
        out.append("s_waitcnt lgkmcnt(0)")
        out += pair_phase(1, 1)
        return out
    out += mixing_mfmas(ks)
    return out


def split_ops(buf):
    """acc -> xm[buf]: per (jt, t) 8 VALU ops (2+2 packs, 4 mixed-precision subtractions)."""
    ops = []
    k = 0
    for jt in range(NT):
        for t in range(2):
            tt = [V_T + 4 * (k % 2) + r for r in range(4)]
            k += 1
            a = ACC(t, jt)
            hh = XM(buf, jt, "h") + 2 * t
            ll = XM(buf, jt, "l") + 2 * t
            ops += [f"v_cvt_pk_f16_f32 v{hh}, v{a}, v{a + 1}", f"v_cvt_pk_f16_f32 v{hh + 1}, v{a + 2}, v{a + 3}"]
            if H1:
                continue
            for r in range(4):
                sel = "op_sel:[1,0,0] " if r % 2 else ""
                ops.append(f"v_fma_mix_f32 v{tt[r]}, v{hh + r // 2}, -1.0, v{a + r} {sel}op_sel_hi:[1,0,0]")
            ops += [f"v_cvt_pk_f16_f32 v{ll}, v{tt[0]}, v{tt[1]}", f"v_cvt_pk_f16_f32 v{ll + 1}, v{tt[2]}, v{tt[3]}"]
    return ops


def tile_reads(pair):
    return [f"ds_read_b128 {vr(SLOT(pair, 'h'))}, v{V_TILE} offset:{2048 * pair}",
            f"ds_read_b128 {vr(SLOT(pair, 'l'))}, v{V_TILE} offset:{2048 * pair + 1024}"]


def handoff(next_reads, label, aux_cnt=1):
    """`aux_cnt`: wave 0 also moves the 1 KiB bias / scale block while the head counter is <= aux_cnt (see below)."""
    h = [
        f"s_mov_b32 s{S_REL}, s{S_OFF}",
        f"s_add_u32 s{S_OFF}, s{S_OFF}, {STAGE}",
        [f"s_cmp_eq_u32 s{S_OFF}, s{S_END}", f"s_cselect_b32 s{S_OFF}, %[ring], s{S_OFF}"],
    ]
    if R6:   # s{S_REL} is persistent: the slot of the stage before this one is the one refilled
        h = [f"s_add_u32 m0, s{S_REL}, s{S_W2048}"] + h
    if next_reads:
        h += [f"v_add_u32 v{V_TILE}, s{S_OFF}, v{V_LANE16}"] + tile_reads(0) + tile_reads(1)
    h += ([] if R6 else [f"s_add_u32 m0, s{S_REL}, s{S_W2048}", "s_nop 0"]) + [
        f"global_load_lds_dwordx4 {vr(V_GN, 2)}, off",
        f"global_load_lds_dwordx4 {vr(V_GN, 2)}, off offset:1024",
        # Only the FFN's A0 stages carry a bias/scale block; the hand-offs that fetch them (five ahead) all belong to the
        # layer's LAST head, so wave 0 moves the aux block there only and is not the straggler at every barrier.
        ["s_cmp_lg_u32 %[wave], 0",
         f"s_cbranch_scc1 .Lh3att_noaux_{label}_%=",
         f"s_cmp_lg_u32 s{S_CNT}, 1" if aux_cnt == 1 else f"s_cmp_gt_u32 s{S_CNT}, {aux_cnt}",
         f"s_cbranch_scc1 .Lh3att_noaux_{label}_%=",
         f"v_lshl_add_u64 {vr(V_TMP, 2)}, {vr(V_GN, 2)}, 0, s[{S_AUXOFF}:{S_AUXOFF + 1}]",
         f"s_add_u32 m0, m0, {TILES}" if R6 else f"s_add_u32 m0, s{S_REL}, {TILES}",   # (R6: wave 0's share offset is 0)
         "s_nop 0",
         f"global_load_lds_dwordx4 {vr(V_TMP, 2)}, off",
         f".Lh3att_noaux_{label}_%=:"],
        f"v_lshl_add_u64 {vr(V_GN, 2)}, {vr(V_GN, 2)}, 0, s[{S_STRIDE}:{S_STRIDE + 1}]",
    ]
    if "nodma" in EXPERIMENT:  # timing experiment, see gen_h3_ffn_asm.py: no weight DMA after the prologue, results WRONG
        h = [x for x in h if isinstance(x, str) and not x.startswith("global_load_lds")]
    return h


def weave(mfmas, valu, misc, valu_per=1, misc_per=3, skip=0):
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


def gemm_stage(half, xm_buf, valu, next_reads, label, vm_allow=6, skip=0, tail_misc=(), aux_cnt=1):
    """One Wc stage: y[4 half + p] += tile pair p . xm[xm_buf], with `valu` woven under the MFMAs.
    H1: the stage holds all eight output tiles of the k-step (`half` unused): y[2 p + j] += tile j of pair p . xm.h."""
    groups = []
    for p in range(4):
        g = []
        if H1:
            for j, part in enumerate(("h", "l")):
                for jt in range(NT):
                    g.append(mfma32(YACC(2 * p + j, jt), SLOT(p, part), XM(xm_buf, jt, "h"), dreg="a"))
            groups.append(g)
            continue
        for a_part, b_part in (("h", "h"), ("h", "l"), ("l", "h")):
            for jt in range(NT):
                g.append(mfma32(YACC(4 * half + p, jt), SLOT(p, a_part), XM(xm_buf, jt, b_part), dreg="a"))
        groups.append(g)
    valu = [] if "novalu" in EXPERIMENT else list(valu)
    share = -(-len(valu) // 4)
    parts = [valu[i * share:(i + 1) * share] for i in range(4)]
    out = ["s_waitcnt lgkmcnt(2)"]
    out += weave(groups[0], parts[0], tile_reads(2), skip=skip)
    out.append("s_waitcnt lgkmcnt(2)")
    out += weave(groups[1], parts[1], tile_reads(3))
    out.append(f"s_waitcnt vmcnt({vm_allow}) lgkmcnt(0)")
    if "nobarrier" not in EXPERIMENT:
        out.append("s_barrier")
    out += weave(groups[2], parts[2], handoff(next_reads, label, aux_cnt))
    out += weave(groups[3], parts[3], list(tail_misc))
    return out


def sf_tiles():
    """(jt, with_tail): the K = 16 block of a query tile is only loaded where the mixing uses it (windowed: tile 1 only)."""
    return [(jt, not WINDOWED or jt == 1) for jt in range(NT)]


def sf_loads():
    """Score fragments of the head at V_SF16 (per query tile: K = 32 hi, K = 32 lo, [K = 16 hi | lo] - 16 bytes per lane
    each), then the address advances to the next head.  Every load costs ~60 cycles of issue beside the LDS-DMA stream
    (profiles/r03_attn_deletion_matrix_cycles.txt), so nothing is fetched that the mixing does not read."""
    out = []
    if "nosf" in EXPERIMENT:
        return ["s_nop 0"]
    if NT4:
        for jt in range(NT):
            if jt == 0:
                a16 = V_SF16
            else:
                k = (S_K3072, S_K6144, S_K3)[jt - 1]
                out += [f"v_lshl_add_u64 {vr(V_TMP, 2)}, {vr(V_SF16, 2)}, 0, s[{k}:{k + 1}]"]
                a16 = V_TMP
            for i, name in enumerate(("s0h", "s0l", "s1h", "s1l")):
                if H1 and name.endswith("l"):
                    continue
                # (pair mode: the fragment address is biased by + 4096 - the instruction offset is 13 bits, signed)
                off = 1024 * i - (4096 if PAIR else 0)
                out += [f"global_load_dwordx4 {vr(SF(jt, name))}, {vr(a16, 2)}, off" + (f" offset:{off}" if off else "")]
            if PAIR:   # key groups 2, 3 (the molecule's second wave's half) straight into AGPRs
                for i, name in enumerate(("s0h", "s0l", "s1h", "s1l")):
                    if H1 and name.endswith("l"):
                        continue
                    out += [f"global_load_dwordx4 {ar(SF_AGPR + SF(jt, name))}, {vr(a16, 2)}, off" + (f" offset:{1024 * i}" if i else "")]
        out += [f"v_lshl_add_u64 {vr(V_SF16, 2)}, {vr(V_SF16, 2)}, 0, s[{S_SFHEAD}:{S_SFHEAD + 1}]"]
        return out
    for jt, tail in sf_tiles():
        if jt == 0:
            a16 = V_SF16
        else:
            k = S_K3072 if jt == 1 else S_K6144
            out += [f"v_lshl_add_u64 {vr(V_TMP, 2)}, {vr(V_SF16, 2)}, 0, s[{k}:{k + 1}]"]
            a16 = V_TMP
        out += [f"global_load_dwordx4 {vr(SF(jt, 's0h'))}, {vr(a16, 2)}, off"]
        if not H1:
            out += [f"global_load_dwordx4 {vr(SF(jt, 's0l'))}, {vr(a16, 2)}, off offset:1024"]
        if tail:
            out += [f"global_load_dwordx4 {vr(SF(jt, 's1h'))}, {vr(a16, 2)}, off offset:2048"]
    out += [f"v_lshl_add_u64 {vr(V_SF16, 2)}, {vr(V_SF16, 2)}, 0, s[{S_STRIDE}:{S_STRIDE + 1}]"]   # NT * SF_BYTES == STAGE
    return out


def generate():
    assert NT4 or NT * SF_BYTES == STAGE
    L = []
    A = L.append
    A(f"v_mbcnt_lo_u32_b32 v{V_LANE16}, -1, 0")
    A(f"v_mbcnt_hi_u32_b32 v{V_LANE16}, -1, v{V_LANE16}")
    # transposed copy: this lane's 16 B / 8 B of every image
    A(f"v_lshlrev_b32 v{V_T}, 4, v{V_LANE16}")
    A(f"v_lshlrev_b32 v{V_T + 1}, 3, v{V_LANE16}")
    A(f"v_add_u32 v{V_XT0}, %[priv], v{V_T}")
    if PAIR:   # the molecule's first wave's block: an odd wave steps one block back
        A(f"s_and_b32 s{S_OFF}, %[wave], 1")
        A(f"s_mul_i32 s{S_OFF}, s{S_OFF}, {PAIR_HALF}")
        A(f"v_subrev_u32 v{V_XT0}, s{S_OFF}, v{V_XT0}")
    if not NT4:   # (64-token build: every operand is 16 bytes per lane)
        A(f"v_add_u32 v{V_XT1}, %[priv], v{V_T + 1}")
    # score-fragment lane addresses: sf + 16 lane (128-bit loads), sf + 8 lane (64-bit loads)
    A(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_LANE16}")
    A(f"v_mov_b32 v{V_LANE16 + 1}, 0")
    A(f"v_lshl_add_u64 {vr(V_SF16, 2)}, {SF_BASE}, 0, {vr(V_LANE16, 2)}")
    if PAIR:
        A(f"s_mov_b32 s{S_OFF}, 4096")
        A(f"s_mov_b32 s{S_REL}, 0")
        A(f"v_lshl_add_u64 {vr(V_SF16, 2)}, {vr(V_SF16, 2)}, 0, s[{S_OFF}:{S_REL}]")
    A(f"s_lshl_b32 s{S_W2048}, %[wave], 11")
    A(f"s_mov_b32 s{S_W2048 + 1}, 0")
    A(f"s_mov_b32 s{S_STRIDE}, {STAGE}")
    A(f"s_mov_b32 s{S_STRIDE + 1}, 0")
    A(f"s_mov_b32 s{S_AUXOFF}, {TILES}")
    A(f"s_mov_b32 s{S_AUXOFF + 1}, 0")
    A(f"s_mov_b32 s{S_K3072}, {SF_BYTES}")
    A(f"s_mov_b32 s{S_K3072 + 1}, 0")
    A(f"s_mov_b32 s{S_K6144}, {2 * SF_BYTES}")
    A(f"s_mov_b32 s{S_K6144 + 1}, 0")
    if NT4:
        A(f"s_mov_b32 s{S_K3}, {3 * SF_BYTES}")
        A(f"s_mov_b32 s{S_K3 + 1}, 0")
        A(f"s_mov_b32 s{S_SFHEAD}, {NT * SF_BYTES}")
        A(f"s_mov_b32 s{S_SFHEAD + 1}, 0")
    A(f"s_add_u32 s{S_END}, %[ring], {RING * STAGE}")
    A(f"v_lshl_add_u64 {vr(V_GN, 2)}, %[gn], 0, s[{S_W2048}:{S_W2048 + 1}]")
    A(f"s_mul_i32 s{S_OFF}, %[cur], {STAGE}")
    A(f"s_add_u32 s{S_OFF}, s{S_OFF}, %[ring]")
    if R6:   # the slot before the current one
        A("s_cmp_eq_u32 %[cur], 0")
        A(f"s_cselect_b32 s{S_REL}, s{S_END}, s{S_OFF}")
        A(f"s_sub_u32 s{S_REL}, s{S_REL}, {STAGE}")
    A(f"v_add_u32 v{V_TILE}, s{S_OFF}, v{V_LANE16}")
    if not FUSED:
        for i in range(N_A):
            A(f"v_accvgpr_write_b32 a{i}, 0")
    # ---- prologue: score fragments of head 0, mixing of (0, 0) and its split in the open
    L += sf_loads()
    A("s_waitcnt vmcnt(0)")
    for r in tile_reads(0) + tile_reads(1):
        A(r)
    L += mixing_part(0, False)
    A("s_nop 7")
    L += split_ops(0)
    A("s_nop 1")
    A(f"s_mov_b32 s{S_CNT}, %[heads]")
    # ---- one loop trip = one head: steps ks = 0..3; step ks runs the mixing of the next k-step, then GEMM(ks)
    A(".Lh3att_head_%=:")
    # the X^T operand reads of a step's mixing are woven into the last MFMAs of the previous step (first step of the
    # first head: issued before the loop)
    L += xt_reads_step(1)
    for ks in range(4):
        buf, nbuf = ks % 2, 1 - ks % 2
        if ks < 3:
            L += mixing_part(ks + 1, True)
            if ks == 2:
                # last use of this head's score fragments is issued: fetch the next head's.  Also after the last
                # head (the buffer has one head of slack): the s_waitcnt vmcnt counts below assume these loads.
                L += sf_loads()
            n_sf = (2 if H1 else 4) * NT * (2 if PAIR else 1) if NT4 else sum((3 if tail else 2) - (1 if H1 else 0) for _, tail in sf_tiles())
            base_vm = 2 * (AHEAD - 2)         # stages s + 2 .. s + AHEAD - 1 may stay in flight, two DMAs each
            vm = n_sf + base_vm if ks == 2 else base_vm   # the fragment loads (full: 9, windowed: 7) sit in the same queue behind the stage DMAs
        else:
            # mixing of (h + 1, 0): needs the new score fragments; skipped after the last head.  Newer than the
            # fragment loads are the DMAs of two hand-offs: 4 for every wave (aux blocks move in the last head only,
            # where this block is skipped).
            A(f"s_cmp_eq_u32 s{S_CNT}, 1")
            A("s_cbranch_scc1 .Lh3att_nomix_%=")
            A("s_waitcnt vmcnt(2)" if H1 else "s_waitcnt vmcnt(4)")   # H1: one hand-off (2 DMAs) is newer than the fragment loads
            L += mixing_part(0, True)
            A(".Lh3att_nomix_%=:")
            vm = 2 * (AHEAD - 2)
        split = split_ops(nbuf)
        nxt = (ks + 2) % 4   # k-step whose mixing runs at the start of the next step
        if H1:
            # the hand-off of a stage fetches the stage five ahead: FFN A stages (which carry a bias / scale block) for the
            # last head's four stages and for the last stage of the head before
            L += gemm_stage(0, buf, split, True, f"k{ks}", vm_allow=vm, skip=3, tail_misc=xt_reads_step(nxt),
                            aux_cnt=2 if ks == 3 else 1)
        else:
            L += gemm_stage(0, buf, split[:8 * NT], True, f"k{ks}a", vm_allow=vm, skip=3)
            L += gemm_stage(1, buf, split[8 * NT:], True, f"k{ks}b", vm_allow=vm, tail_misc=xt_reads_step(nxt))
        A("s_nop 1")
    A(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    A(f"s_cmp_eq_u32 s{S_CNT}, 0")
    A("s_cbranch_scc0 .Lh3att_head_%=")
    # The fragment loads issued in the last head (the "next head" that does not exist) must have LANDED before the statement
    # gives its registers back: a load that returns after the exit writes into whatever the code behind keeps there (r04,
    # tools/stress_wide.py: ~0.5 % of the fast mode's launches on the new wide statements returned a corrupted workgroup - two
    # 24-MFMA stages after the loads are only ~800 cycles).  Newer than those loads are the DMAs of the hand-offs of k-steps 2
    # and 3: two per stage for every wave (wave 0's aux blocks only make the wait stricter).
    A(f"s_waitcnt vmcnt({4 if H1 else 8})")
    # ---- out: ring slot index, y through the wave-private block (X^T is dead now), DMA pointer
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
    base = os.path.join(out_dir, ("tw_h1n4_attn_asm.inc" if H1 else "tw_h3n4_attn_asm.inc") if NT4 else
                        "tw_h3_attnw_asm.inc" if WINDOWED else "tw_h3_attn_asm.inc")
    assert not (NT4 and WINDOWED) and (NT4 or not H1 or FUSED)
    assert not PAIR, "--pair exists inside the encoder-stack statement only (tools/gen_h3_enc_asm.py --nt=4 --pair)"
    out = [f"// GENERATED by tools/gen_h3_attn_asm.py{' --mode=windowed' if WINDOWED else ''}{' --nt=4' if NT4 else ''}{' --h1' if H1 else ''} - do not edit.  Body of the attention asm statement."]
    out += ['"' + l + '\\n\\t"' for l in lines]
    open(base, "w").write("\n".join(out) + "\n")
    clob = [f'"v{i}"' for i in range(N_V)] + [f'"a{i}"' for i in range(N_A)] + [f'"s{i}"' for i in range(82 if NT4 else 84, N_S)] + \
           ['"vcc"', '"scc"', '"memory"']
    cl = [f"// GENERATED by tools/gen_h3_attn_asm.py{' --mode=windowed' if WINDOWED else ''}{' --nt=4' if NT4 else ''}{' --h1' if H1 else ''} - clobber list of the attention asm statement."]
    for i in range(0, len(clob), 12):
        cl.append(", ".join(clob[i:i + 12]) + ("," if i + 12 < len(clob) else ""))
    open(base.replace("_asm.inc", "_clobbers.inc"), "w").write("\n".join(cl) + "\n")
    print(f"{len(lines)} instructions, {sum(1 for l in lines if l.startswith('v_mfma'))} MFMAs")


if __name__ == "__main__":
    main()
