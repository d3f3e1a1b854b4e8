#!/usr/bin/env python3
"""Generate timewarp_amd/csrc/tw_h3_ffn_asm.inc: the hand-scheduled FFN chunk loop of the split-fp16
net-block kernel (gfx950), as the body of one `asm volatile` statement.

Why asm: PMC counters showed the hipcc build of this loop with the matrix pipe 48 % busy - 2 VALU ops
per MFMA issued mostly outside the MFMA bursts, LDS latency and the stage barrier exposed 4x per chunk.
Here every non-MFMA instruction is placed in the shadow of an MFMA (measured capacity per K=32 MFMA:
2 VALU or 3 SALU ops, tools/probe/mfma_valu_overlap.hip), weight tiles are read two tile pairs ahead and
the stage barrier sits in the middle of a stage, so the next stage's first tiles are in flight for 18
MFMAs before they are needed.

Register map (all private to the asm statement, listed as clobbers):
  v0..v71     xb[ks][jt] = {h, l} for k-steps 0..2, a96..a119 for k-step 3: B operand of W1 (split x) from LDS
  v96..v127   weight tile slots p=0..3: hi v[96+8p..+3], lo v[100+8p..+3]
  v128..v175  hb[buf][jt] = {h: v[128+24buf+8jt..+3], l: +4..+7}  split hidden activations (double buffer)
  v176..v183  bias[o][r];  v184 scale;  v186..v193 epilogue temporaries
  v194 tile LDS address, v195 aux address, v196 scale address, v198:199 DMA source, v200:201 temp,
  v202 lane*16, v203 (lane>>4)*16
  a0..a95     yacc[ot][jt] (4 each);  v72..v95 hacc[o][jt] (VGPRs: read by the epilogue directly);  s95 scale
Stage order (must match h3_pack_weights):  A0(0) A1(0) | A0(c+1) A1(c+1) B0(c) B1(c) ... | B0(n-1) B1(n-1)."""
import os
import sys

# experiments: comma-separated flags in H3_FFN_EXPERIMENT.  noepi, nobarrier, nodma: timing only (results become wrong).
# (r01's `pairsync` - one barrier per pair of stages on the five-slot ring, both slots refilled after it - measured 2.5 %
# slower: the refill ran two stages ahead of its use.  --ring6 below is the version with the slack to afford it.)
EXPERIMENT = set(filter(None, os.environ.get("H3_FFN_EXPERIMENT", "").split(",")))
# FFN: the epilogue of chunk c + 1 (96 VALU ops) is spread over the second A stage of chunk c + 1 and both B stages of
# chunk c, in gaps that hold nothing else (`nospread`: the r01 placement - all of it in the B stages, beside the LDS
# reads and the hand-off - measured slower)
SPREAD = "nospread" not in EXPERIMENT
# in / out MLPs (SiLU: 36 VALU ops per 4 values, 216 per chunk): `iotail` lets the units whose accumulators are ready last
# only do their scale/bias fma (which reads hacc and the bias registers) in the B stages of the chunk before and finish
# under the A stage(s) of the NEXT chunk - out: o = 0 units under A1, o = 1 tails under the next A0 instead of everything
# beside the 9 MFMAs of its single B stage; in: three tails under the next A stage
IOTAIL = "iotail" in EXPERIMENT

# Set by tools/gen_h3_enc_asm.py (the FFN embedded in the asm statement of the whole encoder stack): the split input is
# already in the xb registers, the accumulators hold the residual ((x + b2) / scale) and y stays in a0..a95.
FUSED = False
# --h1 (or set by tools/gen_h3_enc_asm.py): the single-MFMA "fast" variant (TW_PATH_FUSED_H1).  Every product is ONE
# v_mfma_f32_16x16x32_f16 on the fp16 hi halves; a stage's four 2 KiB "pairs" then hold TWO hi tiles each (the second in
# the place of the lo tile: SLOT(p, 'l')), so a stage is 8 tiles = 24 MFMAs and a chunk is ONE A stage (tile 4 o + ks)
# and ONE B stage (tile ot).  Ring, DMA and hand-off are unchanged; the lo operand registers stay unused.
H1 = "--h1" in sys.argv
# --nt=4: 64-token waves (one molecule of 49-64 atoms per wave, csrc/tw_netblock_h3.hip H3N4_*): four token tiles, 48 MFMAs
# per stage, a ring of THREE stage buffers (the four 32 KiB wave blocks leave room for no more).  Register map:
#   v0..v63 xb k-steps 0, 1;  a128..a191 xb k-steps 2, 3;  a0..a127 yacc;  v64..v95 hacc;  v96..v127 tile slots;
#   v128..v191 hb (two buffers);  v192..v199 bias;  v200 scale;  v202..v209 temporaries;  v210..v219 addresses
NT4 = "--nt=4" in sys.argv

NT = 4 if NT4 else 3
AHEAD = 3 if NT4 else 5    # stages requested ahead of the one being read (two LDS-DMA pieces per wave each)
# --ring6 (48-token waves only: the LDS has 9 KiB to spare there): a ring of SIX stage buffers and ONE workgroup barrier per
# PAIR of stages.  A hand-off refills the slot of the stage BEFORE the current one (with the same stage of the stream as the
# five-slot ring: five ahead), so a "light" stage - no barrier, no DMA wait - is legal right behind a "heavy" one whose
# barrier (a) saw every wave finish its reads of the stage before it and (b) had every wave wait for its shares of the NEXT
# TWO stages (vmcnt one stage tighter).  The DMA issue stays spread one stage's share per stage; what halves is the number of
# barriers - on the fast mode's 24-MFMA stages the hand-off was 30 % of a stage (tools/probe/h1_stage_probe.hip).
R6 = "--ring6" in sys.argv
assert not (R6 and NT4)
RING = AHEAD + (1 if R6 else 0)
# Shapes of the three chunked MLPs (same schedule, one generated file each):
#   ffn : 128 -> 32-unit chunk (ReLU) -> 128     A = 2 stages (o = 0, 1; 4 k-steps),  B = 2 stages (ot 0-3, 4-7)
#   in  :  64 -> 32-unit chunk (SiLU) -> 128     A = 1 stage  (pairs = o x 2 k-steps), B = 2 stages
#   out : 128 -> 32-unit chunk (SiLU) ->  16     A = 2 stages,                         B = 1 stage with one tile pair
SHAPES = {
    "ffn": dict(ks_in=4, ot_out=8, silu=False, tag="ffn"),
    "in": dict(ks_in=2, ot_out=8, silu=True, tag="in"),
    "out": dict(ks_in=4, ot_out=1, silu=True, tag="out"),
}
SHAPE = SHAPES["ffn"]
# xb[ks][jt]: k-steps 0..2 in VGPRs v0..v71, k-step 3 in AGPRs a96..a119 (MFMA B operands may be AGPRs; this makes
# room for the hidden accumulators in v72..v95 without asking the compiler for more registers)
XB = lambda ks, jt, part: (8 * (3 * ks + jt) if ks < 3 else 96 + 8 * jt) + (0 if part == "h" else 4)
XB_SRC = lambda ks: "v" if ks < 3 else "a"
SLOT = lambda p, part: 96 + 8 * p + (0 if part == "h" else 4)
HB = lambda buf, jt, part: 128 + 24 * buf + 8 * jt + (0 if part == "h" else 4)
BIAS = lambda o: 176 + 4 * o
V_U = 204  # SiLU temporaries v204..v207 (clobbered only by the shapes that use SiLU)
V_SC, V_T, V_TILE, V_AUX, V_SCADDR, V_GN, V_TMP, V_LANE16, V_G16 = 184, 186, 194, 195, 196, 198, 200, 202, 203  # tuples even-aligned
YACC = lambda ot, jt: 4 * (3 * ot + jt)
HACC = lambda o, jt: 72 + 4 * (3 * o + jt)   # VGPRs: the epilogue reads them without a v_accvgpr_read
N_VGPR, N_AGPR = None, 120   # (N_VGPR: by shape, main())
if NT4:
    XB = lambda ks, jt, part: (8 * (4 * ks + jt) if ks < 2 else 128 + 8 * (4 * (ks - 2) + jt)) + (0 if part == "h" else 4)
    XB_SRC = lambda ks: "v" if ks < 2 else "a"
    HB = lambda buf, jt, part: 128 + 32 * buf + 8 * jt + (0 if part == "h" else 4)
    BIAS = lambda o: 192 + 4 * o
    V_SC, V_T, V_TILE, V_AUX, V_SCADDR, V_GN, V_TMP, V_LANE16, V_G16 = 200, 202, 210, 211, 212, 214, 216, 218, 219
    V_U = 220  # SiLU temporaries v220..v223
    YACC = lambda ot, jt: 4 * (4 * ot + jt)
    HACC = lambda o, jt: 64 + 4 * (4 * o + jt)
    N_VGPR, N_AGPR = 224, 192
S_SC = 95
S_ROT = 83    # auxrot: the wave that moves the next bias/scale block (rotates 0..3 so no wave is always the slowest)
# scratch SGPRs (clobbered)
S_OFF, S_REL, S_W2048, S_STRIDE, S_AUXOFF, S_END, S_CNT = 84, 85, 86, 88, 90, 92, 94  # pairs are even-aligned
STAGE, TILES = 9216, 8192


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(base, n=4):
    return f"a[{base}:{base + n - 1}]"


def mfma(d, a, b, zero=False, dst="a", bsrc="v", seed=None):
    """`seed`: VGPR quad the chain starts from instead of zero (the folded bias of the fast mode's FFN)."""
    dd = ar(d) if dst == "a" else vr(d)
    bb = ar(b) if bsrc == "a" else vr(b)
    return f"v_mfma_f32_16x16x32_f16 {dd}, {vr(a)}, {bb}, {(vr(seed) if seed is not None else '0') if zero else dd}"


def fold():
    """Fast mode: the first-layer weights of every chunked MLP are packed UNSCALED (no lo half to keep normal:
    h3_pack_weights), so the hidden pre-activation is the accumulator itself once the chain starts from the bias (the MFMA's
    C operand) - the FFN's epilogue of a unit is two packs and the ReLU on the packed pairs, 4 VALU ops instead of 8; the
    SiLU shapes drop their four scale / bias fmas.  The deletion matrix of the six-slot-ring
    build put 13 % of the FFN into those 8 (profiles/r05_h1_ffn_deletion_matrix.txt).  The bias quads are read by the
    hand-off in front of an A stage (nothing else reads them any more)."""
    assert not (H1 and IOTAIL)
    return H1


def pair_mfmas(acc_of_jt, slot, bop_of_jt, first=False, dst="a", bsrc="v"):
    """9 MFMAs of one tile pair: hi x B.h, hi x B.l, lo x B.h for the three token tiles."""
    out = []
    for a_part, b_part, z in (("h", "h", first), ("h", "l", False), ("l", "h", False)):
        for jt in range(NT):
            out.append(mfma(acc_of_jt(jt), SLOT(slot, a_part), bop_of_jt(jt, b_part), zero=z, dst=dst, bsrc=bsrc))
    return out


V_T3 = 208  # third set of epilogue temporaries (iotail: three units' scale/bias results wait for the next A stage)


def epi_unit(o, jt, buf, relu=True, tset=None):
    """hacc[o][jt] -> dwords 2o, 2o+1 of hb[buf].h[jt] / .l[jt]: 16 VALU ops (the hidden pre-activations
    accumulate in VGPRs and the scale sits in an SGPR: every VALU op costs register-file cycles the MFMAs need)."""
    if tset is None:
        tset = (o * NT + jt) % 2
    t = [(V_T, V_T + 4, V_T3)[tset] + i for i in range(4)]
    # timing experiments (results WRONG): `episrc` reads the bias registers instead of the MFMA results, `epidst` writes
    # scratch registers instead of the B operands of the second GEMM - is it the op count or the MFMA <-> VALU register traffic?
    src = (lambda r: BIAS(o) + r) if "episrc" in EXPERIMENT else (lambda r: HACC(o, jt) + r)
    ops = [f"v_fma_f32 v{t[r]}, v{src(r)}, s{S_SC}, v{BIAS(o) + r}" for r in range(4)]
    pre = t      # registers holding the pre-activation
    if fold():   # ... it is the accumulator itself (bias in the chain's start value, first-layer weights unscaled)
        ops, pre = [], [src(r) for r in range(4)]
    nofma = "epinofma" in EXPERIMENT and not SHAPE["silu"] and not fold()
    if nofma:    # timing experiment (results WRONG): the FFN epilogue without its four scale / bias fmas - what an epilogue of
        ops = []  # 12 instead of 16 ops per unit could gain at most (profiles/r06_headline_experiments.txt)
    if SHAPE["silu"]:
        # v * 1 / (1 + 2^(-v log2 e)) with the hardware exp2 / rcp
        u = [V_U + r for r in range(4)]
        ops += [f"v_mul_f32 v{u[r]}, 0xbfb8aa3b, v{pre[r]}" for r in range(4)]
        ops += [f"v_exp_f32 v{u[r]}, v{u[r]}" for r in range(4)]
        ops += [f"v_add_f32 v{u[r]}, 1.0, v{u[r]}" for r in range(4)]
        ops += [f"v_rcp_f32 v{u[r]}, v{u[r]}" for r in range(4)]
        ops += [f"v_mul_f32 v{t[r]}, v{pre[r]}, v{u[r]}" for r in range(4)]
    else:
        ops += [f"v_max_f32 v{t[r]}, v{src(r) if nofma else t[r]}, 0" for r in range(4)]
    hh = HB(buf, jt, "h") + 2 * o
    ll = HB(buf, jt, "l") + 2 * o
    if H1:
        # no lo half; ReLU after the rounding (max(rne(v), 0) == rne(max(v, 0))) on the packed pair: 8 ops per unit ...
        if not SHAPE["silu"]:
            ops, t = ops[:-4], pre          # ... 4 with fold()
        ops += [f"v_cvt_pk_f16_f32 v{hh}, v{t[0]}, v{t[1]}", f"v_cvt_pk_f16_f32 v{hh + 1}, v{t[2]}, v{t[3]}"]
        if not SHAPE["silu"]:
            ops += [f"v_pk_max_f16 v{hh}, v{hh}, 0", f"v_pk_max_f16 v{hh + 1}, v{hh + 1}, 0"]
        return ops
    if "epidst" in EXPERIMENT:
        hh, ll = V_T3, V_T3 + 2
    ops += [f"v_cvt_pk_f16_f32 v{hh}, v{t[0]}, v{t[1]}", f"v_cvt_pk_f16_f32 v{hh + 1}, v{t[2]}, v{t[3]}"]
    for r in range(4):
        sel = "op_sel:[1,0,0] " if r % 2 else ""
        ops.append(f"v_fma_mix_f32 v{t[r]}, v{hh + r // 2}, -1.0, v{t[r]} {sel}op_sel_hi:[1,0,0]")
    ops += [f"v_cvt_pk_f16_f32 v{ll}, v{t[0]}, v{t[1]}", f"v_cvt_pk_f16_f32 v{ll + 1}, v{t[2]}, v{t[3]}"]
    return ops


def tile_reads(pair):
    return [f"ds_read_b128 {vr(SLOT(pair, 'h'))}, v{V_TILE} offset:{2048 * pair}",
            f"ds_read_b128 {vr(SLOT(pair, 'l'))}, v{V_TILE} offset:{2048 * pair + 1024}"]


def aux_reads():
    return [f"ds_read_b128 {vr(BIAS(0))}, v{V_AUX} offset:{TILES}",
            f"ds_read_b128 {vr(BIAS(1))}, v{V_AUX} offset:{TILES + 64}"] + \
           ([] if fold() else [f"ds_read_b32 v{V_SC}, v{V_SCADDR} offset:{TILES + 128}"])


def next_stage_reads(next_bias):
    r = []
    if next_bias:   # fold(): the next stage is an A stage - its bias quads seed its first MFMAs (in front of the tiles:
        r += [f"v_add_u32 v{V_AUX}, s{S_OFF}, v{V_G16}"] + aux_reads()   # the stage's entry wait counts from the end)
    return r + [f"v_add_u32 v{V_TILE}, s{S_OFF}, v{V_LANE16}"] + tile_reads(0) + tile_reads(1)


def handoff(next_reads, with_aux, label, next_bias=False):
    """Everything after the stage barrier: advance the ring, start reading the next stage, refill the
    released slot by LDS-DMA.  A list of items to be woven between MFMAs in order; an item that is itself
    a list is atomic (the wave-0-only branch).  `with_aux`: the stage being fetched (5 ahead) may carry a
    bias/scale block (only A0 stages do), which wave 0 moves with a third DMA.
    The second KiB of a wave's share uses the instruction offset, which LDS-DMA applies to the global
    address AND to the LDS address (M0 + offset + 16*lane)."""
    h = [
        f"s_mov_b32 s{S_REL}, s{S_OFF}",                       # slot being released = ring base + cur*STAGE
        f"s_add_u32 s{S_OFF}, s{S_OFF}, {STAGE}",
        [f"s_cmp_eq_u32 s{S_OFF}, s{S_END}", f"s_cselect_b32 s{S_OFF}, %[ring], s{S_OFF}"],
    ]
    if R6:
        # six-slot ring: s{S_REL} is persistent = the slot of the stage before this one, which is what gets refilled
        h = [f"s_add_u32 m0, s{S_REL}, s{S_W2048}"] + h
    if next_reads:
        h += next_stage_reads(next_bias)
    h += ([] if R6 else [f"s_add_u32 m0, s{S_REL}, s{S_W2048}", "s_nop 0"]) + [
        f"global_load_lds_dwordx4 {vr(V_GN, 2)}, off",
        f"global_load_lds_dwordx4 {vr(V_GN, 2)}, off offset:1024",
    ]
    if "nodma" in EXPERIMENT:
        # timing experiment: no weight DMA after the kernel's prologue - the ring keeps serving the first five stages
        # (finite, random-looking operands: the clock sees the same switching activity), every vmcnt wait is satisfied at
        # once; results are WRONG.  What remains is issue + LDS + barriers: the difference to `base` is what the stream costs.
        h = [x for x in h if not (isinstance(x, str) and x.startswith("global_load_lds"))]
        with_aux = False
    if with_aux:
        who = f"s{S_ROT}" if "auxrot" in EXPERIMENT else "0"
        if "auxrot" in EXPERIMENT:
            h += [f"s_add_u32 s{S_ROT}, s{S_ROT}, 1", f"s_and_b32 s{S_ROT}, s{S_ROT}, 3"]
        tail_only = [f"s_cmp_gt_u32 s{S_CNT}, 2", f"s_cbranch_scc1 .Lh3mlp_noaux_{label}_%="] if with_aux == "tail" else []
        h += [[f"s_cmp_lg_u32 %[wave], {who}",
               f"s_cbranch_scc1 .Lh3mlp_noaux_{label}_%="] + tail_only + [
               f"v_lshl_add_u64 {vr(V_TMP, 2)}, {vr(V_GN, 2)}, 0, s[{S_AUXOFF}:{S_AUXOFF + 1}]",
               # (six-slot ring: m0 still holds the refilled slot + this wave's share offset, and this is wave 0: offset 0)
               f"s_add_u32 m0, m0, {TILES}" if R6 else f"s_add_u32 m0, s{S_REL}, {TILES}",
               "s_nop 0",
               f"global_load_lds_dwordx4 {vr(V_TMP, 2)}, off",
               f".Lh3mlp_noaux_{label}_%=:"]]
    h += [f"v_lshl_add_u64 {vr(V_GN, 2)}, {vr(V_GN, 2)}, 0, s[{S_STRIDE}:{S_STRIDE + 1}]"]
    return h


def sync(light, before_light):
    """The middle of a stage.  Heavy: every read of this slot has returned, this wave's shares of the next stage (six-slot ring,
    in front of a light stage: of the next two) have landed, workgroup barrier.  Light (six-slot ring only): the stage before
    did all of that for this one as well."""
    if light:
        assert R6
        return ["s_waitcnt lgkmcnt(0)"]
    out = [f"s_waitcnt vmcnt({2 * (AHEAD - (3 if before_light else 2))}) lgkmcnt(0)"]
    if "nobarrier" not in EXPERIMENT:
        out.append("s_barrier")
    return out


def weave(mfmas, valu, misc, valu_per=2, misc_per=2, skip=0):
    """Each MFMA (after the first `skip`) is followed by up to `valu_per` VALU ops and `misc_per` other items, spread so
    the queues empty by the last MFMA (leftovers are appended)."""
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


def place_valu(lines, valu, cap_empty=2, skip_gaps=0):
    """`lines`: a stage's instructions with the MFMAs and everything that has a fixed place (LDS reads, waits, the
    barrier, the hand-off).  Distribute `valu` (order kept) over the gaps behind the MFMAs, preferring gaps that hold
    nothing else: an MFMA issues every 16 cycles and the wave issues about one instruction per 4, so a gap takes two or
    three fillers for free and every further one delays the next MFMA."""
    if not valu:
        return lines
    gaps = []  # (index of the MFMA line, weight of the other items behind it)
    idx = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
    # scalar instructions issue beside a VALU op for free (tools/probe/mfma_valu_overlap.hip: three s_mov per MFMA cost
    # nothing); vector, LDS and DMA instructions each take the slot a VALU op would
    heavy = lambda l: not (l.endswith(":") or l.startswith("s_")) or l.startswith("s_waitcnt") or l.startswith("s_barrier")
    for k, i in enumerate(idx):
        end = idx[k + 1] if k + 1 < len(idx) else len(lines)
        gaps.append((i, 99 if k < skip_gaps else sum(1 for l in lines[i + 1:end] if heavy(l))))
    cap = cap_empty
    while True:
        caps = [0 if n == 99 else max(0, cap - n) for _, n in gaps]
        if sum(caps) >= len(valu):
            break
        cap += 1
    total, placed, out, pos, cum = sum(caps), 0, [], 0, 0
    take = {}
    for (i, _), c in zip(gaps, caps):
        cum += c
        want = -(-len(valu) * cum // total)  # ceil: fill early rather than late
        take[i] = min(c, want - placed)
        placed += take[i]
    assert placed == len(valu), (placed, len(valu))
    for i, l in enumerate(lines):
        out.append(l)
        if i in take:
            out.extend(valu[pos:pos + take[i]])
            pos += take[i]
    return out


def stage(kind, o_or_b, hb_cur, epi_ops, next_reads, with_aux, label, is_a0, light=False, before_light=False, next_is_a=False):
    """One 4-pair stage.  kind 'A': hacc[o] += W1tile . xb[ks] (ks_in = 4: stage = one o, pairs = k-steps; ks_in = 2:
    the single A stage holds both o, pair = 2 o + ks);  kind 'B': yacc[4b+p] += W2tile(p) . hb (pairs beyond ot_out
    do not exist: no MFMAs, no tile reads)."""
    ks_in, ot_out = SHAPE["ks_in"], SHAPE["ot_out"]
    if H1:
        return stage_h1(kind, hb_cur, epi_ops, next_reads, with_aux, label, light, before_light, next_is_a)
    groups = []
    for p in range(4):
        if kind == "A":
            o, ks = (o_or_b, p) if ks_in == 4 else (p // 2, p % 2)
            groups.append(pair_mfmas(lambda jt: HACC(o, jt), p, lambda jt, part: XB(ks, jt, part), first=(ks == 0), dst="v", bsrc=XB_SRC(ks)))
        else:
            b = o_or_b
            if 4 * b + p < ot_out:
                groups.append(pair_mfmas(lambda jt: YACC(4 * b + p, jt), p, lambda jt, part: HB(hb_cur, jt, part)))
            else:
                groups.append([])
    live = [bool(g) for g in groups]
    aux = 3 if is_a0 else 0
    epi = [] if "noepi" in EXPERIMENT else list(epi_ops)
    spread = SPREAD and (SHAPE["tag"] == "ffn" or "spreadio" in EXPERIMENT or IOTAIL)
    share = -(-len(epi) // 4)
    parts = [[] for _ in range(4)] if spread else [epi[i * share:(i + 1) * share] for i in range(4)]
    out = []
    # outstanding LDS reads at entry: p0.hi p0.lo p1.hi p1.lo (+3 aux)
    out.append("s_waitcnt lgkmcnt(2)")
    first_misc = tile_reads(2) if live[2] else []
    if is_a0:
        # bias / scale of this chunk: read here (not at the previous hand-off) so the previous chunk's epilogue,
        # still running in the stage before, keeps its values
        first_misc = [f"v_add_u32 v{V_AUX}, s{S_OFF}, v{V_G16}", f"v_mov_b32 v{V_SCADDR}, s{S_OFF}"] + aux_reads() + first_misc
    out += weave(groups[0], parts[0], first_misc, misc_per=3)
    out.append(f"s_waitcnt lgkmcnt({2 + aux})" if aux else "s_waitcnt lgkmcnt(2)")
    out += weave(groups[1], parts[1], tile_reads(3) if live[3] else [])
    out += sync(light, before_light)   # all my reads of this slot returned; next stage's DMA share landed
    h = handoff(next_reads, with_aux, label)
    if is_a0:
        h = [f"v_readfirstlane_b32 s{S_SC}, v{V_SC}"] + h   # aux block landed (lgkmcnt(0) above)
    out += weave(groups[2], parts[2], h, misc_per=3)
    out += weave(groups[3], parts[3], [])
    if spread:
        out = place_valu(out, epi)
    return out


def stage_h1(kind, hb_cur, epi_ops, next_reads, with_aux, label, light=False, before_light=False, next_is_a=False):
    """One 8-tile stage of the single-MFMA variant.  kind 'A' (always carries the chunk's bias / scale block):
    ks_in = 4: pair p = tiles (o = p // 2, ks = 2 (p % 2) + j), so hacc[0] is complete after pair 1 and its epilogue units
    (`epi_ops`) run under pairs 2, 3; ks_in = 2: pairs 0, 1 = tiles (o = p, ks = j), pairs 2, 3 do not exist.
    kind 'B': pair p = tiles ot = 2 p + j (beyond ot_out: none); `epi_ops` spread over the whole stage."""
    ks_in, ot_out = SHAPE["ks_in"], SHAPE["ot_out"]
    groups = []
    for p in range(4):
        g = []
        for j, part in enumerate(("h", "l")):
            if kind == "A":
                if ks_in == 4:
                    o, ks = p // 2, 2 * (p % 2) + j
                elif p < 2:
                    o, ks = p, j
                else:
                    continue
                g += [mfma(HACC(o, jt), SLOT(p, part), XB(ks, jt, "h"), zero=(ks == 0), dst="v", bsrc=XB_SRC(ks),
                           seed=BIAS(o) if fold() else None) for jt in range(NT)]
            else:
                ot = 2 * p + j
                if ot < ot_out:
                    g += [mfma(YACC(ot, jt), SLOT(p, part), HB(hb_cur, jt, "h")) for jt in range(NT)]
        groups.append(g)
    live = [bool(g) for g in groups]
    is_a0 = kind == "A" and not fold()   # (fold(): the bias quads came with the hand-off before this stage, no scale)
    aux = 3 if is_a0 else 0
    epi = [] if "noepi" in EXPERIMENT else list(epi_ops)
    first = ["s_waitcnt lgkmcnt(2)"]
    first_misc = tile_reads(2) if live[2] else []
    if is_a0:
        first_misc = [f"v_add_u32 v{V_AUX}, s{S_OFF}, v{V_G16}", f"v_mov_b32 v{V_SCADDR}, s{S_OFF}"] + aux_reads() + first_misc
    first += weave(groups[0], [], first_misc, misc_per=3)
    first.append(f"s_waitcnt lgkmcnt({(2 if live[2] else 0) + aux})")
    first += weave(groups[1], [], tile_reads(3) if live[3] else [])
    first += sync(light, before_light)   # all my reads of this slot returned; next stage's DMA share landed
    # fold(), an A stage in front of an A stage (the prologue's): this stage's second half still starts chains from the bias
    # quads the next stage's reads overwrite - those reads (and the tile reads behind them) wait for the last pair
    late = fold() and next_is_a and kind == "A" and next_reads
    h = handoff(next_reads and not late, with_aux, label, next_bias=fold() and next_is_a)
    if is_a0:
        h = [f"v_readfirstlane_b32 s{S_SC}, v{V_SC}"] + h   # aux block landed (lgkmcnt(0) above)
    second = weave(groups[2], [], h, misc_per=3)
    second += weave(groups[3], [], next_stage_reads(True) if late else [], skip=NT if late else 0)
    if kind == "A" and ks_in == 4:
        # epilogue of hacc[0] (complete after pair 1) under pairs 2, 3; two MFMAs of distance to its last writer
        return first + place_valu(second, epi, skip_gaps=2)
    return place_valu(first + second, epi, skip_gaps=2)


def generate():
    L = []
    A = L.append
    # ---- setup
    A(f"v_mbcnt_lo_u32_b32 v{V_LANE16}, -1, 0")
    A(f"v_mbcnt_hi_u32_b32 v{V_LANE16}, -1, v{V_LANE16}")
    A(f"v_lshrrev_b32 v{V_G16}, 4, v{V_LANE16}")
    A(f"v_lshlrev_b32 v{V_G16}, 4, v{V_G16}")
    A(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_LANE16}")
    A(f"s_lshl_b32 s{S_W2048}, %[wave], 11")
    A(f"s_mov_b32 s{S_W2048 + 1}, 0")
    A(f"s_mov_b32 s{S_STRIDE}, {STAGE}")
    A(f"s_mov_b32 s{S_STRIDE + 1}, 0")
    if "auxrot" in EXPERIMENT:
        # the aux block sits at stage + TILES whoever moves it: take this wave's 2 KiB offset out of its DMA source
        A(f"s_sub_u32 s{S_AUXOFF}, {TILES}, s{S_W2048}")
    else:
        A(f"s_mov_b32 s{S_AUXOFF}, {TILES}")
    A(f"s_mov_b32 s{S_AUXOFF + 1}, 0")
    A(f"s_add_u32 s{S_END}, %[ring], {RING * STAGE}")
    if "auxrot" in EXPERIMENT:
        A(f"s_mov_b32 s{S_ROT}, 0")
    # DMA source of this wave = gnext + wave*2048
    A(f"v_lshl_add_u64 {vr(V_GN, 2)}, %[gn], 0, s[{S_W2048}:{S_W2048 + 1}]")
    A(f"s_mul_i32 s{S_OFF}, %[cur], {STAGE}")
    A(f"s_add_u32 s{S_OFF}, s{S_OFF}, %[ring]")
    if R6:   # the slot before the current one (the last slot of the ring in front of slot 0)
        assert "auxrot" not in EXPERIMENT
        A("s_cmp_eq_u32 %[cur], 0")
        A(f"s_cselect_b32 s{S_REL}, s{S_END}, s{S_OFF}")
        A(f"s_sub_u32 s{S_REL}, s{S_REL}, {STAGE}")
    A(f"v_add_u32 v{V_TILE}, s{S_OFF}, v{V_LANE16}")
    ks_in, ot_out = SHAPE["ks_in"], SHAPE["ot_out"]
    ffn = SHAPE["tag"] == "ffn"
    n_a, n_b = (2 if ks_in == 4 else 1), (ot_out + 3) // 4
    if H1:
        n_a = n_b = 1
    # input operand (split activations) from the wave-private LDS block: 2 images per (ks, jt)
    if not FUSED:
        A(f"v_add_u32 v{V_TMP}, %[priv], v{V_LANE16}")
        for i in range(2 * NT * ks_in):
            if H1 and i % 2:
                continue   # lo images: not used
            if NT4:
                dst = vr(4 * i) if i < 16 else ar(128 + 4 * (i - 16))
            else:
                dst = vr(4 * i) if i < 18 else ar(96 + 4 * (i - 18))
            A(f"ds_read_b128 {dst}, v{V_TMP} offset:{1024 * i}")
        for i in range(4 * NT * ot_out):
            A(f"v_accvgpr_write_b32 a{i}, 0")
        A("s_waitcnt lgkmcnt(0)")
    # first stage's reads
    if fold():
        A(f"v_add_u32 v{V_AUX}, s{S_OFF}, v{V_G16}")
        for r in aux_reads():
            A(r)
    for r in tile_reads(0) + tile_reads(1):
        A(r)

    def sync_mode(kind, idx, where):
        """(light, before_light) of a stage on the six-slot ring (FFN only: the short in / out MLPs keep a barrier per stage).
        Fast mode, A(c+1) B(c) per trip: A light, B heavy - the prologue's A(0) and the tail's B(n-1) heavy.  Split form, A0 A1 B0
        B1 per chunk: the second stage of each pair is light.  A statement starts and may end either way: heavy stages have
        no precondition, and nothing outside this loop is ever light."""
        if not (R6 and ffn):
            return dict(light=False, before_light=False)
        if H1:
            if "r6swap" in EXPERIMENT:   # A heavy, B light
                return dict(light=(kind == "B" and where == "loop"), before_light=(kind == "A" and where == "loop"))
            return dict(light=(kind == "A" and where == "loop"), before_light=(kind == "A" or where == "loop"))
        return dict(light=(idx == 1), before_light=(idx == 0))

    def a_stages(buf, tag, aux_of, epi_of=None, where="loop"):
        out = []
        for o in range(n_a):
            out += stage("A", o, buf, epi_of[o] if epi_of else [], True, aux_of("A", o), f"{tag}a{o}", o == 0,
                         next_is_a=(where == "prologue"), **sync_mode("A", o, where))
        return out

    def b_stages(buf, tag, epi, aux_of, last=False):
        out = []
        per = -(-len(epi) // n_b)
        for b_ in range(n_b):
            out += stage("B", b_, buf, epi[b_ * per:(b_ + 1) * per], not (last and b_ == n_b - 1), aux_of("B", b_),
                         f"{tag}b{b_}", False, next_is_a=not last, **sync_mode("B", b_, "tail" if last else "loop"))
        return out

    always = lambda kind, idx: True
    # FFN steady state: the stage fetched by a hand-off is 5 ahead = the kind after this one; only A0 stages carry a
    # bias/scale block, so B1 fetches one.  A1 keeps the aux DMA as well: in the last trips its hand-off fetches the
    # first stage AFTER the FFN (an A0 of out_mlp in the last layer).  The short in/out MLPs always move it.
    steady = (lambda kind, idx: (kind, idx) in (("A", 1), ("B", 1))) if ffn else always
    if H1:
        # two stages per chunk: a B stage's hand-off (five ahead) fetches an A stage, which carries a bias / scale block;
        # an A stage's fetches a B stage - except in the last two trips (chunk counter <= 2), where it is a stage BEHIND this
        # MLP (the out-MLP's first A stage after the last layer): wave 0 moves the block there only, instead of being the
        # straggler of every barrier
        steady = (lambda kind, idx: True if kind == "B" else "tail") if "auxalways" not in EXPERIMENT else always

    iotail = IOTAIL and SPREAD and not ffn
    UNITS = [(o, jt) for o in range(2) for jt in range(NT)]

    def io_units(buf):
        return [epi_unit(o, jt, buf, tset=k % 3) for k, (o, jt) in enumerate(UNITS)]

    def io_tail(buf):  # what the last three units still have to do after their scale/bias fma
        u_ = io_units(buf)
        return u_[3][4:] + u_[4][4:] + u_[5][4:]

    # ---- prologue: A(0), epilogue of chunk 0 (not hidden)
    L += a_stages(0, "p", always, where="prologue")
    A("s_nop 7")
    spread4 = ffn and SPREAD and "spread4" in EXPERIMENT
    if iotail:
        u_ = io_units(0)
        L += u_[0] + u_[1] + u_[2] + u_[3][:4] + u_[4][:4] + u_[5][:4]   # the three tails: first A stage of the loop / exit code
    else:
        for o in range(2):
            for jt in range(NT):
                # spread4: the last unit's tail is left to the next A0 stage (or to the code in front of the final B stages)
                L += epi_unit(o, jt, 0)[:4] if spread4 and (o, jt) == (1, NT - 1) else epi_unit(o, jt, 0)
    A(f"s_sub_u32 s{S_CNT}, %[chunks], 1")
    A(f"s_cmp_eq_u32 s{S_CNT}, 0")
    A("s_cbranch_scc1 .Lh3mlp_tail_%=")
    # ---- steady state, two chunks per loop trip so the hb double buffer alternates statically
    #      trip: A(c+1) B(c)[epi c+1 -> buf1]  A(c+2) B(c+1)[epi c+2 -> buf0]
    A(".Lh3mlp_loop_%=:")
    for half, (cur_buf, nxt_buf) in enumerate(((0, 1), (1, 0))):
        units = [epi_unit(o, jt, nxt_buf) for o in range(2) for jt in range(NT)]
        if H1 and ks_in == 4:
            # hacc[0] is complete after the first half of the A stage: its three units under the second half, hacc[1]'s
            # under the B stage of the chunk before (stage_h1)
            a_epi = {0: [op for u_ in units[:NT] for op in u_]}
            epi = [op for u_ in units[NT:] for op in u_]
        elif H1:
            a_epi = None
            epi = [op for u_ in units for op in u_]
        elif ffn and SPREAD:
            # hacc[0] is complete after stage A0: two of its three units run under A1, the rest under B0 and B1
            # (B0 starts with the last hacc[0] unit, so hacc[1] - finished by A1's last MFMAs - is read well after it)
            n_a1 = NT - 1   # units of hacc[0] under A1
            a_epi = {0: [], 1: [op for u_ in units[:n_a1] for op in u_]}
            epi = [op for u_ in units[n_a1:] for op in u_]
            if "spread4" in EXPERIMENT:
                # fourth window: the tail of the last unit (everything behind its bias/scale fma) runs under stage A0 of
                # the NEXT chunk - hacc[1] is not rewritten before A1, the bias registers are only re-read by A0
                stream = [op for u_ in units for op in u_]
                a_epi = {0: epi_unit(1, NT - 1, cur_buf)[4:], 1: stream[:28]}   # A0: the tail left over from the chunk before
                epi = stream[28:84]
        elif iotail:
            u_ = io_units(nxt_buf)
            if n_a == 2:   # out: hacc[0] is ready after A0 -> its units under A1; o = 1: fma under B, tails under the next A0
                a_epi = {0: io_tail(cur_buf), 1: u_[0] + u_[1] + u_[2]}
                epi = u_[3][:4] + u_[4][:4] + u_[5][:4]
            else:          # in: one A stage holds both o
                a_epi = {0: io_tail(cur_buf)}
                epi = u_[0] + u_[1] + u_[2] + u_[3][:4] + u_[4][:4] + u_[5][:4]
        else:
            a_epi = None
            epi = [op for u_ in units for op in u_]
        L += a_stages(cur_buf, f"l{half}", steady, a_epi)
        L += b_stages(cur_buf, f"l{half}", epi, steady)
        A(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
        A(f"s_cmp_eq_u32 s{S_CNT}, 0")
        if half == 0:
            A("s_cbranch_scc1 .Lh3mlp_tail1_%=")
        else:
            A("s_cbranch_scc0 .Lh3mlp_loop_%=")
    # ---- tails: last chunk's B stages (hb in buf0 after an even number of loop halves, buf1 after odd)
    A(".Lh3mlp_tail_%=:")
    if iotail:
        L += io_tail(0) + ["s_nop 1"]
    if spread4:
        L += epi_unit(1, NT - 1, 0)[4:] + ["s_nop 1"]
    L += b_stages(0, "t0", [], always, last=True)
    A("s_branch .Lh3mlp_done_%=")
    A(".Lh3mlp_tail1_%=:")
    if iotail:
        L += io_tail(1) + ["s_nop 1"]
    if spread4:
        L += epi_unit(1, NT - 1, 1)[4:] + ["s_nop 1"]
    L += b_stages(1, "t1", [], always, last=True)
    A(".Lh3mlp_done_%=:")
    # ring slot index back to the caller: cur = (S_OFF - ring) / STAGE, 0..4
    A(f"s_sub_u32 s{S_REL}, s{S_OFF}, %[ring]")
    A("s_mov_b32 %[cur], 0")
    for k in range(1, RING):
        A(f"s_cmp_eq_u32 s{S_REL}, {k * STAGE}")
        A(f"s_cselect_b32 %[cur], {k}, %[cur]")
    # ---- y out through the wave-private LDS block
    A("s_nop 15")
    A("s_nop 15")
    if not FUSED:
        A(f"v_add_u32 v{V_TMP}, %[priv], v{V_LANE16}")
        for i in range(NT * ot_out):
            for r in range(4):
                A(f"v_accvgpr_read_b32 v{V_T + (i % 2) * 4 + r}, a{4 * i + r}")
            A(f"ds_write_b128 v{V_TMP}, {vr(V_T + (i % 2) * 4)} offset:{1024 * i}")
        A("s_waitcnt lgkmcnt(0)")
    # gnext back (without the wave term)
    A(f"s_sub_u32 s{S_AUXOFF}, 0, s{S_W2048}")
    A(f"s_subb_u32 s{S_AUXOFF + 1}, 0, 0")
    A(f"v_lshl_add_u64 %[gn], {vr(V_GN, 2)}, 0, s[{S_AUXOFF}:{S_AUXOFF + 1}]")
    return L


def main():
    global SHAPE
    shape, out_dir = "ffn", "timewarp_amd/csrc"
    for a in sys.argv[1:]:
        if a.startswith("--shape="):
            shape = a.split("=", 1)[1]
        if a.startswith("--out-dir="):
            out_dir = a.split("=", 1)[1]
    SHAPE = SHAPES[shape]
    lines = generate()
    fam, flag = ("h1n4", " --h1 --nt=4") if H1 and NT4 else ("h1", " --h1") if H1 else ("h3n4", " --nt=4") if NT4 else ("h3", "")
    if R6:
        fam, flag = fam + "r", flag + " --ring6"
    base = os.path.join(out_dir, f"tw_{fam}_{SHAPE['tag']}_asm.inc")
    out = [f"// GENERATED by tools/gen_h3_ffn_asm.py --shape={shape}{flag} - do not edit.  Body of the {shape} MLP asm statement",
           "// (see the generator for the register map and the schedule)."]
    for l in lines:
        out.append('"' + l + '\\n\\t"')
    open(base, "w").write("\n".join(out) + "\n")
    n_v = (212 if IOTAIL else 208) if SHAPE["silu"] else 204
    if NT4:
        n_v = N_VGPR
    clob = [f'"v{i}"' for i in range(n_v)] + [f'"a{i}"' for i in range(N_AGPR)] + [f'"s{i}"' for i in range(83 if "auxrot" in EXPERIMENT else 84, 96)] + \
           ['"vcc"', '"scc"', '"memory"']
    cl = [f"// GENERATED by tools/gen_h3_ffn_asm.py --shape={shape}{flag} - clobber list of the {shape} MLP asm statement."]
    for i in range(0, len(clob), 12):
        cl.append(", ".join(clob[i:i + 12]) + ("," if i + 12 < len(clob) else ""))
    open(base.replace("_asm.inc", "_clobbers.inc"), "w").write("\n".join(cl) + "\n")
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    print(f"{shape}: {len(lines)} instructions, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
