#!/usr/bin/env python3
"""Generate timewarp_amd/csrc/tw_h3_attns_asm.inc: the kernel-attention block of the split-fp16 net-block kernel for
molecules of 49 .. 160 atoms ("wide" layout).  A workgroup's 4 x 48 = 192 token slots hold floor(192 / V) whole molecules
back to back, so a molecule spans waves; the transposed fp16 hi/lo copy of X is ONE tile shared by the workgroup
([feature][192 tokens], row stride 104 dwords: conflict-free for ds_read_b128's lane groups) and every wave mixes its three
query tiles against a window of key tiles that covers the molecules it touches:

    xm[ks] = split( sum over groups gi  X^T[32 ks .. +31][keys of group gi] . A_h[keys of group gi][queries] )

One group = 32 key tokens = one K = 32 MFMA per term; NG = 5 groups (160 keys) starting at the wave's window offset `win`
(the fragment producer writes zeros where a key is outside the query's molecule: h3w_score_frag_kernel).  Per head and
k-step: 90 mixing MFMAs (5 groups x 3 terms x 2 feature tiles x 3 query tiles) + 72 for the folded Wc GEMM; everything else
(software pipeline over k-steps, split under the GEMM stages, stage hand-off, weight ring) is the schedule of
gen_h3_attn_asm.py.  The in / FFN / out sections are token-local and are used unchanged.

Register map (private to the asm statement):
  v0..v31     X^T operands XA[buf][t] = {hi 4, lo 4}: two buffers (groups alternate), t = the two 16-feature tiles of a k-step
  v32..v55    acc[t][jt]
  v56..v103   xm[buf][jt] = {h 4, l 4}
  v104..v135  weight tile slots p = 0..3: hi v[104+8p..], lo v[108+8p..]
  v136..v143  temporaries;  v144 tile address;  v145 / v146 X^T hi / lo lane addresses
  v148:149 DMA source; v150:151 temp; v152:153 lane*16 (64-bit); v154:155 running score-fragment pointer
  a0..a95     y[ot][jt];   a96..a159 + v156..v211  score fragments SF[gi][jt] = {hi 4, lo 4} of the head in flight"""
import os
import sys

NT = 3
NG = 5                      # K = 32 key groups per wave window
# --ng=3: molecules of 65 .. 96 atoms at a slot stride of 96 (two per workgroup, each on its own pair of waves): a wave's
# keys are its molecule's <= 96 slots = three groups - 54 mixing MFMAs per head and k-step instead of 90, 18 fragment
# loads per head instead of 30 (tw_h3_attns3_asm.inc; H3Wide::ng in csrc/tw_netblock_h3.hip)
# --ng=6: ONE molecule of 161 .. 192 atoms over the workgroup's 192 slots - every wave mixes against all twelve key tiles
# (tw_h3_attns6_asm.inc; the 144 fragment registers reach v239, so the compiler keeps less of its own state across this
# statement than across the others - these sizes ran on the per-op path before)
for _a in sys.argv[1:]:
    if _a.startswith("--ng="):
        NG = int(_a.split("=", 1)[1])
assert NG in (3, 5, 6)
STAGE, TILES = 9216, 8192
XT_ROW = 416                # bytes per feature row of the shared transposed tile (104 dwords = 192 tokens + 16 pad)
XT_LO = 128 * XT_ROW        # offset of the lo half
GROUP_BYTES = 64            # 32 key tokens x 2 bytes
FRAG_BLOCK = 2048           # one (group, query tile): hi 1 KiB + lo 1 KiB
XA = lambda buf, t, part: 16 * buf + 8 * t + (0 if part == "h" else 4)
ACC = lambda t, jt: 32 + 4 * (3 * t + jt)
XM = lambda buf, jt, part: 56 + 24 * buf + 8 * jt + (0 if part == "h" else 4)
SLOT = lambda p, part: 104 + 8 * p + (0 if part == "h" else 4)
# score fragments: 120 registers; the first 64 in a96..a159, the rest in v156..v211 - together with v0..v155 / a0..a95 the
# statement then owns v0..v211 and a0..a159, which leaves the compiler v212..v255 + a160..a255 for the fp32 residual it
# carries across ALL asm sections (the others own at most v0..v211 / a0..a119), so nothing has to move between them
def SF(gi, jt, part):
    n = 24 * gi + 8 * jt + (0 if part == "h" else 4)
    return ("a", 96 + n) if n < 64 else ("v", 156 + n - 64)
V_T, V_TILE, V_XTH, V_XTL, V_GN, V_TMP, V_LANE16, V_SF = 136, 144, 145, 146, 148, 150, 152, 154
YACC = lambda ot, jt: 4 * (3 * ot + jt)
S_OFF, S_REL, S_W2048, S_STRIDE, S_AUXOFF, S_END, S_CNT, S_K2048 = 84, 85, 86, 88, 90, 92, 93, 94
N_V, N_A = (240 if NG == 6 else 212), 160
# --h1: the single-MFMA "fast" variant (TW_PATH_FUSED_H1; see gen_h3_ffn_asm.py / gen_h3_attn_asm.py): hi halves only - 30
# mixing MFMAs per k-step, ONE weight stage of eight hi tiles per k-step, a split is a pack, half the fragment loads
H1 = "--h1" in sys.argv
N_SF_LOADS = (1 if H1 else 2) * NT * NG    # global loads per head
EXPERIMENT = set(filter(None, os.environ.get("H3_ATTN_EXPERIMENT", "").split(",")))
# Set by tools/gen_h3_enc_asm.py --wide, which embeds this block in the asm statement of the whole encoder stack: the
# accumulators arrive holding the residual (x / scale) instead of zeros, y stays in a0..a95, and the score fragments of the
# layer are addressed through SF_BASE (an SGPR pair that statement advances per layer).
FUSED = False
SF_BASE = "%[sf]"


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def ar(base, n=4):
    return f"a[{base}:{base + n - 1}]"


def reg(cls_idx, n=4):
    cls, base = cls_idx
    return ar(base, n) if cls == "a" else vr(base, n)


def mfma_mix(d, a, b, zero=False):
    return f"v_mfma_f32_16x16x32_f16 {vr(d)}, {vr(a)}, {reg(b)}, {'0' if zero else vr(d)}"


def mfma_gemm(d, a, b):
    return f"v_mfma_f32_16x16x32_f16 {ar(d)}, {vr(a)}, {vr(b)}, {ar(d)}"


def xt_reads(ks, gi):
    """A operands of group gi of k-step ks: rows 16 (2 ks + t) + (lane & 15), key columns of the group."""
    buf = gi % 2
    out = []
    for t in range(2):
        off = XT_ROW * 16 * (2 * ks + t) + GROUP_BYTES * gi
        out += [f"ds_read_b128 {vr(XA(buf, t, 'h'))}, v{V_XTH} offset:{off}"]
        if not H1:
            out += [f"ds_read_b128 {vr(XA(buf, t, 'l'))}, v{V_XTL} offset:{off}"]
    return out


def group_mfmas(gi):
    """18 MFMAs: hi x hi, hi x lo, lo x hi for (t, jt); the six accumulator chains are issued round-robin."""
    out = []
    buf = gi % 2
    terms = (("h", "h"), ("h", "l"), ("l", "h"))
    for k, (ap, bp) in enumerate(terms[:1] if H1 else terms):
        for t in range(2):
            for jt in range(NT):
                out.append(mfma_mix(ACC(t, jt), XA(buf, t, ap), SF(gi, jt, bp), zero=(gi == 0 and k == 0)))
    return out


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


def mixing_part(ks):
    """The 90 mixing MFMAs of k-step ks.  Group 0's operand reads were issued before (prologue / tail of the previous GEMM
    stage); group gi + 1's are issued behind the first MFMAs of group gi into the other buffer (the MFMAs of group gi - 1,
    its previous readers, have all been issued)."""
    out = []
    for gi in range(NG):
        out.append("s_waitcnt lgkmcnt(0)")
        nxt = xt_reads(ks, gi + 1) if gi + 1 < NG else []
        out += weave(group_mfmas(gi), [], nxt, misc_per=1, skip=1)
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
        # only the FFN's A0 stages carry a bias/scale block; the hand-offs that fetch them (five ahead) belong to the layer's
        # LAST head, so wave 0 moves the aux block there only
        ["s_cmp_lg_u32 %[wave], 0",
         f"s_cbranch_scc1 .Lh3atw_noaux_{label}_%=",
         f"s_cmp_lg_u32 s{S_CNT}, 1" if aux_cnt == 1 else f"s_cmp_gt_u32 s{S_CNT}, {aux_cnt}",
         f"s_cbranch_scc1 .Lh3atw_noaux_{label}_%=",
         f"v_lshl_add_u64 {vr(V_TMP, 2)}, {vr(V_GN, 2)}, 0, s[{S_AUXOFF}:{S_AUXOFF + 1}]",
         f"s_add_u32 m0, s{S_REL}, {TILES}",
         "s_nop 0",
         f"global_load_lds_dwordx4 {vr(V_TMP, 2)}, off",
         f".Lh3atw_noaux_{label}_%=:"],
        f"v_lshl_add_u64 {vr(V_GN, 2)}, {vr(V_GN, 2)}, 0, s[{S_STRIDE}:{S_STRIDE + 1}]",
    ]
    return h


def gemm_stage(half, xm_buf, valu, next_reads, label, vm_allow=6, skip=0, tail_misc=(), aux_cnt=1):
    """One Wc stage: y[4 half + p] += tile pair p . xm[xm_buf], with `valu` woven under the MFMAs.
    H1: all eight output tiles of the k-step in one stage: y[2 p + j] += tile j of pair p . xm.h"""
    groups = []
    for p in range(4):
        g = []
        if H1:
            for j, part in enumerate(("h", "l")):
                for jt in range(NT):
                    g.append(mfma_gemm(YACC(2 * p + j, jt), SLOT(p, part), XM(xm_buf, jt, "h")))
            groups.append(g)
            continue
        for a_part, b_part in (("h", "h"), ("h", "l"), ("l", "h")):
            for jt in range(NT):
                g.append(mfma_gemm(YACC(4 * half + p, jt), SLOT(p, a_part), XM(xm_buf, jt, b_part)))
        groups.append(g)
    valu = list(valu)
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


def sf_loads():
    """The NG x NT fragment blocks of the head at V_SF (hi + lo each), consecutive in memory; the pointer ends at the next
    head."""
    out = []
    for gi in range(NG):
        for jt in range(NT):
            out += [f"global_load_dwordx4 {reg(SF(gi, jt, 'h'))}, {vr(V_SF, 2)}, off"]
            if not H1:
                out += [f"global_load_dwordx4 {reg(SF(gi, jt, 'l'))}, {vr(V_SF, 2)}, off offset:1024"]
            out += [f"v_lshl_add_u64 {vr(V_SF, 2)}, {vr(V_SF, 2)}, 0, s[{S_K2048}:{S_K2048 + 1}]"]
    return out


def generate():
    L = []
    A = L.append
    A(f"v_mbcnt_lo_u32_b32 v{V_LANE16}, -1, 0")
    A(f"v_mbcnt_hi_u32_b32 v{V_LANE16}, -1, v{V_LANE16}")
    # X^T lane addresses: row (lane & 15), 16-byte column group (lane >> 4), the wave's key window
    A(f"v_and_b32 v{V_T}, 15, v{V_LANE16}")
    A(f"v_mul_u32_u24 v{V_T}, {XT_ROW}, v{V_T}")
    A(f"v_lshrrev_b32 v{V_T + 1}, 4, v{V_LANE16}")
    A(f"v_lshlrev_b32 v{V_T + 1}, 4, v{V_T + 1}")
    A(f"v_add3_u32 v{V_XTH}, v{V_T}, v{V_T + 1}, %[xt]")
    A(f"v_add_u32 v{V_XTH}, %[win], v{V_XTH}")
    A(f"v_add_u32 v{V_XTL}, {XT_LO}, v{V_XTH}")
    A(f"v_lshlrev_b32 v{V_LANE16}, 4, v{V_LANE16}")
    A(f"v_mov_b32 v{V_LANE16 + 1}, 0")
    A(f"v_lshl_add_u64 {vr(V_SF, 2)}, {SF_BASE}, 0, {vr(V_LANE16, 2)}")
    A(f"s_lshl_b32 s{S_W2048}, %[wave], 11")
    A(f"s_mov_b32 s{S_W2048 + 1}, 0")
    A(f"s_mov_b32 s{S_STRIDE}, {STAGE}")
    A(f"s_mov_b32 s{S_STRIDE + 1}, 0")
    A(f"s_mov_b32 s{S_AUXOFF}, {TILES}")
    A(f"s_mov_b32 s{S_AUXOFF + 1}, 0")
    A(f"s_mov_b32 s{S_K2048}, {FRAG_BLOCK}")
    A(f"s_mov_b32 s{S_K2048 + 1}, 0")
    A(f"s_add_u32 s{S_END}, %[ring], {5 * STAGE}")
    A(f"v_lshl_add_u64 {vr(V_GN, 2)}, %[gn], 0, s[{S_W2048}:{S_W2048 + 1}]")
    A(f"s_mul_i32 s{S_OFF}, %[cur], {STAGE}")
    A(f"s_add_u32 s{S_OFF}, s{S_OFF}, %[ring]")
    A(f"v_add_u32 v{V_TILE}, s{S_OFF}, v{V_LANE16}")
    if not FUSED:
        for i in range(96):
            A(f"v_accvgpr_write_b32 a{i}, 0")
    # ---- prologue: score fragments of head 0, mixing of (0, 0) and its split in the open
    L += sf_loads()
    A("s_waitcnt vmcnt(0)")
    for r in tile_reads(0) + tile_reads(1):
        A(r)
    L += xt_reads(0, 0)
    L += mixing_part(0)
    A("s_nop 7")
    L += split_ops(0)
    A("s_nop 1")
    A(f"s_mov_b32 s{S_CNT}, %[heads]")
    # ---- one loop trip = one head: steps ks = 0..3; step ks runs the mixing of the next k-step, then GEMM(ks)
    A(".Lh3atw_head_%=:")
    L += xt_reads(1, 0)
    for ks in range(4):
        buf, nbuf = ks % 2, 1 - ks % 2
        if ks < 3:
            L += mixing_part(ks + 1)
            if ks == 2:
                # last use of this head's score fragments is issued: fetch the next head's (also after the last head - the
                # buffer has one head of slack; the s_waitcnt vmcnt counts below assume the loads)
                L += sf_loads()
            vm = 6 + N_SF_LOADS if ks == 2 else 6   # the fragment loads sit in the same queue behind the stage DMAs
        else:
            # mixing of (h + 1, 0): needs the new fragments; skipped after the last head.  Newer than the fragment loads
            # are the DMAs of two hand-offs: 4 for every wave (aux blocks move in the last head only, where this is skipped)
            A(f"s_cmp_eq_u32 s{S_CNT}, 1")
            A("s_cbranch_scc1 .Lh3atw_nomix_%=")
            A("s_waitcnt vmcnt(2)" if H1 else "s_waitcnt vmcnt(4)")   # H1: one hand-off is newer than the fragment loads
            L += mixing_part(0)
            A(".Lh3atw_nomix_%=:")
            vm = 6
        split = split_ops(nbuf)
        nxt = (ks + 2) % 4   # k-step whose mixing runs at the start of the next step
        if H1:
            L += gemm_stage(0, buf, split, True, f"k{ks}", vm_allow=vm, skip=3, tail_misc=xt_reads(nxt, 0),
                            aux_cnt=2 if ks == 3 else 1)
        else:
            L += gemm_stage(0, buf, split[:24], True, f"k{ks}a", vm_allow=vm, skip=3)
            L += gemm_stage(1, buf, split[24:], True, f"k{ks}b", vm_allow=vm, tail_misc=xt_reads(nxt, 0))
        A("s_nop 1")
    A(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    A(f"s_cmp_eq_u32 s{S_CNT}, 0")
    A("s_cbranch_scc0 .Lh3atw_head_%=")
    # the last head's fragment prefetch (of a head that does not exist) must have landed before the statement returns its
    # registers - see gen_h3_attn_asm.py; newer than those loads: two DMAs per hand-off of k-steps 2 and 3
    A(f"s_waitcnt vmcnt({4 if H1 else 8})")
    # ---- out: ring slot index; every wave is done with the shared X^T tile before anyone writes y over it
    A(f"s_sub_u32 s{S_REL}, s{S_OFF}, %[ring]")
    A("s_mov_b32 %[cur], 0")
    for k in range(1, 5):
        A(f"s_cmp_eq_u32 s{S_REL}, {k * STAGE}")
        A(f"s_cselect_b32 %[cur], {k}, %[cur]")
    A("s_waitcnt lgkmcnt(0)")
    A("s_barrier")
    A("s_nop 15")
    A("s_nop 15")
    if not FUSED:
        A(f"v_add_u32 v{V_TMP}, %[priv], v{V_LANE16}")
        for i in range(24):
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
    sfx = str(NG) if NG != 5 else ""
    flags = (f" --ng={NG}" if NG != 5 else "") + (" --h1" if H1 else "")
    base = os.path.join(out_dir, f"tw_h1_attns{sfx}_asm.inc" if H1 else f"tw_h3_attns{sfx}_asm.inc")
    out = [f"// GENERATED by tools/gen_h3_attn_wide_asm.py{flags} - do not edit.  Body of the wide-layout attention asm statement."]
    out += ['"' + l + '\\n\\t"' for l in lines]
    open(base, "w").write("\n".join(out) + "\n")
    clob = [f'"v{i}"' for i in range(N_V)] + [f'"a{i}"' for i in range(N_A)] + [f'"s{i}"' for i in range(84, 96)] + \
           ['"vcc"', '"scc"', '"memory"']
    cl = [f"// GENERATED by tools/gen_h3_attn_wide_asm.py{flags} - clobber list of the wide-layout attention asm statement."]
    for i in range(0, len(clob), 12):
        cl.append(", ".join(clob[i:i + 12]) + ("," if i + 12 < len(clob) else ""))
    open(base.replace("_asm.inc", "_clobbers.inc"), "w").write("\n".join(cl) + "\n")
    print(f"wide{flags}: {len(lines)} instructions, {sum(1 for l in lines if l.startswith('v_mfma'))} MFMAs")


if __name__ == "__main__":
    main()
