// Split-fp16 ("h3") variant of the fused net-block kernel (gfx950 / CDNA4).
//
// Same structure as tw_netblock.hip (transposed formulation, activations chained in registers in
// MFMA D/B layout, one wave = whole molecules, folded kernel attention), but every fp32 product
// a*b is evaluated on the half-precision matrix pipe as
//        a*b  ~=  ah*bh + ah*bl + al*bh        (a = ah + al, ah = fp16(a), al = fp16(a - ah))
// with fp32 accumulation: three v_mfma_f32_16x16x32_f16 per 32-deep k-step instead of eight
// v_mfma_f32_16x16x4_f32, i.e. 5.3x fewer matrix-pipe cycles at a representation error of 2^-22
// per operand (fp32-class; measured end-to-end error vs the reference ~3e-6 relative).
//
// Differences that follow from the 5x shorter compute time per weight byte:
//  * weights (fp16 hi/lo tile pairs, per-matrix power-of-two scaled so the lo halves stay normal)
//    are staged through LDS by LDS-DMA (global_load_lds, 16 B/lane) in 9 KiB stages (4 tile pairs +
//    1 KiB of bias/scale) shared by the 4 waves of a workgroup, in a 5-deep ring, one barrier per stage;
//  * the mixing A_h X runs on the same 3-product scheme from an fp16 hi/lo copy of X stored
//    TRANSPOSED in LDS ([feature][token], row stride 48 halfs -> conflict-free ds_read_b128).
// Waves hold 48 tokens (NT = 3): every molecule of up to 48 atoms, floor(48 / V) molecules per wave; larger ones use
// the f32 kernel.
//
// Toolchain note (ROCm 7.2 hipcc, gfx950): a dependent accumulation chain that alternates
// v_mfma_f32_16x16x32_f16 (K=32) and v_mfma_f32_16x16x16_f16 (K=16) on ONE accumulator is emitted back
// to back, but the hardware needs >= 5 wait states between the two shapes; terms are silently dropped
// (tools/probe/).  The mixing below therefore keeps the K=32 body and the K=16 tail in separate
// accumulators.  Before this was understood it showed up as results that changed with any scheduling
// perturbation of the kernel.
#include <utility>
#include <vector>

#include "tw_common.h"

namespace tw {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));  // raw 128-bit register tuple
typedef unsigned u2 __attribute__((ext_vector_type(2)));

#define H3_NT 3
#define H3_FFN_SPLIT 4   // TW_PATH_SIMPLE_H3, small launches: workgroups per token tile of the FFN launch (h3_ffn_split_pack)
#define H3_TOK (16 * H3_NT)
#define H3_XT 48      // tokens of a wave
#define H3_XT_IMG 1536  // transposed copy: bytes per (feature tile, hi | lo) = [T0 | T1] 1024 + T2 512
// ^ halfs per feature row of the transposed X tile: the 48 tokens, no padding.  Row stride 24 dwords is conflict-free for
//   the ds_read_b128 lane groups of gfx950; 56 halfs (the r01/r02 value) was 2-way (tools/gen_h3_attn_asm.py)
#define H3_PAIR_BYTES 2048          // one (hi, lo) tile pair: 16 out x 32 k
#define H3_STAGE_PAIRS 4
#define H3_STAGE_TILE_BYTES 8192    // 4 pairs
#define H3_STAGE_BYTES 9216         // + 1 KiB aux (bias[32], scale)
#define H3_RING 5                   // stage buffers: 1 being read + 4 in flight (~1 us of LDS-DMA latency)
#define H3_XT_BYTES (128 * H3_XT * 2)  // one of (hi, lo)
#define H3_WAVE_LDS (2 * H3_XT_BYTES)
#define H3_SIDE_LAYER_FLOATS 656      // per-layer side block: n1w n1b b2 n2w n2b [128 each], wc scale, w2 scale, pad
#define H3_SIDE_LDS_OFFSET (H3_RING * H3_STAGE_BYTES + 4 * H3_WAVE_LDS)
#define H3_SIDE_LDS_BYTES 3072        // three 1 KiB LDS-DMA chunks (the block is 2624 B; the rest of LDS up to 160 KiB)
#define H3_LDS_BYTES (H3_SIDE_LDS_OFFSET + H3_SIDE_LDS_BYTES)
// instantiations on the six-slot ring (netblock_h3_kernel R6): + one stage buffer = 153 KiB
// (-DTW_H1_RING5: A/B builds of the five-slot arrangement, tools/ab_ring6.sh)
#ifdef TW_H1_RING5
#define H3_R6(NT, DENSE, WIDE, RFF, ENC, H1) false
#else
#define H3_R6(NT, DENSE, WIDE, RFF, ENC, H1) ((ENC) && (H1) && (NT) == 3 && !(DENSE) && !(WIDE) && !(RFF))
#endif
#define H3_R6_LDS_BYTES (H3_LDS_BYTES + H3_STAGE_BYTES)
// 64-token waves (NT = 4): one molecule of 49 .. 64 atoms per wave, so BASELINE config 3 (60 atoms x 512 proposals) is 128
// workgroups per net = ONE round of the chip instead of the wide layout's 1.34.  The wave-private block grows to 32 KiB (the
// 128 x 64 transposed tile, hi + lo; equally the 32 register images the sections exchange), which leaves room for a ring of
// three stage buffers (the r03 DMA probe: same stage time as five).
#define H3N4_NT 4
#define H3N4_RING 3
#define H3N4_XT_IMG 2048              // transposed copy: bytes per (feature tile, hi | lo) = [T0 | T1] 1024 + [T2 | T3] 1024
#define H3N4_WAVE_LDS (2 * 128 * 64 * 2)
#define H3N4_SF_BYTES 4096            // score fragments per (head, query tile): two K = 32 k-steps x (hi, lo) KiB
#define H3N4_SIDE_LDS_OFFSET (H3N4_RING * H3_STAGE_BYTES + 4 * H3N4_WAVE_LDS)
#define H3N4_LDS_BYTES (H3N4_SIDE_LDS_OFFSET + H3_SIDE_LDS_BYTES)
// dense softmax variant (transformer_nvp): no transposed X tile, so the wave-private block only carries the 24 register
// images the asm sections exchange operands through; the layer's side block grows by the in_proj / out_proj biases
#define H3D_WAVE_LDS (24 * 1024)
#define H3D_SIDE_LAYER_FLOATS 1280    // 656 as above (slot 642: out_proj scale) + in_proj bias [384] at 656 + out_proj bias [128] at 1040
#define H3D_SIDE_LDS_OFFSET (H3_RING * H3_STAGE_BYTES + 4 * H3D_WAVE_LDS)
#define H3D_SIDE_LDS_BYTES 5120       // five 1 KiB LDS-DMA chunks
#define H3D_LDS_BYTES (H3D_SIDE_LDS_OFFSET + H3D_SIDE_LDS_BYTES)
#define H3D_ENC_LDS_BYTES (H3D_LDS_BYTES + H3D_SIDE_LDS_BYTES)  // encoder-stack statement: the side block double-buffered
#define H3D4_LDS_BYTES (H3N4_SIDE_LDS_OFFSET + H3D_SIDE_LDS_BYTES)  // dense model on 64-token waves: three-slot ring + 4 x 32 KiB + 5 KiB = 160 KiB
#define H3D_INB 656
#define H3D_OUTB 1040
// "wide" layout (49 .. 160 atoms): a workgroup's 4 x 48 token slots hold floor(192 / V) whole molecules back to back, the
// transposed copy of X is ONE tile shared by the workgroup over the four wave-private blocks (tools/gen_h3_attn_wide_asm.py)
#define H3W_WAVE_LDS 28672
#define H3W_XT_ROW 416                  // bytes per feature row: 192 tokens + 16 pad (104 dwords: conflict-free ds_read_b128)
#define H3W_XT_LO (128 * H3W_XT_ROW)
#define H3W_NG 5                        // K = 32 key groups in a wave's key window (tw_h3_attns_asm.inc) ...
#define H3W_NG3 3                       // ... or three, for 65 .. 96 atoms at a slot stride of 96 (tw_h3_attns3_asm.inc)
#define H3W_NG6 6                       // ... or all six: one molecule of 161 .. 192 atoms per workgroup (tw_h3_attns6_asm.inc)
#define H3W_FRAG_HEAD(ng) ((ng) * H3_NT * 2048)  // score fragments of one (wave, head): [group][query tile][hi 1 KiB | lo 1 KiB]
#define H3W_SIDE_LDS_OFFSET (H3_RING * H3_STAGE_BYTES + 4 * H3W_WAVE_LDS)
#define H3W_LDS_BYTES (H3W_SIDE_LDS_OFFSET + H3_SIDE_LDS_BYTES)
#define H3_TARGET_MAX 4096.0f       // |w| * 2^s is scaled up to just below this

// stage sequence per net (each 9 KiB = 4 tile pairs + aux); the A and B stages of the chunked MLPs are
// emitted software pipelined,  A(0) | A(1) B(0) | A(2) B(1) | ... | B(n-1)  (h3_mlp_chain):
//   IN   : hid_chunks x { A: [W0 chunk (2 ot x 2 ks) + aux(b0 chunk, scale0)]  B: [W2 chunk ot 0-3][ot 4-7] }
//   layer: H heads x 4 ks x 2 x [Wc_h: four ot (4 half + oo) of k-step ks]  (ks-major: the GEMM for k-step ks
//          can start as soon as the mixing has produced xm[ks])
//          ff_chunks x { A: [W1 chunk o=0 (4 ks) + aux(b1 chunk, scale1)][o=1]  B: [W2 chunk ot 0-3][ot 4-7] }
//   OUT  : hid_chunks x { A: [W0 chunk o=0 + aux][o=1]  B: [W2 chunk (1 pair)] }
// dense variant, attention part of a layer (4 H stages instead of 8 H):  per head pair (2 hp, 2 hp + 1):
//          per head [in_proj rows of q_h: 4 ks][k_h][v_h], then [out_proj k-step hp: ot 0-3][ot 4-7]
// side floats per net: in2_b[128] { n1w n1b [128] b2[128] n2w n2b [128] } out2_b[16]
//                      scales: in0, in2, per layer (wc, w1, w2), out0, out2  (as 2^-s multipliers)
struct H3Geom {
  int hid_chunks, ff_chunks, H, L;
  int in_a_stages;  // A stages per chunk of the in-MLP: 1 (64 input columns) or 3 (192: dense model with 128 RFF features)
  int64_t stages;
  int64_t side_in2b, side_layers, side_layer_size, side_out2b, side_scales, side_size;
  int64_t net_stride_bytes;
};

// h1: the single-MFMA stream (TW_PATH_FUSED_H1).  A stage's four 2 KiB pair slots hold EIGHT fp16 hi
// tiles (tile t at 1 KiB t) instead of four hi / lo pairs, so every chunked MLP is one A stage + one B stage per chunk and
// the attention GEMM one stage per (head, k-step):
//   IN   : hid_chunks x { A: [W0 chunk: tile 2 o + ks (4 tiles) + aux]          B: [W2 chunk: tile ot] }
//   layer: H heads x 4 ks x [Wc_h k-step ks: tile ot]
//          ff_chunks x { A: [W1 chunk: tile 4 o + ks + aux]                     B: [W2 chunk: tile ot] }
//   OUT  : hid_chunks x { A: [W0 chunk: tile 4 o + ks + aux]                    B: [W2 chunk: tile 0] }
static H3Geom h3_geom(const tw_flow_desc& d, bool h1 = false) {
  H3Geom g;
  g.hid_chunks = d.d_hidden / 32;
  g.ff_chunks = d.d_ff / 32;
  g.H = d.n_heads;
  g.L = d.n_layers;
  const int64_t att_stages = d.variant == 1 ? 4LL * g.H : 8LL * g.H;
  g.in_a_stages = (d.variant == 1 && d.d_rff > 0) ? 3 : 1;
  g.stages = (int64_t)(g.in_a_stages + 2) * g.hid_chunks + (int64_t)g.L * (att_stages + 4LL * g.ff_chunks) + 3LL * g.hid_chunks;
  // (dense model: its attention stages keep the split form - q_h k_h v_h and the out_proj half-steps, 4 H stages per layer,
  //  which happens to be the count of the single-MFMA folded attention too; only the MLP sections change)
  //  With position features the in-MLP stays in split form as well - it runs as compiled C++ on 192 input columns)
  if (h1)
    g.stages = (g.in_a_stages == 1 ? 2LL : g.in_a_stages + 2LL) * g.hid_chunks + (int64_t)g.L * (4LL * g.H + 2LL * g.ff_chunks) +
               2LL * g.hid_chunks;
  int64_t o = 0;
  g.side_in2b = o; o += 128;
  g.side_layers = o;
  g.side_layer_size = d.variant == 1 ? H3D_SIDE_LAYER_FLOATS : H3_SIDE_LAYER_FLOATS;
  o += g.L * g.side_layer_size;
  g.side_out2b = o; o += 16;
  g.side_scales = o; o += 4 + 4 * g.L;  // in-MLP 2, layers 3 each, out-MLP 2, then (dense) out_proj's scale per layer
  g.side_size = (o + 63) / 64 * 64 + 256;  // + slack: the per-layer block is fetched as three whole KiB
  g.net_stride_bytes = (g.stages * H3_STAGE_BYTES + g.side_size * 4 + 1023) / 1024 * 1024;
  return g;
}

int64_t h3_packed_bytes(const tw_flow_desc& d, bool h1) {
  H3Geom g = h3_geom(d, h1);
  return g.net_stride_bytes * 2 * d.n_coupling + (H3_RING + 1) * H3_STAGE_BYTES;  // DMA prefetch overrun slack
}

// LDS of h3_score_frag_kernel: [MV][3] coordinates, [H][MV][V] basis values / scores, [H] means, masks and token maps.
// Up to 64 KiB launches as is; up to the CU's 160 KiB after raising the kernel's limit; h3_supported refuses the rest
// (e.g. 48 atoms x 18 heads), so such shapes fall back to the exact-f32 kernels instead of failing at launch.
#define H3_SF_LDS_MAX ((size_t)160 * 1024)
static size_t h3_sf_lds_bytes(int H, int V, int mpw) {
  const size_t MV = (size_t)mpw * V;
  return (MV * 3 + (size_t)H * MV * V + H) * 4 + MV + 32 * H3N4_NT;
}

// Wide layout: molecules per workgroup and, per wave, the byte offset of its key window in a row of the shared X^T tile.
// Wave w's query tokens [48 w, 48 w + 48) touch the molecules overlapping that range; their keys span key tiles
// first .. last; the window is H3W_NG groups of 32 keys from tile K0 = min(first, 12 - 2 NG) (it never leaves the 192
// tokens).  false if the molecule does not fit a workgroup's 192 slots (161 .. 192 atoms: all six groups, H3W_NG6).
struct H3Wide {
  int mpwg;
  int stride;  // token slots per molecule: V (back to back), or 96 (65 .. 96 atoms: each molecule on its own pair of waves)
  int ng;      // key groups of a wave's window: 5 (160 keys), or 3 with the 96-slot stride (a wave's keys = its molecule's 96 slots)
  int win[4];  // bytes: 32 * K0
  int nt = H3_NT;  // token tiles per wave: 3, or 4 = the PAIRED layout (r05): one molecule of 97 .. 128 atoms per pair of 64-token
                   // waves, two per workgroup (slot stride 128, four key groups = the molecule's 128 slots; tools/gen_h3_attn_asm.py --pair)
};
#define H3P_NG 4
// bytes of the score fragments of one (wave, head): [group][query tile] (wide) or [query tile][group] (paired) x [hi 1 KiB | lo 1 KiB]
static inline int64_t h3w_frag_head(const H3Wide& w) { return (int64_t)w.ng * w.nt * 2048; }
// The paired layout exists as the encoder-stack statement only: a launch that asks for activation dumps / section stamps between
// the sections (tw_debug_netblock, tw_debug_set_flags bits 2, 4, 12 without 13) takes the 48-token wide layout instead, and so
// does tw_debug_set_flags bit 20 (1048576; A/B, tests).
static thread_local bool t_no_pair = false;
static bool h3_pair_enabled();
struct NoPairScope {  // the calling thread's layout decisions inside the scope never take the paired layout (when `on`)
  bool prev;
  explicit NoPairScope(bool on = true) : prev(t_no_pair) { t_no_pair = prev || on; }
  ~NoPairScope() { t_no_pair = prev; }
};
// Geometrically possible from 25 atoms on (below that a 48-token wave already holds two or more whole molecules and its
// windowed mixing is cheaper).  Molecules sit back to back (slot stride V) with five-group windows - except 65 .. 96 atoms:
// two molecules per workgroup either way, so each takes 96 slots (molecule 0 on waves 0-1, molecule 1 on waves 2-3, the
// slots between V and 96 are padding tokens) and a wave's keys are its own molecule's three groups: 54 instead of 90 mixing
// MFMAs per head and k-step, 18 instead of 30 fragment loads per head.  (Back to back, 81 .. 95 atoms would even need a SIXTH
// group for wave 1 and were refused until r04.)  tw_debug_set_flags bit 18 (262144): five-group statement only (A/B, tests).
#define H3W_MIN_ATOMS 25
static bool h3_wide_geom_stride(int V, int P, int NG, H3Wide* w) {
  w->mpwg = (64 * H3_NT) / P;
  w->stride = P;
  w->ng = NG;
  for (int wave = 0; wave < 4; ++wave) {
    const int lo = 16 * H3_NT * wave, hi = lo + 16 * H3_NT - 1;
    int m0 = lo / P, m1 = hi / P;
    if (lo - m0 * P >= V) ++m0;  // the wave starts in the padding behind molecule m0
    if (m1 >= w->mpwg) m1 = w->mpwg - 1;
    int k0 = 0;
    if (m0 <= m1) {
      const int first = (m0 * P) / 16, last = (m1 * P + V - 1) / 16;
      if ((last - first + 2) / 2 > NG) return false;
      k0 = first < 12 - 2 * NG ? first : 12 - 2 * NG;
    }
    w->win[wave] = 32 * k0;
  }
  return true;
}
static bool h3_wide_geom(int V, H3Wide* w, bool allow_pair = true) {
  if (V < H3W_MIN_ATOMS || V > 64 * H3_NT) return false;
  w->nt = H3_NT;
  if (V > 96 && V <= 128 && allow_pair && h3_pair_enabled()) {
    w->mpwg = 2;
    w->stride = 128;
    w->ng = H3P_NG;
    w->nt = H3N4_NT;
    for (int i = 0; i < 4; ++i) w->win[i] = 0;
    return true;
  }
  if (V > 64 && V <= 96 && !(g_debug_flags & 262144) && h3_wide_geom_stride(V, 96, H3W_NG3, w)) return true;
  if (h3_wide_geom_stride(V, V, H3W_NG, w)) return true;
  if (V <= 96) return h3_wide_geom_stride(V, 96, H3W_NG, w);
  return h3_wide_geom_stride(V, V, H3W_NG6, w);  // 161 .. 192 atoms: the whole 192-key row
}
static size_t h3w_sf_lds_bytes(int V, int mpwg) {
  const size_t MV = (size_t)mpwg * V;
  return (MV * 3 + MV * V) * 4 + MV + 16;
}

static bool h3_narrow_ok(const tw_flow_desc& d, int V) {
  FusedGeom fg;
  return fused_geom_nt(V, H3_NT, &fg) && h3_sf_lds_bytes(d.n_heads, V, fg.mpw) <= H3_SF_LDS_MAX;
}
static bool h3_wide_ok(int V) {
  H3Wide wd;
  return h3_wide_geom(V, &wd) && h3w_sf_lds_bytes(V, wd.mpwg) <= H3_SF_LDS_MAX;
}
// Which layout a kernel-attention launch of n_rows conformations of V atoms takes.  Above 48 atoms only the wide one
// exists.  From 25 to 48 atoms both do: a 48-token wave holds ONE such molecule (4 per workgroup, 52-100 % of the token
// slots), the wide packing floor(192 / V) (V = 30: 6, 94 %) - fewer workgroups, but each pays the wide mixing (90 instead
// of 36 MFMAs per head and k-step on the shared X^T tile, two more workgroup barriers per layer: measured 1.12-1.24x per
// workgroup).  One workgroup occupies a CU, so what counts is ROUNDS of the chip: the wide layout is taken when
// rounds x cost is lower (V = 30: 768 proposals -> 1 round instead of 2; 1000 proposals -> 2 rounds either way, narrow).
// tw_debug_set_flags bit 14 (16384): never wide below 49 atoms; bit 15 (32768): always wide where it exists (A/B, tests).
#define H3_CUS 256
// 64-token waves (NT = 4): ONE molecule of 49-64 atoms per wave, four per workgroup - against the wide layout's floor(192 / V)
// = 3 per workgroup, but without its shared tile, its 160-key windows and its two extra barriers per layer.  What decides
// is rounds of the chip x cost per workgroup, in units of the 48-token kernel's workgroup: wide 1.2 (measured), 64-token
// H3N4_COST - measured 1.03-1.06x the wide workgroup with the FFN as generated asm and the rest compiled C++
// (profiles/r04_nt4_layout_choice.txt: 60 atoms x 256 proposals 4.03 against 3.80 ms, x 768 8.37 against 8.10).  BASELINE
// config 3 (60 atoms x 512 proposals): 128 workgroups per net = ONE round against the wide layout's two: 4.73 against 7.45 ms.
// tw_debug_set_flags bit 16 (65536): always where it exists; bit 17 (131072): never (A/B, tests).
#define H3N4_COST 1.25
static bool h3_nt4_ok(const tw_flow_desc& d, int V, bool h1) {
  // kernel attention: both the split-fp16 and the single-MFMA stream have a 64-token build; dense softmax model (r05): the
  // split-fp16 per-section build only, no position features
  if (d.variant == 1) return !h1 && d.d_rff == 0 && V > 16 * H3_NT && V <= 16 * H3N4_NT;
  return d.variant == 0 && V > 16 * H3_NT && V <= 16 * H3N4_NT && h3_sf_lds_bytes(d.n_heads, V, 1) <= H3_SF_LDS_MAX;
}
static int64_t h3_rounds(int64_t wgs_per_net) { return (8 * ((wgs_per_net + 3) / 4) + H3_CUS - 1) / H3_CUS; }
static bool h3_nt4_choice(const tw_flow_desc& d, int V, int64_t n_rows, bool h1) {
  if (d.variant == 1) return h3_nt4_ok(d, V, h1);  // the dense model has no other layout above 48 atoms
  if (!h3_nt4_ok(d, V, h1) || (g_debug_flags & 131072)) return false;
  if (g_debug_flags & 65536) return true;
  H3Wide wd;
  if (!h3_wide_ok(V) || !h3_wide_geom(V, &wd)) return true;
  const int64_t rows = n_rows > 0 ? n_rows : 1;
  return H3N4_COST * (double)h3_rounds((rows + 3) / 4) < 1.2 * (double)h3_rounds((rows + wd.mpwg - 1) / wd.mpwg);
}
static bool h3_wide_choice(const tw_flow_desc& d, int V, int64_t n_rows, bool h1) {
  if (h3_nt4_choice(d, V, n_rows, h1)) return false;
  if (d.variant != 0 || !h3_wide_ok(V)) return false;
  if (!h3_narrow_ok(d, V)) return true;
  if (g_debug_flags & 16384) return false;
  if (g_debug_flags & 32768) return true;
  FusedGeom fg;
  H3Wide wd;
  fused_geom_nt(V, H3_NT, &fg);
  h3_wide_geom(V, &wd);
  if (wd.mpwg <= 4 * fg.mpw) return false;  // no denser than the narrow layout
  const int64_t rows = n_rows > 0 ? n_rows : 1;
  const int64_t wg_n = ((rows + fg.mpw - 1) / fg.mpw + 3) / 4, wg_w = (rows + wd.mpwg - 1) / wd.mpwg;
  auto rounds = [](int64_t wgs_per_net) { return (8 * ((wgs_per_net + 3) / 4) + H3_CUS - 1) / H3_CUS; };
  return 1.2 * (double)rounds(wg_w) < (double)rounds(wg_n);  // measured 1.12 (26-30 atoms) .. 1.24 (36-44): profiles/r04_layout_choice.txt
}

bool h3_supported(const tw_flow_desc& d, int n_atoms) {
  FusedGeom fg;
  if (d.variant == 0)
    return d.d_model == 128 && d.d_hidden % 32 == 0 && d.d_ff % 32 == 0 && d.d_emb + 9 <= 64 &&
           (h3_narrow_ok(d, n_atoms) || h3_wide_ok(n_atoms) || h3_nt4_ok(d, n_atoms, false));
  if (d.variant == 1)  // dense softmax attention: 8 heads of 16 = one MFMA tile each; input width <= 64, or 32 + 9 + 128
                       // random Fourier position features (192 columns: the in-MLP then runs as compiled C++)
    return d.d_model == 128 && d.n_heads == 8 && d.d_hidden % 32 == 0 && d.d_ff % 32 == 0 &&
           ((d.d_rff == 0 && d.d_emb + 9 <= 64) || (d.d_rff == 128 && d.d_emb == 32)) &&
           (fused_geom_nt(n_atoms, H3_NT, &fg) || h3_nt4_ok(d, n_atoms, false));
  return false;
}

// the single-MFMA variant exists for kernel attention (the 48-token encoder-stack build and the wide layout) and for the dense
// model on 48-token waves (in / FFN / out sections single-MFMA, the softmax attention block in split form; with position
// features the in-MLP stays in split form too - r05)
bool h1_supported(const tw_flow_desc& d, int n_atoms) {
  FusedGeom fg;
  if (d.variant == 1) return h3_supported(d, n_atoms) && fused_geom_nt(n_atoms, H3_NT, &fg);  // (48-token waves only)
  return d.variant == 0 && h3_supported(d, n_atoms);
}

// ================================================================================================
// packing: fp32 raw weights -> scaled fp16 hi/lo tile pairs
// pair (ot, ks) element (lane, e): W[16 ot + (lane&15)][32 ks + 16 (e/4) + 4 (lane>>4) + e%4]
// ================================================================================================
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  // v >= 0: integer compare is order preserving
  atomicMax((int*)addr, __float_as_int(v));
}

__global__ void h3_absmax_kernel(const float* __restrict__ src, int64_t n, float* __restrict__ out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(src[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomic_max_float(out, m);
}

__device__ __forceinline__ double fold_elem(const float* wv, const float* wo, int H, int h, int o_row, int i_col) {
  double acc = 0.0;
  for (int k = 0; k < 128; ++k)
    acc += (double)wo[(int64_t)o_row * (H * 128) + h * 128 + k] * (double)wv[(int64_t)(h * 128 + k) * 128 + i_col];
  return acc;
}

__global__ void h3_fold_absmax_kernel(const float* __restrict__ wv, const float* __restrict__ wo, int H,
                                      float* __restrict__ out) {
  const int h = blockIdx.y;
  float m = 0.f;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 128 * 128; idx += gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf((float)fold_elem(wv, wo, H, h, idx / 128, idx % 128)));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomic_max_float(out, m);
}

// scale exponent: largest power of two with max * 2^s < H3_TARGET_MAX; writes 2^s and 2^-s
__global__ void h3_scale_kernel(const float* __restrict__ absmax, float* __restrict__ up, float* __restrict__ down) {
  float m = absmax[0];
  int s = 0;
  if (m > 0.f && isfinite(m)) {
    s = (int)floorf(log2f(H3_TARGET_MAX / m));
    if (s > 24) s = 24;
    if (s < -24) s = -24;
    while (ldexpf(m, s) >= H3_TARGET_MAX) --s;
  }
  up[0] = ldexpf(1.f, s);
  down[0] = ldexpf(1.f, -s);
}

__device__ __forceinline__ void store_pair(char* pair, int lane, int e, float v) {
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  ((_Float16*)(pair + lane * 16))[e] = hi;
  ((_Float16*)(pair + 1024 + lane * 16))[e] = lo;
}

__global__ void h3_unit_scale_kernel(float* __restrict__ up, float* __restrict__ down) { up[0] = down[0] = 1.0f; }

// tile pairs ordered ot-major over (n_ot, n_ks); src row-major [rows, cols] (ld)
// h1: hi tiles only, 1 KiB each, in the same (ot, ks) order
__global__ void h3_pack_block_kernel(const float* __restrict__ src, int ld, int rows_valid, int cols_valid, int row0,
                                     int col0, int n_ks, const float* __restrict__ scale_up, char* __restrict__ dst, int h1) {
  const int ot = blockIdx.x, ks = blockIdx.y, lane = threadIdx.x;
  const float sc = scale_up[0];
  const int row = row0 + 16 * ot + (lane & 15);
  char* pair = dst + (int64_t)(ot * n_ks + ks) * (h1 ? 1024 : H3_PAIR_BYTES);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = col0 + 32 * ks + 16 * (e / 4) + 4 * (lane >> 4) + (e % 4);
    const float v = (row < rows_valid && col < cols_valid) ? src[(int64_t)row * ld + col] * sc : 0.f;
    if (h1) ((_Float16*)(pair + lane * 16))[e] = (_Float16)v;
    else store_pair(pair, lane, e, v);
  }
}

// one Wc stage = k-step ks of four output tiles (ot = 4 half + oo): tile pair oo
// h1: the grid runs over all eight output tiles of the k-step (half = 0), hi tiles of 1 KiB
__global__ void h3_pack_fold_kernel(const float* __restrict__ wv, const float* __restrict__ wo, int H, int h, int ks,
                                    int half, const float* __restrict__ scale_up, char* __restrict__ dst, int h1) {
  const int oo = blockIdx.x, lane = threadIdx.x;
  const float sc = scale_up[0];
  const int row = 16 * (4 * half + oo) + (lane & 15);
  char* pair = dst + (int64_t)oo * (h1 ? 1024 : H3_PAIR_BYTES);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = 32 * ks + 16 * (e / 4) + 4 * (lane >> 4) + (e % 4);
    const float v = (float)(fold_elem(wv, wo, H, h, row, col) * (double)sc);
    if (h1) ((_Float16*)(pair + lane * 16))[e] = (_Float16)v;
    else store_pair(pair, lane, e, v);
  }
}

__global__ void h3_copy_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int n_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) dst[i] = i < n ? src[i] : 0.f;
}

int h3_pack_weights(const tw_flow_desc& d, const float* raw, char* packed, float* scratch /* >= 64 floats */,
                    hipStream_t s, bool h1) {
  const RawLayout L = raw_layout(d);
  const H3Geom g = h3_geom(d, h1);
  TW_HIP_CHECK(hipMemsetAsync(packed, 0, h3_packed_bytes(d, h1), s));
  auto absmax = [&](const float* src, int64_t n, float* up, float* down) -> int {
    TW_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(float), s));
    hipLaunchKernelGGL(h3_absmax_kernel, dim3(64), dim3(256), 0, s, src, n, scratch);
    TW_LAUNCH_CHECK();
    hipLaunchKernelGGL(h3_scale_kernel, dim3(1), dim3(1), 0, s, scratch, up, down);
    TW_LAUNCH_CHECK();
    return TW_OK;
  };
  auto block_fmt = [&](const float* src, int ld, int rows_valid, int cols_valid, int row0, int col0, int n_ot, int n_ks,
                       const float* up, char* dst, int hi_only) -> int {
    hipLaunchKernelGGL(h3_pack_block_kernel, dim3(n_ot, n_ks), dim3(64), 0, s, src, ld, rows_valid, cols_valid, row0,
                       col0, n_ks, up, dst, hi_only);
    TW_LAUNCH_CHECK();
    return TW_OK;
  };
  auto block = [&](const float* src, int ld, int rows_valid, int cols_valid, int row0, int col0, int n_ot, int n_ks,
                   const float* up, char* dst) -> int {
    return block_fmt(src, ld, rows_valid, cols_valid, row0, col0, n_ot, n_ks, up, dst, h1 ? 1 : 0);
  };
  auto copy = [&](const float* src, int n, float* dst, int n_pad) -> int {
    hipLaunchKernelGGL(h3_copy_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, s, src, n, dst, n_pad);
    TW_LAUNCH_CHECK();
    return TW_OK;
  };
  float* up = scratch + 8;  // scratch[8] holds the current 2^s
  int rc;
  // stage index of the A / B stages of chunk `ch` in the pipelined order  A(0) | A(1) B(0) | ... | B(n-1)
  auto a_off = [](int ch, int /*n*/, int A, int B) -> int64_t { return ch == 0 ? 0 : A + (int64_t)(ch - 1) * (A + B); };
  auto b_off = [](int ch, int n, int A, int B) -> int64_t { return A + (int64_t)ch * (A + B) + (ch < n - 1 ? A : 0); };
  for (int c = 0; c < d.n_coupling; ++c)
    for (int net = 0; net < 2; ++net) {
      const float* nb = raw + net_base(L, c, net);
      char* pn = packed + (int64_t)(c * 2 + net) * g.net_stride_bytes;
      float* side = (float*)(pn + g.stages * H3_STAGE_BYTES);
      float* scales = side + g.side_scales;  // 2^-s multipliers, in stream order
      char* st = pn;
      // ---- IN
      if ((rc = absmax(nb + L.net.in0_w, (int64_t)d.d_hidden * L.d_in, up, scales + 0))) return rc;
      if (h1 && !d.d_rff) {  // fast mode: first-layer weights unscaled (see the FFN below); the RFF in-MLP is compiled, split form
        hipLaunchKernelGGL(h3_unit_scale_kernel, dim3(1), dim3(1), 0, s, up, scales + 0);
        TW_LAUNCH_CHECK();
      }
      const int ia = g.in_a_stages, ks_in = 2 * ia;  // k-steps of the first GEMM: 2 or 6
      const int ib = h1 ? 1 : 2;                      // B stages per chunk of the in-MLP / FFN
      const bool in_h1 = h1 && !d.d_rff;              // (position features: the in-MLP keeps the split form, see h3_geom)
      const int ib_in = in_h1 ? 1 : 2;
      if (in_h1) {
        for (int ch = 0; ch < g.hid_chunks; ++ch) {
          char* a = st + a_off(ch, g.hid_chunks, 1, 1) * H3_STAGE_BYTES;
          if ((rc = block(nb + L.net.in0_w, L.d_in, d.d_hidden, L.d_in, 32 * ch, 0, 2, 2, up, a))) return rc;  // tile 2 o + ks
          if ((rc = copy(nb + L.net.in0_b + 32 * ch, 32, (float*)(a + H3_STAGE_TILE_BYTES), 32))) return rc;
          if ((rc = copy(scales + 0, 1, (float*)(a + H3_STAGE_TILE_BYTES) + 32, 1))) return rc;
        }
      } else
      for (int ch = 0; ch < g.hid_chunks; ++ch)
        for (int a_ = 0; a_ < ia; ++a_) {
          char* a = st + (a_off(ch, g.hid_chunks, ia, 2) + a_) * H3_STAGE_BYTES;
          for (int pp = 0; pp < H3_STAGE_PAIRS; ++pp) {  // pair q = (o, ks), four to a stage (h3_mlp_chain)
            const int q = a_ * H3_STAGE_PAIRS + pp, o = q / ks_in, ks = q % ks_in;
            if ((rc = block_fmt(nb + L.net.in0_w, L.d_in, d.d_hidden, L.d_in, 32 * ch + 16 * o, 32 * ks, 1, 1, up, a + pp * H3_PAIR_BYTES,
                                in_h1 ? 1 : 0)))
              return rc;
          }
          if (a_ == 0) {
            if ((rc = copy(nb + L.net.in0_b + 32 * ch, 32, (float*)(a + H3_STAGE_TILE_BYTES), 32))) return rc;
            if ((rc = copy(scales + 0, 1, (float*)(a + H3_STAGE_TILE_BYTES) + 32, 1))) return rc;
          }
        }
      if ((rc = absmax(nb + L.net.in2_w, (int64_t)128 * d.d_hidden, up, scales + 1))) return rc;
      for (int ch = 0; ch < g.hid_chunks; ++ch)
        for (int hf = 0; hf < ib_in; ++hf) {
          char* b = st + (b_off(ch, g.hid_chunks, ia, ib_in) + hf) * H3_STAGE_BYTES;
          if ((rc = block_fmt(nb + L.net.in2_w, d.d_hidden, 128, d.d_hidden, 64 * hf, 32 * ch, in_h1 ? 8 : 4, 1, up, b, in_h1 ? 1 : 0)))
            return rc;
        }
      st += (int64_t)(ia + ib_in) * g.hid_chunks * H3_STAGE_BYTES;
      if ((rc = copy(nb + L.net.in2_b, 128, side + g.side_in2b, 128))) return rc;
      // ---- layers
      for (int l = 0; l < d.n_layers; ++l) {
        const float* lb = nb + L.net.layers + (int64_t)l * L.layer.size;
        float* sl = side + g.side_layers + (int64_t)l * g.side_layer_size;
        float* lsc = scales + 2 + 3 * l;
        if (d.variant == 1) {
          // dense: in_proj [384, 128] (one scale) as per-head stages q_h | k_h | v_h of four k-steps each, and after every
          // second head the k-step of out_proj [128, 128] (second scale) those two heads feed
          float* osc = sl + 642;
          // stage order = consumption order of the attention section (tools/gen_h3_dense_attn_asm.py): q_0 k_0, then per
          // head h: v_h, q_{h+1} k_{h+1} (while the softmax of head h runs), and after an odd h the out_proj k-step of its pair
          std::vector<std::pair<int, int>> order;  // (kind 0 q / 1 k / 2 v / 3 out_proj, head or 2 * pair + half)
          order.push_back({0, 0});
          order.push_back({1, 0});
          for (int h = 0; h < d.n_heads; ++h) {
            order.push_back({2, h});
            if (h + 1 < d.n_heads) {
              order.push_back({0, h + 1});
              order.push_back({1, h + 1});
            }
            if (h % 2 == 1) {
              order.push_back({3, 2 * (h / 2)});
              order.push_back({3, 2 * (h / 2) + 1});
            }
          }
          for (int pass = 0; pass < 2; ++pass) {  // pass 0: in_proj stages (scale of in_proj in `up`), pass 1: out_proj stages
            if (pass == 0) {
              if ((rc = absmax(lb + L.layer.in_w, (int64_t)384 * 128, up, lsc + 0))) return rc;
            } else {
              if ((rc = absmax(lb + L.layer.out_w, (int64_t)128 * 128, up, osc))) return rc;
            }
            for (size_t i = 0; i < order.size(); ++i) {
              char* a = st + (int64_t)i * H3_STAGE_BYTES;
              const int kind = order[i].first, idx = order[i].second;
              // (hi / lo pairs also in the single-MFMA stream: the softmax attention block stays in split form there)
              if (pass == 0 && kind < 3) {
                if ((rc = block_fmt(lb + L.layer.in_w, 128, 384, 128, kind * 128 + 16 * idx, 0, 1, 4, up, a, 0))) return rc;
              } else if (pass == 1 && kind == 3) {
                if ((rc = block_fmt(lb + L.layer.out_w, 128, 128, 128, 64 * (idx % 2), 32 * (idx / 2), 4, 1, up, a, 0))) return rc;
              }
            }
          }
          st += (int64_t)4 * d.n_heads * H3_STAGE_BYTES;
          if ((rc = copy(lb + L.layer.in_b, 384, sl + H3D_INB, 384))) return rc;
          if ((rc = copy(lb + L.layer.out_b, 128, sl + H3D_OUTB, 128))) return rc;
          // out_proj's scale once more where the encoder-stack statement can read it a layer AHEAD (it seeds the attention
          // block's accumulators with x / s_o before that layer's side block is in the LDS): scales[4 + 3 L + l]
          if ((rc = copy(osc, 1, scales + 4 + 3 * d.n_layers + l, 1))) return rc;
        } else {
        // folded attention: one scale for all heads of the layer
        TW_HIP_CHECK(hipMemsetAsync(scratch, 0, sizeof(float), s));
        hipLaunchKernelGGL(h3_fold_absmax_kernel, dim3(16, d.n_heads), dim3(256), 0, s, lb + L.layer.wv, lb + L.layer.wo,
                           d.n_heads, scratch);
        TW_LAUNCH_CHECK();
        hipLaunchKernelGGL(h3_scale_kernel, dim3(1), dim3(1), 0, s, scratch, up, lsc + 0);
        TW_LAUNCH_CHECK();
        for (int h = 0; h < d.n_heads; ++h)
          for (int ks = 0; ks < 4; ++ks)
            for (int half = 0; half < (h1 ? 1 : 2); ++half) {
              hipLaunchKernelGGL(h3_pack_fold_kernel, dim3(h1 ? 8 : 4), dim3(64), 0, s, lb + L.layer.wv, lb + L.layer.wo, d.n_heads, h,
                                 ks, half, up, st + (int64_t)(h1 ? 4 * h + ks : 8 * h + 2 * ks + half) * H3_STAGE_BYTES, h1 ? 1 : 0);
              TW_LAUNCH_CHECK();
            }
        st += (int64_t)(h1 ? 4 : 8) * d.n_heads * H3_STAGE_BYTES;
        }
        if ((rc = absmax(lb + L.layer.w1, (int64_t)d.d_ff * 128, up, lsc + 1))) return rc;
        if (h1) {
          // Fast mode: W1 goes in UNSCALED (there is no lo half whose exponent range the scale protects), so the hidden
          // pre-activation is the accumulator itself and the FFN's epilogue needs neither the multiply nor - with the bias as
          // the chain's start value - the add (tools/gen_h3_ffn_asm.py fold()).  The scale slots carry 1.0.
          hipLaunchKernelGGL(h3_unit_scale_kernel, dim3(1), dim3(1), 0, s, up, lsc + 1);
          TW_LAUNCH_CHECK();
        }
        const int fa = h1 ? 1 : 2;  // A stages per chunk of the FFN / out-MLP (h1: both o in one stage, tile 4 o + ks)
        for (int ch = 0; ch < g.ff_chunks; ++ch)
          for (int o = 0; o < fa; ++o) {
            char* a = st + (a_off(ch, g.ff_chunks, fa, ib) + o) * H3_STAGE_BYTES;
            if ((rc = block(lb + L.layer.w1, 128, d.d_ff, 128, 32 * ch + 16 * o, 0, h1 ? 2 : 1, 4, up, a))) return rc;
            if (o == 0) {
              if ((rc = copy(lb + L.layer.b1 + 32 * ch, 32, (float*)(a + H3_STAGE_TILE_BYTES), 32))) return rc;
              if ((rc = copy(lsc + 1, 1, (float*)(a + H3_STAGE_TILE_BYTES) + 32, 1))) return rc;
            }
          }
        if ((rc = absmax(lb + L.layer.w2, (int64_t)128 * d.d_ff, up, lsc + 2))) return rc;
        for (int ch = 0; ch < g.ff_chunks; ++ch)
          for (int hf = 0; hf < ib; ++hf) {
            char* b = st + (b_off(ch, g.ff_chunks, fa, ib) + hf) * H3_STAGE_BYTES;
            if ((rc = block(lb + L.layer.w2, d.d_ff, 128, d.d_ff, 64 * hf, 32 * ch, h1 ? 8 : 4, 1, up, b))) return rc;
          }
        st += (int64_t)(fa + ib) * g.ff_chunks * H3_STAGE_BYTES;
        if ((rc = copy(lb + L.layer.n1w, 128, sl, 128))) return rc;
        if ((rc = copy(lb + L.layer.n1b, 128, sl + 128, 128))) return rc;
        if ((rc = copy(lb + L.layer.b2, 128, sl + 256, 128))) return rc;
        if ((rc = copy(lb + L.layer.n2w, 128, sl + 384, 128))) return rc;
        if ((rc = copy(lb + L.layer.n2b, 128, sl + 512, 128))) return rc;
        if ((rc = copy(lsc + 0, 1, sl + 640, 1))) return rc;  // folded-attention scale
        if ((rc = copy(lsc + 2, 1, sl + 641, 1))) return rc;  // W2 scale
      }
      // ---- OUT
      float* osc = scales + 2 + 3 * d.n_layers;
      if ((rc = absmax(nb + L.net.out0_w, (int64_t)d.d_hidden * 128, up, osc + 0))) return rc;
      if (h1) {
        hipLaunchKernelGGL(h3_unit_scale_kernel, dim3(1), dim3(1), 0, s, up, osc + 0);
        TW_LAUNCH_CHECK();
      }
      const int oa = h1 ? 1 : 2;
      for (int ch = 0; ch < g.hid_chunks; ++ch)
        for (int o = 0; o < oa; ++o) {
          char* a = st + (a_off(ch, g.hid_chunks, oa, 1) + o) * H3_STAGE_BYTES;
          if ((rc = block(nb + L.net.out0_w, 128, d.d_hidden, 128, 32 * ch + 16 * o, 0, h1 ? 2 : 1, 4, up, a))) return rc;
          if (o == 0) {
            if ((rc = copy(nb + L.net.out0_b + 32 * ch, 32, (float*)(a + H3_STAGE_TILE_BYTES), 32))) return rc;
            if ((rc = copy(osc + 0, 1, (float*)(a + H3_STAGE_TILE_BYTES) + 32, 1))) return rc;
          }
        }
      if ((rc = absmax(nb + L.net.out2_w, (int64_t)3 * d.d_hidden, up, osc + 1))) return rc;
      for (int ch = 0; ch < g.hid_chunks; ++ch) {
        char* b = st + b_off(ch, g.hid_chunks, oa, 1) * H3_STAGE_BYTES;
        if ((rc = block(nb + L.net.out2_w, d.d_hidden, 3, d.d_hidden, 0, 32 * ch, 1, 1, up, b))) return rc;
      }
      if ((rc = copy(nb + L.net.out2_b, 3, side + g.side_out2b, 16))) return rc;
    }
  return TW_OK;
}

// ================================================================================================
// block-diagonal score fragments, fp16 hi/lo:
// per (blk, head, jt): [k-step 0: 64 x h8 hi][64 x h8 lo][k-step 1 (keys 32..47): 64 x h4 hi][64 x h4 lo] = 3 KiB
//   k-step 0 element (lane, e): S[query 16jt + (lane&15)][key 8 (lane>>4) + e]
//   k-step 1 element (lane, e): S[query 16jt + (lane&15)][key 32 + 4 (lane>>4) + e]
// ================================================================================================
#define H3_SF_BYTES 3072

// use_mm: torch.cdist's matmul formulation, which the reference gets above 25 atoms (same arithmetic as
// tw_netblock.hip::pair_dist / tw_kernels.hip::pair_distance)
__device__ __forceinline__ float h3_pair_dist(const float* x, int q, int m, int use_mm) {
  const float qx = x[3 * q], qy = x[3 * q + 1], qz = x[3 * q + 2];
  const float mx = x[3 * m], my = x[3 * m + 1], mz = x[3 * m + 2];
  if (!use_mm) {
    float dx = qx - mx, dy = qy - my, dz = qz - mz;
    return sqrtf(dx * dx + dy * dy + dz * dz);
  }
  return tw_cdist_mm(qx, qy, qz, mx, my, mz);
}

// One workgroup per wave-block (all heads): distances and basis values once per (pair, head), rows normalised in LDS, then
// one thread per (head, query tile, lane) assembles that lane's 16 + 16 + 8 + 8 fragment bytes and stores them whole.
// (The first version ran one workgroup per (block, head) and one thread per fragment element: its runtime integer
// divisions and the basis values recomputed per element made it 42 us per 1000-row forward pass against 16 us now, bit-identical output:
// profiles/r02_ab_score_frag.txt.)
__global__ void h3_score_frag_kernel(const float* __restrict__ x, const uint8_t* __restrict__ masked,
                                     const float* __restrict__ ls, int H, int V, int mpw, int64_t n_rows,
                                     int64_t n_cond, int normalise, char* __restrict__ sfrag, ScoreBasis basis,
                                     int64_t variant_bytes, int windowed, int use_mm, int nt) {
  extern __shared__ float sm[];
  const int MV = mpw * V, MVV = MV * V;
  const int sf_bytes = nt == 4 ? H3N4_SF_BYTES : H3_SF_BYTES;
  float* xs = sm;                // [MV][3]
  float* E = xs + MV * 3;        // [H][MV][V]: basis values, then normalised scores
  float* cmean_s = E + H * MVV;  // [H]
  uint8_t* msk = (uint8_t*)(cmean_s + H);  // [MV]
  uint8_t* tmol = msk + MV;                // [16 NT] token -> molecule of the block (255: padding token)
  uint8_t* tatm = tmol + 16 * nt;          // [16 NT] token -> atom
  const int64_t blk = blockIdx.x;
  const int nthr = blockDim.x, t = threadIdx.x;
  for (int i = t; i < MV; i += nthr) {
    const int q = i / V, a = i - q * V;
    int64_t n = blk * mpw + q;
    if (n >= n_rows) n = n_rows - 1;
    const int64_t src = (n % n_cond) * V + a;
    xs[3 * i] = x[3 * src];
    xs[3 * i + 1] = x[3 * src + 1];
    xs[3 * i + 2] = x[3 * src + 2];
    msk[i] = masked[src];
  }
  for (int i = t; i < 16 * nt; i += nthr) {
    const int q = i / V;
    tmol[i] = q < mpw ? (uint8_t)q : (uint8_t)255;
    tatm[i] = (uint8_t)(i - q * V);
  }
  const float* cf0 = nullptr;
  if (basis.order > 0) {
    if (t < H) {
      float cm;
      basis_coeffs(basis, blockIdx.z, t, &cm);
      cmean_s[t] = cm;
    }
    cf0 = basis.coeff0 + ((int)blockIdx.z / basis.n_layers) * basis.net_stride +
          ((int)blockIdx.z % basis.n_layers) * basis.layer_stride;
  } else if (t < H) {
    cmean_s[t] = 0.f;
  }
  __syncthreads();
  for (int i = t; i < MVV; i += nthr) {
    const int qa = i / V, m = i - qa * V;
    const int q = tmol[qa];
    const float dd = h3_pair_dist(xs + q * V * 3, qa - q * V, m, use_mm);
    const bool dead = msk[q * V + m] != 0;
    for (int h = 0; h < H; ++h) {
      const float sc = dd / ls[h];
      E[h * MVV + i] = dead ? 0.f : basis_value(sc, cf0 + (int64_t)h * basis.order, basis.order, cmean_s[h]);
    }
  }
  __syncthreads();
  for (int i = t; i < H * MV; i += nthr) {
    float* row = E + (int64_t)i * V;  // (h, q, a) rows are contiguous
    double sum = 0.0;
    for (int m = 0; m < V; ++m) sum += (double)fabsf(row[m]);
    const float den = (float)sum + 1e-5f;
    if (normalise)
      for (int m = 0; m < V; ++m) row[m] = row[m] / den;
  }
  __syncthreads();
  char* out = sfrag + blockIdx.z * variant_bytes + blk * (int64_t)H * nt * sf_bytes;
  for (int i = t; i < H * nt * 64; i += nthr) {
    const int lane = i & 63, hj = i >> 6;
    const int h = hj / nt, jt = hj - h * nt;
    const int tq = 16 * jt + (lane & 15), g = lane >> 4;
    const int mq = tmol[tq];
    const float* row = E + h * MVV + ((mq == 255 ? 0 : mq) * V + tatm[tq]) * V;
    // windowed (h3_windowed): query tile 2 only has keys in [16, 48) - its K = 32 block covers those, its K = 16 block
    // is not used (nor is tile 0's, whose keys all lie in [0, 32))
    // K = 32 block: the K index runs over two key tiles, element e of lane group g = key 16 Ta + 4 g + e (e < 4),
    // 16 Tb + 4 g + e - 4 (e >= 4) - the order the transposed copy of x comes out of the matrix pipe in (kernel, "x ->
    // transposed"); (Ta, Tb) = (0, 1), windowed tile 2: (1, 2)
    const int k0 = ((windowed && jt == 2) ? 16 : 0) + 4 * g, k1 = 32 + 4 * g;
    h8 hi0, lo0;
    h4 hi1, lo1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int tk = k0 + (e < 4 ? e : e + 12);
      const float val = (mq != 255 && tmol[tk] == mq) ? row[tatm[tk]] : 0.f;
      const _Float16 hi = (_Float16)val;
      hi0[e] = hi;
      lo0[e] = (_Float16)(val - (float)hi);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int tk = k1 + e;
      const float val = (mq != 255 && tmol[tk] == mq) ? row[tatm[tk]] : 0.f;
      const _Float16 hi = (_Float16)val;
      hi1[e] = hi;
      lo1[e] = (_Float16)(val - (float)hi);
    }
    char* base = out + (int64_t)hj * sf_bytes;
    *(h8*)(base + lane * 16) = hi0;
    *(h8*)(base + 1024 + lane * 16) = lo0;
    if (nt == 4) {
      // 64-token waves: the second k-step is a K = 32 block as well - keys 32 + 4 g + e (e < 4), 48 + 4 g + e - 4 (e >= 4)
      h8 hi2, lo2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int tk = 32 + 4 * g + (e < 4 ? e : e + 12);
        const float val = (mq != 255 && tmol[tk] == mq) ? row[tatm[tk]] : 0.f;
        const _Float16 hi = (_Float16)val;
        hi2[e] = hi;
        lo2[e] = (_Float16)(val - (float)hi);
      }
      *(h8*)(base + 2048 + lane * 16) = hi2;
      *(h8*)(base + 3072 + lane * 16) = lo2;
      continue;
    }
    *(h4*)(base + 2048 + lane * 16) = hi1;  // [hi | lo] of the K = 16 block side by side: one 16-byte load per lane
    *(h4*)(base + 2048 + lane * 16 + 8) = lo1;
  }
}

// Wide layout: fragments of one head of one workgroup block.  grid (blocks, heads, basis variants).  Output per
// (wave w, head h): NG groups x 3 query tiles x [hi | lo] KiB; element (lane, e) of (group gi, tile jt) =
//   S[query token 48 w + 16 jt + (lane & 15)][key token 16 (K0_w + 2 gi) + 8 (lane >> 4) + e]
// with S the block-diagonal matrix of the block's molecules' normalised scores (zero outside a molecule / masked keys).
struct H3WideWin { int w[4]; };
__global__ void h3w_score_frag_kernel(const float* __restrict__ x, const uint8_t* __restrict__ masked,
                                      const float* __restrict__ ls, int H, int V, int P, int NG, int mpwg, int64_t n_rows,
                                      int64_t n_cond, int normalise, char* __restrict__ sfrag, ScoreBasis basis,
                                      int64_t variant_bytes, int use_mm, H3WideWin win, int nt) {
  extern __shared__ float sm[];
  const int MV = mpwg * V;
  float* xs = sm;                         // [MV][3]
  float* E = xs + MV * 3;                 // [MV][V]: basis values, then normalised scores, of this head
  uint8_t* msk = (uint8_t*)(E + MV * V);  // [MV]
  const int64_t blk = blockIdx.x;
  const int h = blockIdx.y;
  const int nthr = blockDim.x, t = threadIdx.x;
  for (int i = t; i < MV; i += nthr) {
    const int q = i / V, a = i - q * V;
    int64_t n = blk * mpwg + q;
    if (n >= n_rows) n = n_rows - 1;
    const int64_t src = (n % n_cond) * V + a;
    xs[3 * i] = x[3 * src];
    xs[3 * i + 1] = x[3 * src + 1];
    xs[3 * i + 2] = x[3 * src + 2];
    msk[i] = masked[src];
  }
  float cmean = 0.f;
  const float* cf = nullptr;
  if (basis.order > 0) cf = basis_coeffs(basis, blockIdx.z, h, &cmean);
  __syncthreads();
  const float lh = ls[h];
  for (int i = t; i < MV * V; i += nthr) {
    const int qa = i / V, m = i - qa * V;
    const int q = qa / V;
    const float dd = h3_pair_dist(xs + q * V * 3, qa - q * V, m, use_mm);
    E[i] = msk[q * V + m] ? 0.f : basis_value(dd / lh, cf, basis.order, cmean);
  }
  __syncthreads();
  for (int i = t; i < MV; i += nthr) {
    float* row = E + (int64_t)i * V;
    double sum = 0.0;
    for (int m = 0; m < V; ++m) sum += (double)fabsf(row[m]);
    const float den = (float)sum + 1e-5f;
    if (normalise)
      for (int m = 0; m < V; ++m) row[m] = row[m] / den;
  }
  __syncthreads();
  char* out = sfrag + blockIdx.z * variant_bytes + blk * (int64_t)4 * H * ((int64_t)NG * nt * 2048);
  const bool paired = nt == H3N4_NT;
  for (int i = t; i < 4 * NG * nt * 64; i += nthr) {
    const int lane = i & 63, rest = i >> 6;
    const int jt = rest % nt, gi = (rest / nt) % NG, w = rest / (nt * NG);
    const int tq = 16 * nt * w + 16 * jt + (lane & 15);
    const int mq = tq / P, aq = tq - mq * P;  // slot -> (molecule, atom); atoms >= V are the padding of a strided molecule
    const bool qok = mq < mpwg && aq < V;
    const float* row = E + (int64_t)(qok ? mq * V + aq : 0) * V;
    const int k0 = 16 * (win.w[w] / 32 + 2 * gi) + 8 * (lane >> 4);
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // paired layout: the keys of the molecule's 128 slots in the order its two waves' X^T images hold them (the 64-token
      // statement's: element e of lane group g of group gi = slot 32 gi + 4 g + e, e < 4; 32 gi + 16 + 4 g + e - 4, e >= 4)
      const int tk = paired ? mq * P + 32 * gi + 4 * (lane >> 4) + (e < 4 ? e : e + 12) : k0 + e;
      const int ak = tk - mq * P;
      const float val = (qok && ak >= 0 && ak < V) ? row[ak] : 0.f;
      const _Float16 hh = (_Float16)val;
      hi[e] = hh;
      lo[e] = (_Float16)(val - (float)hh);
    }
    // wide: [group][query tile]; paired: [query tile][group] (8 KiB per query tile, as the statement's loads walk them)
    char* base = out + ((int64_t)(w * H + h) * (NG * nt) + (paired ? jt * NG + gi : gi * nt + jt)) * 2048;
    *(h8*)(base + lane * 16) = hi;
    *(h8*)(base + 1024 + lane * 16) = lo;
  }
}

// ================================================================================================
// the kernel
// ================================================================================================
struct H3Params {
  const char* packed;  // net 0 of this coupling layer
  int64_t net_stride_bytes;
  int64_t stages;
  int64_t side_in2b, side_layers, side_layer_size, side_out2b, side_scales;
  const float* emb;
  const int32_t* types;
  const float* xc;
  const float* xv;
  const float* z_other;
  const char* sfrag;
  int sfrag_shared;
  int64_t sf_variant_bytes;  // chebyshev_kernel: bytes between the fragment sets of (net, layer) variants; else 0
  int windowed;              // block-diagonal mixing with per-tile key windows (h3_windowed); asm variant only
  const float* rff;          // dense model with position features: [3, d_rff / 2] Gaussian vectors of this coupling layer
  int d_rff;
  int win[4];                // wide layout: byte offset of each wave's key window in a row of the shared X^T tile
  float* out[2];
  float* dump;
  int64_t n_rows, n_cond;
  int V, mpw, nblocks;
  int ng;  // wide layout: key groups of a wave's window (H3Wide::ng: 5, or 3 -> the tw_h?_attns3 statement)
  int P;  // token slots per molecule in the block: V, or the wide layout's padded stride (H3Wide::stride)
  int H, n_layers, ff_chunks, hid_chunks, d_emb;
  float eps;
  int net_sel;
  int debug;  // timing experiments only: bit 0 = no weight DMA after the prologue, bit 1 = no barriers
  const uint8_t* masked;
  PrevCoupling prev;  // the previous coupling layer's update, applied here (flow_pass_h3)
};
// 64-token waves: four query tiles x [k-step 0 hi | lo | k-step 1 hi | lo], 16 B per lane each
__device__ __forceinline__ void h3_load_sf4(const char* p, u4 (&r)[16]) {
  asm volatile(
      "global_load_dwordx4 %0, %16, off\n\t"
      "global_load_dwordx4 %1, %16, off offset:1024\n\t"
      "global_load_dwordx4 %2, %16, off offset:2048\n\t"
      "global_load_dwordx4 %3, %16, off offset:3072\n\t"
      "global_load_dwordx4 %4, %17, off\n\t"
      "global_load_dwordx4 %5, %17, off offset:1024\n\t"
      "global_load_dwordx4 %6, %17, off offset:2048\n\t"
      "global_load_dwordx4 %7, %17, off offset:3072\n\t"
      "global_load_dwordx4 %8, %18, off\n\t"
      "global_load_dwordx4 %9, %18, off offset:1024\n\t"
      "global_load_dwordx4 %10, %18, off offset:2048\n\t"
      "global_load_dwordx4 %11, %18, off offset:3072\n\t"
      "global_load_dwordx4 %12, %19, off\n\t"
      "global_load_dwordx4 %13, %19, off offset:1024\n\t"
      "global_load_dwordx4 %14, %19, off offset:2048\n\t"
      "global_load_dwordx4 %15, %19, off offset:3072\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
        "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
      : "v"(p), "v"(p + H3N4_SF_BYTES), "v"(p + 2 * H3N4_SF_BYTES), "v"(p + 3 * H3N4_SF_BYTES)
      : "memory");
}

template <int NT>
struct BOp {  // B operand of one 32-deep k-step for NT token tiles, split fp16
  h8 h[NT], l[NT];
};

__device__ __forceinline__ f4 mfma32(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 mfma16(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

// acc[jt] += (ah + al) x (b.h + b.l) without the lo x lo term
template <int NT>
__device__ __forceinline__ void mma3(const h8 ah, const h8 al, const BOp<NT>& b, f4 (&acc)[NT]) {
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) acc[jt] = mfma32(ah, b.h[jt], acc[jt]);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) acc[jt] = mfma32(ah, b.l[jt], acc[jt]);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) acc[jt] = mfma32(al, b.h[jt], acc[jt]);
}

// fp32 pair -> packed fp16 hi pair + packed fp16 lo pair (lo = fp16(v - hi)), four VALU ops: v_cvt_pk_f16_f32 (RNE),
// two v_fma_mix_f32 (hi as an fp16 source: no conversion back) and a second pack - the sequence the generated asm
// sections use.  hipcc's translation of the plain C++ form costs eight (cvt, cvt back, sub, and hi packed a second
// time); the values are the same (v - hi is exact in fp32 either way).
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  float ta, tb;
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_cvt_pk_f16_f32 %1, %2, %3"
      : "=&v"(hi), "=&v"(lo), "=&v"(ta), "=&v"(tb)
      : "v"(a), "v"(b));
}

// two D tiles (features 16t..16t+15, 16t+16..16t+31 of one token) -> split B-operand element vector
__device__ __forceinline__ void split8(const f4 a, const f4 b, h8& hi, h8& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split_pair(a[0], a[1], h0, l0);
  split_pair(a[2], a[3], h1, l1);
  split_pair(b[0], b[1], h2, l2);
  split_pair(b[2], b[3], h3, l3);
  hi = __builtin_bit_cast(h8, (u4){h0, h1, h2, h3});
  lo = __builtin_bit_cast(h8, (u4){l0, l1, l2, l3});
}

// one D tile -> split K = 16 operand
__device__ __forceinline__ void split4(const f4 a, h4& hi, h4& lo) {
  unsigned h0, h1, l0, l1;
  split_pair(a[0], a[1], h0, l0);
  split_pair(a[2], a[3], h1, l1);
  hi = __builtin_bit_cast(h4, (u2){h0, h1});
  lo = __builtin_bit_cast(h4, (u2){l0, l1});
}

// max / sum over the four lanes that share (lane & 15): v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16- and
// 32-lane blocks between two registers without touching the LDS:
//   v_permlane16_swap a, b:  a' = [a.r0, b.r0, a.r2, b.r2],  b' = [a.r1, b.r1, a.r3, b.r3]   (rows of 16 lanes)
//   v_permlane32_swap a, b:  a' = [a.lo32, b.lo32],          b' = [a.hi32, b.hi32]
// so with two copies of one value every lane ends up holding its own and its partner's.  Inline asm: hipcc (ROCm 7.2)
// folds the SECOND result of __builtin_amdgcn_permlane{16,32}_swap into the first when both inputs are the same value
// (tools/probe/permlane_swap_probe.hip); the s_nop covers the VALU-write -> permlane-swap hazard the compiler would pad.
__device__ __forceinline__ void h3_swap16(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void h3_swap32(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float h3_quad_max(float v) {
  float a = v, b = v;
  h3_swap16(a, b);
  a = b = fmaxf(a, b);
  h3_swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float h3_quad_sum(float v) {
  float a = v, b = v;
  h3_swap16(a, b);
  a = b = a + b;
  h3_swap32(a, b);
  return a + b;
}

// sum over the four lanes that share (lane & 15): same association as the former __shfl_xor(16) / __shfl_xor(32) pair
// ((a + b) in both partners, then the two pair sums), so the LayerNorms are bit-identical - without the two LDS round trips
__device__ __forceinline__ float h3_xor_sum(float v) { return h3_quad_sum(v); }

template <int NT, int KS>
__device__ __forceinline__ void to_bop(const f4 (&x)[2 * KS][NT], BOp<NT> (&b)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) split8(x[2 * ks][jt], x[2 * ks + 1][jt], b[ks].h[jt], b[ks].l[jt]);
}

// hipcc's s_waitcnt insertion cannot be trusted for ordinary global loads issued while YOUNGER LDS-DMAs join the queue
// (a missing wait was observed once control flow was added around the score-fragment loads).  Global loads inside the
// pipelined part of the kernel are therefore issued from inline asm together with their own s_waitcnt vmcnt(0) in ONE
// statement (early-clobber outputs): the compiler neither counts nor schedules around them, and the wait also drains
// the LDS-DMA queue.  (Loads in the prologue are plain C++: the only LDS-DMAs in flight there are older than they are.)
// score fragments of one head for three token tiles (H3_SF_BYTES apart): 16-B and 8-B parts
__device__ __forceinline__ void h3_load_sf3(const char* p16, const char* p8, u4 (&a)[3], u4 (&b)[3], u2 (&c)[3], u2 (&d)[3]) {
  asm volatile(
      "global_load_dwordx4 %0, %12, off\n\t"
      "global_load_dwordx4 %1, %12, off offset:1024\n\t"
      "global_load_dwordx2 %2, %13, off offset:2048\n\t"
      "global_load_dwordx2 %3, %13, off offset:2056\n\t"
      "global_load_dwordx4 %4, %14, off\n\t"
      "global_load_dwordx4 %5, %14, off offset:1024\n\t"
      "global_load_dwordx2 %6, %15, off offset:2048\n\t"
      "global_load_dwordx2 %7, %15, off offset:2056\n\t"
      "global_load_dwordx4 %8, %16, off\n\t"
      "global_load_dwordx4 %9, %16, off offset:1024\n\t"
      "global_load_dwordx2 %10, %17, off offset:2048\n\t"
      "global_load_dwordx2 %11, %17, off offset:2056\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a[0]), "=&v"(b[0]), "=&v"(c[0]), "=&v"(d[0]), "=&v"(a[1]), "=&v"(b[1]), "=&v"(c[1]), "=&v"(d[1]),
        "=&v"(a[2]), "=&v"(b[2]), "=&v"(c[2]), "=&v"(d[2])
      : "v"(p16), "v"(p8), "v"(p16 + H3_SF_BYTES), "v"(p8 + H3_SF_BYTES), "v"(p16 + 2 * H3_SF_BYTES),
        "v"(p8 + 2 * H3_SF_BYTES)
      : "memory");
}



template <int NT>
__device__ __forceinline__ void h3_add_layernorm(f4 (&x)[8][NT], const f4 (&y)[8][NT], const float* lnw_lane,
                                                 const float* lnb_lane, float eps) {
  // lnw_lane / lnb_lane point into the layer's side block in LDS (staged there by LDS-DMA at the top of the layer)
  f4 w[8], b[8];
#pragma unroll
  for (int ft = 0; ft < 8; ++ft) {
    w[ft] = *(const f4*)(lnw_lane + 16 * ft);
    b[ft] = *(const f4*)(lnb_lane + 16 * ft);
  }
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      x[ft][jt] = x[ft][jt] + y[ft][jt];
      s += (x[ft][jt][0] + x[ft][jt][1]) + (x[ft][jt][2] + x[ft][jt][3]);
    }
    const float mean = h3_xor_sum(s) * (1.f / 128.f);
    float q = 0.f;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      x[ft][jt] = x[ft][jt] - mean;
      q += (x[ft][jt][0] * x[ft][jt][0] + x[ft][jt][1] * x[ft][jt][1]) +
           (x[ft][jt][2] * x[ft][jt][2] + x[ft][jt][3] * x[ft][jt][3]);
    }
    const float var = h3_xor_sum(q) * (1.f / 128.f);
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) x[ft][jt] = x[ft][jt] * rstd * w[ft] + b[ft];
  }
}

// The weight pipeline: a ring of H3_RING stage buffers in LDS, refilled by LDS-DMA.  `advance`:
//   1. s_waitcnt vmcnt(0): every LDS-DMA this wave has issued has landed (stated in asm: hipcc neither
//      counts LDS-DMA nor reliably waits for it).  This C++ hand-off simply drains the queue; the ring still
//      keeps RING-1 stages of prefetch distance because the drain happens one full stage of compute after
//      the youngest fetch was issued.  (Counted waits, vmcnt(6) = three stages left in flight, are what the
//      generated asm sections use; here they measured correct but not faster.)
//   2. raw s_barrier: every wave has finished reading the current stage and all shares have landed;
//   3. refill the buffer just released with stage s+RING.
// Each wave moves 2 KiB of every stage; wave 0 additionally moves the 1 KiB aux block.
struct H3Pipe {
  const char* gnext;  // global address of the next stage to fetch (this lane's 16 B)
  char* lds;          // ring base
  int cur;            // ring slot of the current stage
  int wave;
  int debug;
  int ring = H3_RING;  // stage buffers (H3N4_RING for the 64-token build)

  __device__ __forceinline__ void fetch(int slot) {
    char* dst = lds + slot * H3_STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gnext + wave * 2048 + i * 1024),
                                       (__attribute__((address_space(3))) void*)(dst + wave * 2048 + i * 1024), 16, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gnext + H3_STAGE_TILE_BYTES),
                                       (__attribute__((address_space(3))) void*)(dst + H3_STAGE_TILE_BYTES), 16, 0, 0);
    gnext += H3_STAGE_BYTES;
  }
  // The first H3_RING stages are requested at the very top of the kernel and waited for only when the in-MLP is about to
  // read them: their latency (L2 / Infinity Cache, ~1-2 us) runs under the prologue's own dependent loads (token
  // bookkeeping, embedding gather, the previous coupling layer's update) instead of behind them.  The compiler's vmcnt
  // waits for those loads stay correct with LDS-DMAs OLDER than the loads in the queue (returns are in order; the waits
  // only become conservative).
  __device__ __forceinline__ void start_issue() {
#pragma unroll
    for (int i = 0; i < H3_RING; ++i)
      if (i < ring) fetch(i);
    cur = 0;
  }
  __device__ __forceinline__ void start_wait() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  __device__ __forceinline__ const char* stage() const { return lds + cur * H3_STAGE_BYTES; }
  __device__ __forceinline__ void advance() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!TW_EXPERIMENT(debug & 2)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int released = cur;
    cur = (cur + 1 == ring) ? 0 : cur + 1;
    if (!TW_EXPERIMENT(debug & 1)) fetch(released);
  }
};

// All tile pairs of the current stage, read up front into distinct registers (hipcc otherwise reuses
// one register quad for every read and exposes the LDS latency eight times per stage); the
// sched_barrier keeps the reads ahead of the MFMAs that consume them.
struct H3Tiles {
  // Rolling prefetch of the stage's four tile pairs: pair p+2 is read while pair p is consumed.
  // hipcc's default is read -> wait -> use per pair, which exposes the LDS latency eight times per
  // stage; reading all eight at once is slower still (four waves burst-read the same 8 KiB and queue
  // on the 128 B/clk LDS port).  sched_barrier(0) pins the source order: pair p's MFMAs may not sink
  // below, nor pair p+2's reads rise above, the fence between them.
  h8 t[2 * H3_STAGE_PAIRS];
  const char* base;
  __device__ __forceinline__ void rd(int p) {
    t[2 * p] = *(const h8*)(base + p * H3_PAIR_BYTES);
    t[2 * p + 1] = *(const h8*)(base + p * H3_PAIR_BYTES + 1024);
  }
  __device__ __forceinline__ void load(const char* st, int lane) {
    base = st + lane * 16;
    rd(0);
    rd(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // call before consuming pair p
  __device__ __forceinline__ void ready(int p) {
    if (p + 2 < H3_STAGE_PAIRS) rd(p + 2);
  }
  // call after consuming pair p
  __device__ __forceinline__ void done(int) { __builtin_amdgcn_sched_barrier(0); }
  __device__ __forceinline__ h8 hi(int p) const { return t[2 * p]; }
  __device__ __forceinline__ h8 lo(int p) const { return t[2 * p + 1]; }
};
// chained MLP stage:  y[OT_OUT] += W2 . act(sc0 * (W0 . xin) + b0), 32 hidden units per chunk.
// A stages (W0 chunk -> hidden pre-activations, aux block on the first) and B stages (W2 chunk) of
// consecutive chunks are software pipelined:  A(0) | A(1) B(0) | A(2) B(1) | ... | B(n-1).  The
// epilogue of chunk c+1 (scale, bias, activation, fp16 hi/lo split: ~27 VALU per 4 values) has no
// dependence on the B(c) MFMAs, so it is placed inside B(c)'s stages, one 4-value unit per tile pair,
// where it issues in the shadow of the MFMAs instead of between two MFMA bursts (PMC: 22% of the wave's
// cycles were VALU issue with the matrix pipe idle).  h3_pack_weights emits the stages in this order.
template <int NT, int KS_IN, int OT_OUT, bool SILU>
__device__ __forceinline__ void h3_mlp_chain(const BOp<NT> (&xin)[KS_IN], f4 (&yacc)[OT_OUT][NT], H3Pipe& pipe,
                                             int n_chunks, int lane) {
  const int g = lane >> 4;
  // the 2 KS_IN tile pairs of a chunk's first GEMM, pair q = (o = q / KS_IN, ks = q % KS_IN), four to a stage:
  // KS_IN = 2: one stage, 4: two, 6 (dense model with RFF position features, 192 input columns): three
  static_assert((2 * KS_IN) % H3_STAGE_PAIRS == 0, "whole stages");
  constexpr int A_STAGES = 2 * KS_IN / H3_STAGE_PAIRS;
  constexpr int B_STAGES = (OT_OUT + 3) / 4;
  constexpr int UNITS = 2 * NT;  // epilogue units: (o, jt), four hidden values of one token column each
  f4 hacc[2][NT];
  f4 bias[2];
  float sc = 1.f;

  auto a_stages = [&]() {
#pragma unroll
    for (int a = 0; a < A_STAGES; ++a) {
      const char* st = pipe.stage();
      if (a == 0) {
        const float* aux = (const float*)(st + H3_STAGE_TILE_BYTES);
        sc = aux[32];
        bias[0] = *(const f4*)(aux + 4 * g);
        bias[1] = *(const f4*)(aux + 16 + 4 * g);
      }
      H3Tiles w;
      w.load(st, lane);
#pragma unroll
      for (int pr = 0; pr < H3_STAGE_PAIRS; ++pr) {
        const int q = a * H3_STAGE_PAIRS + pr, o = q / KS_IN, ks = q % KS_IN;
        if (ks == 0) {
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) hacc[o][jt] = (f4){0.f, 0.f, 0.f, 0.f};
        }
        w.ready(pr);
        mma3<NT>(w.hi(pr), w.lo(pr), xin[ks], hacc[o]);
        w.done(pr);
      }
      pipe.advance();
    }
  };
  // unit u = (o, jt): activation of hacc[o][jt] -> elements 4o..4o+3 of the split B operand.  The two empty
  // asm statements pin the unit between the scheduling fences of the tile pair it is issued with (pure
  // arithmetic is otherwise free to be linearised anywhere in the block, i.e. in front of all MFMAs).
  auto epi_unit = [&](int u, BOp<NT>& dst, bool pin) {
    const int o = u / NT, jt = u % NT;
    f4 in = hacc[o][jt];
    if (pin) asm volatile("" : "+v"(in));
    h4 hi4, lo4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pre = fmaf(in[r], sc, bias[o][r]);
      // SiLU with the hardware exp2 / rcp (1 ulp each; |pre| * 2^-24 argument rounding): five VALU ops per value
      // instead of ~25 for expf + IEEE division - the in/out MLP epilogues were 70 % of those stages' time
      const float v = SILU ? pre * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * pre))
                           : fmaxf(pre, 0.f);
      const _Float16 hi = (_Float16)v;
      hi4[r] = hi;
      lo4[r] = (_Float16)(v - (float)hi);
    }
    if (pin) asm volatile("" : "+v"(hi4), "+v"(lo4));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dst.h[jt][4 * o + r] = hi4[r];
      dst.l[jt][4 * o + r] = lo4[r];
    }
  };
  auto b_stages = [&](const BOp<NT>& cur, bool with_epi, BOp<NT>& nxt) {
#pragma unroll
    for (int b = 0; b < B_STAGES; ++b) {
      const char* st = pipe.stage();
      H3Tiles w;
      w.load(st, lane);
#pragma unroll
      for (int oo = 0; oo < 4; ++oo) {
        const int ot = 4 * b + oo;
        w.ready(oo);
        if (with_epi) {
          if (B_STAGES == 1) {
            if (oo == 0)
#pragma unroll
              for (int u = 0; u < UNITS; ++u) epi_unit(u, nxt, false);
          } else {
            const int u = 4 * b + oo;
            if (u < UNITS) epi_unit(u, nxt, true);
          }
        }
        if (ot < OT_OUT) mma3<NT>(w.hi(oo), w.lo(oo), cur, yacc[ot]);
        if (with_epi && B_STAGES > 1 && 4 * b + oo < UNITS) {
          // issue order inside this pair's region: one MFMA, then two of the unit's VALU ops in its shadow
          // (tools/probe/mfma_valu_overlap.hip: two VALU ops per K=32 MFMA issue for free, the third costs)
#pragma unroll
          for (int i = 0; i < 3 * NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
        }
        w.done(oo);
      }
      pipe.advance();
    }
  };

  BOp<NT> hb;
  a_stages();
#pragma unroll
  for (int u = 0; u < UNITS; ++u) epi_unit(u, hb, false);
  for (int c = 0; c + 1 < n_chunks; ++c) {
    a_stages();  // chunk c+1
    BOp<NT> nxt;
    b_stages(hb, true, nxt);  // chunk c, with the epilogue of chunk c+1 in its shadow
    hb = nxt;
  }
  BOp<NT> unused;
  b_stages(hb, false, unused);
}

// ASM = true: attention and FFN run in the generated asm blocks (the product path).  ASM = false: the same two
// sections as compiled C++ (tw_debug_set_flags bit 3; kept as the readable statement of what the asm computes and
// for A/B checks).  Two instantiations rather than a runtime branch: with both variants in one function the
// register allocator spilled 88 VGPRs to scratch.
// DENSE = true: the dense softmax variant (transformer_nvp, nn.TransformerEncoderLayer post-norm, 8 heads of 16;
// transformer_block.py:18-72).  Same in / FFN / out sections and weight pipeline; the attention block is
//   per head h:  q_h, k_h = W_{q,k}[16 h ..] . x^T  (standard orientation: D tile = [feature][token], which IS the
//                K = 16 MFMA operand layout: k_h as A, q_h as B);  v_h with the operands swapped,
//                D = x . W_v^T = [token][feature]: lane = feature, registers = tokens - the A operand of P.V;
//                S^T[key][query] = k_h q_h^T / 4 on K = 16 MFMAs (3-term split);  masked softmax over the keys of the
//                query's molecule in registers (the four lanes of a query meet through v_permlane swaps);
//                O^T[d][query] = V^T P^T on K = 32 (key tiles 0, 1) + K = 16 (tile 2) MFMAs -> standard orientation;
//   per head pair: the two O tiles are one 32-deep k-step of out_proj:  y += W_o[:, 32 hp ..] . O
// Nothing is transposed through memory and nothing leaves the registers.
// WIDE = true (kernel attention, asm sections only): molecules of 25 .. 192 atoms, packed over the workgroup's 192 token slots
// (back to back, or at a slot stride of 96); token-local sections unchanged, attention through the shared X^T tile
// (tw_h3_attns{,3,6}_asm.inc).  WIDE with NT = 4 (r05, ENC only): the PAIRED layout - one molecule of 97 .. 128 atoms per pair of
// 64-token waves, no shared tile: a wave mixes against its own and its partner's X^T images (tw_h?n4p_enc_asm.inc).
// ENC = true: the whole encoder stack is ONE generated asm statement (tools/gen_h3_enc_asm.py) - the product build of every
// family since r05 (ScratchSize 0); ENC = false: the per-section build (asm sections, compiled glue), kept behind
// tw_debug_set_flags bit 12 as the A/B reference and for activation dumps / section stamps between the sections.
// RFF = true (dense only): 128 random Fourier features of the conditioning positions appended to the in-MLP's input
// (transformer_nvp_posenc.yaml); six input k-steps, the in-MLP as compiled C++ (the asm section takes two).
// H1 = true (encoder-stack build only): the single-MFMA "fast" variant, TW_PATH_FUSED_H1 - one half-precision MFMA per
// product (fp16 hi halves only), the weight stream of h3_geom(d, true); tools/gen_h3_*_asm.py --h1 -> tw_h1_*_asm.inc.
// Not a parity path: operands carry 11 significand bits.
template <int NT, bool ASM, bool DENSE = false, bool WIDE = false, bool RFF = false, bool ENC = false, bool H1 = false, bool NG6 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
netblock_h3_kernel(const H3Params p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  static_assert(!H1 || ENC || WIDE || (DENSE && ASM && !RFF) || (NT == 4 && ASM),
                "the single-MFMA variant exists as the encoder-stack build (<= 48 atoms), for the wide layout, and - MLP sections "
                "only, the softmax attention block stays in split form - for the dense model");
  static_assert(!WIDE || (ASM && !DENSE), "the wide layout exists for the asm build of the kernel-attention variant");
  // NG6: the wide layout's six-group statement (161 .. 192 atoms) as an instantiation of its own - its clobber list reaches v239,
  // and inside one kernel with the three- / five-group statements it cost every wide launch 30 more spilled registers
  static_assert(!NG6 || WIDE, "six-group windows belong to the wide layout");
  static_assert(!RFF || DENSE, "position features belong to the dense model");
  static_assert(!ENC || ASM, "the encoder-stack statement embeds the generated asm sections");
  static_assert(NT == 3 || (NT == 4 && !RFF && (!WIDE || (ENC && !NG6)) && (!DENSE || (ASM && !WIDE && !H1))),
                "64-token waves: the kernel-attention variant - one molecule of 49-64 atoms per wave, or (WIDE: the paired layout, "
                "encoder-stack statement only) one of 97-128 atoms per pair of waves; the dense softmax model on one molecule of "
                "49-64 atoms per wave: the encoder-stack statement (r06, tools/gen_h3_enc_asm.py --dense --nt=4), or the per-section "
                "build with its attention block compiled C++ (r05; activation dumps, section stamps)");
  constexpr int KIN = RFF ? 6 : 2;  // 32-column k-steps of the in-MLP's input
  // 64-token build: all four GEMM sections are generated asm (tools/gen_h3_ffn_asm.py / gen_h3_attn_asm.py --nt=4), the glue
  // between them compiled C++ (the per-section build)
  // (dense model at 64 tokens: the generated softmax block exists for 48 tokens only - r03 measured the compiled block within
  // 0.4 % of it in wall time - so that instantiation runs the MLP sections as asm and the attention block as compiled C++)
  constexpr bool ASM_IO = ASM, ASM_ATT = ASM && !(DENSE && NT == 4);
  // Six stage buffers and one barrier per PAIR of FFN stages (tools/gen_h3_ffn_asm.py --ring6) where the LDS has the 9 KiB:
  // the 48-token kernel-attention layout.  Built for the fast mode's encoder-stack build, whose 24-MFMA stages pay most per
  // hand-off.  The stream, the five stages requested ahead and the prologue are the five-slot ring's.
  constexpr bool R6 = H3_R6(NT, DENSE, WIDE, RFF, ENC, H1);
  constexpr int RING = NT == 4 ? H3N4_RING : (R6 ? H3_RING + 1 : H3_RING);
  constexpr int XT_IMG = NT == 4 ? H3N4_XT_IMG : H3_XT_IMG;
  constexpr int SF_BYTES = NT == 4 ? H3N4_SF_BYTES : H3_SF_BYTES;
  constexpr int WAVE_LDS = NT == 4 ? H3N4_WAVE_LDS : (DENSE ? H3D_WAVE_LDS : (WIDE ? H3W_WAVE_LDS : H3_WAVE_LDS));
  constexpr int SIDE_LDS_OFFSET = NT == 4 ? H3N4_SIDE_LDS_OFFSET
                                          : (DENSE ? H3D_SIDE_LDS_OFFSET : (WIDE ? H3W_SIDE_LDS_OFFSET
                                                                                  : H3_SIDE_LDS_OFFSET + (R6 ? H3_STAGE_BYTES : 0)));
  constexpr int SIDE_CHUNKS = DENSE ? 5 : 3;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i16 = lane & 15;

  int net, wg;
  if (p.net_sel < 0) {
    const int xcd = blockIdx.x & 7;
    net = xcd >> 2;
    wg = (blockIdx.x >> 3) * 4 + (xcd & 3);
  } else {
    net = p.net_sel;
    wg = blockIdx.x;
  }
  // every wave of the workgroup takes part in the weight pipeline, even if it owns no rows
  // narrow layouts: wave-block blk holds p.mpw whole molecules; wide: workgroup wg holds p.mpw molecules over its 192 slots
  const int blk = wg * 4 + wave;
  const bool active = WIDE ? true : blk < p.nblocks;
  if (WIDE ? wg >= p.nblocks : wg * 4 >= p.nblocks) return;  // whole workgroup idle (uniform)
  const int64_t row0 = (WIDE ? (int64_t)wg : (int64_t)blk) * p.mpw;  // first conformation of this wave's / workgroup's block
  const int slot0 = WIDE ? 16 * NT * wave : 0;                       // this wave's first token slot in the block

  const char* net_base = p.packed + (int64_t)net * p.net_stride_bytes;
  const float* side = (const float*)(net_base + p.stages * H3_STAGE_BYTES);
  const float* scales = side + p.side_scales;
  _Float16* xt_hi = (_Float16*)(lds + RING * H3_STAGE_BYTES + wave * WAVE_LDS);

  const unsigned long long t_kernel_top = (p.debug & 16) ? __builtin_readcyclecounter() : 0ull;  // section profile only
  // ---- request the first stages of the weight stream ----
  H3Pipe pipe;
  pipe.gnext = net_base + lane * 16;
  pipe.lds = lds;
  pipe.cur = 0;
  pipe.wave = wave;
  pipe.debug = p.debug;
  pipe.ring = RING;
  if constexpr (ENC && DENSE && NT == 3) {   // (64-token waves: single-buffered, the statement fetches every layer's block itself)
    // The dense encoder-stack statement double-buffers the layers' side blocks in the LDS (layer l in buffer l % 2, fetched
    // a whole layer ahead: its attention block reads biases and the in_proj scale at its very entry).  Layer 0's goes first,
    // older than every weight stage: start_wait()'s vmcnt(0) + barrier cover it.
    if (wave == 0) {
      const char* src = (const char*)(side + p.side_layers) + lane * 16;
      char* dst = lds + SIDE_LDS_OFFSET;
#pragma unroll
      for (int i = 0; i < SIDE_CHUNKS; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    }
  }
  pipe.start_issue();

  // Second-layer bias and output scale of the in and out MLPs: plain loads at the very top, so that they are back long
  // before the first use (behind the asm statements each of them was a global round trip of its own, with the pipe idle)
  const float sc_in2 = scales[1], sc_out2 = scales[2 + 3 * p.n_layers + 1];
  f4 bb_in2[8];
#pragma unroll
  for (int ot = 0; ot < 8; ++ot) bb_in2[ot] = *(const f4*)(side + p.side_in2b + 4 * g + 16 * ot);
  const f4 bb_out2 = *(const f4*)(side + p.side_out2b + 4 * g);

  // ---- token bookkeeping and input features (ordinary loads, compiler-scheduled: the only LDS-DMAs in flight are older) ----
  int64_t tok_row[NT];
  int64_t tok_cond[NT];  // the conditioning state (row of atom_types / x / masked) of the token's conformation: row % n_cond
  int tok_atom[NT];
  // token slot -> (molecule, atom) without an integer division per token: floor(t / V) = (t * ceil(2^16 / V)) >> 16 for
  // t < 192, V <= 160 (t * (ceil(2^16 / V) * V - 2^16) < 2^16); the row's conditioning state without a 64-bit modulo in the
  // two layouts the flow uses (one shared state: the reverse pass of an MH iteration; one per row: the forward pass)
  const unsigned inv_v = (65536u + (unsigned)p.P - 1u) / (unsigned)p.P;
  const bool cond_shared = p.n_cond == 1, cond_per_row = p.n_cond >= p.n_rows;
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int t = slot0 + 16 * jt + i16;
    const int q = (int)(((unsigned)t * inv_v) >> 16);
    const int64_t n = row0 + q;
    const int atom = t - q * p.P;
    const bool ok = active && q < p.mpw && n < p.n_rows && atom < p.V;  // (atom >= V: padding slots of a strided molecule)
    tok_row[jt] = ok ? n : -1;
    tok_cond[jt] = !ok || cond_shared ? 0 : (cond_per_row ? n : n % p.n_cond);
    tok_atom[jt] = ok ? atom : 0;
  }
  // z_other of this lane's tokens: from memory, or - when the previous coupling layer's update is still pending - that
  // update applied on the fly (PrevCoupling, tw_common.h; same arithmetic as coupling_kernel)
  float zo[NT][3];
  {
    float part[NT];
    bool bad = false;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int64_t n = tok_row[jt];
      part[jt] = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) zo[jt][k] = 0.f;
      if (n < 0) continue;
      const int64_t idx = (n * p.V + tok_atom[jt]) * 3;
      if (p.prev.s_raw) {
        float ld = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float sr = p.prev.s_raw[idx + k], shift = p.prev.t[idx + k];
          const float scale = expf(sr);
          bad |= !(isfinite(sr) && isfinite(shift));
          const float z = p.prev.z_in[idx + k];
          zo[jt][k] = p.prev.reverse ? (z - shift) / scale : z * scale + shift;
          ld += logf(scale);
        }
        part[jt] = p.masked[tok_cond[jt] * p.V + tok_atom[jt]] ? 0.f : ld;
        if (net == 0 && g == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) p.prev.z_out[idx + k] = zo[jt][k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) zo[jt][k] = p.z_other[idx + k];
      }
    }
    if constexpr (WIDE) {
      // molecules span waves: the tokens' partial sums meet in a workgroup-shared LDS array (192 floats at the start of
      // the wave-private blocks, which nothing uses yet); wave 0 adds them up in atom order.  Both nets take the barriers.
      if (p.prev.s_raw) {
        float* scr = (float*)(lds + RING * H3_STAGE_BYTES);
        if (net == 0) {
          if (bad) atomicOr(p.prev.nonfinite, 1);
          if (g == 0) {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) scr[slot0 + 16 * jt + i16] = part[jt];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (net == 0 && wave == 0 && lane < p.mpw) {
          const int64_t n = row0 + lane;
          if (n < p.n_rows) {
            float acc = 0.f;
            for (int a = 0; a < p.V; ++a) acc += scr[lane * p.P + a];
            p.prev.delta_logp[n] -= p.prev.reverse ? -acc : acc;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else
    if (p.prev.s_raw && net == 0) {
      // log-determinant per conformation: the tokens' partial sums through the (still unused) wave-private LDS block,
      // added up in atom order by one lane per molecule; delta_logp -= logdet (nvp.py:86)
      if (bad) atomicOr(p.prev.nonfinite, 1);
      float* scr = (float*)xt_hi;
      if (g == 0) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) scr[16 * jt + i16] = part[jt];
      }
      if (lane < p.mpw) {
        const int64_t n = (int64_t)blk * p.mpw + lane;
        if (active && n < p.n_rows) {
          float acc = 0.f;
          for (int a = 0; a < p.V; ++a) acc += scr[lane * p.V + a];
          p.prev.delta_logp[n] -= p.prev.reverse ? -acc : acc;
        }
      }
    }
  }
  const unsigned long long t_pro1 = (p.debug & 16) ? __builtin_readcyclecounter() : 0ull;
  // u in B-operand element order: k-step ks, element e  <->  feature 32 ks + 16 (e/4) + 4 g + e%4
  BOp<NT> u[KIN];
  const bool std_input = !RFF && p.d_emb == 32;  // (uniform) 32 embedding columns + 9 of the second k-step's 32
  if (std_input) {
    // The configured shape (atom_embedding_dim 32): lane group g holds embedding columns 4 g.., 16 + 4 g.. - two 16-byte
    // loads of the row - and columns 32 + 4 g.. = [xc 3 | xv 3 | z 3 | 0...] picked by g; columns 48.. are padding.  The
    // general loop below walks the 16 elements through runtime comparisons and a dynamically indexed register array: 27 k
    // cycles of prologue per launch (tools/profile_h3_sections.py) were mostly that.
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int64_t n = tok_row[jt];
      const int64_t ca = n < 0 ? 0 : tok_cond[jt] * p.V + tok_atom[jt];
      const int ty = n < 0 ? 0 : p.types[ca];
      const float* er = p.emb + ty * 32 + 4 * g;
      f4 e0, e1, m;
#pragma unroll
      for (int r = 0; r < 4; ++r) { e0[r] = er[r]; e1[r] = er[16 + r]; }
      const float c0 = p.xc[ca * 3], c1 = p.xc[ca * 3 + 1], c2 = p.xc[ca * 3 + 2];
      const float v0 = p.xv[ca * 3], v1 = p.xv[ca * 3 + 1], v2 = p.xv[ca * 3 + 2];
      m[0] = g == 0 ? c0 : g == 1 ? v1 : g == 2 ? zo[jt][2] : 0.f;
      m[1] = g == 0 ? c1 : g == 1 ? v2 : 0.f;
      m[2] = g == 0 ? c2 : g == 1 ? zo[jt][0] : 0.f;
      m[3] = g == 0 ? v0 : g == 1 ? zo[jt][1] : 0.f;
      if (n < 0) e0 = e1 = m = (f4){0.f, 0.f, 0.f, 0.f};
      split8(e0, e1, u[0].h[jt], u[0].l[jt]);
      split8(m, (f4){0.f, 0.f, 0.f, 0.f}, u[1].h[jt], u[1].l[jt]);
    }
  } else
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int64_t n = tok_row[jt];
    const int64_t c = tok_cond[jt];
    const int a = tok_atom[jt];
    const int ty = n < 0 ? 0 : p.types[c * p.V + a];
#pragma unroll
    for (int ks = 0; ks < KIN; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int f = 32 * ks + 16 * (e / 4) + 4 * g + (e % 4);
        float val = 0.f;
        if (n >= 0) {
          if (f < p.d_emb) val = p.emb[ty * p.d_emb + f];
          else if (f < p.d_emb + 3) val = p.xc[(c * p.V + a) * 3 + (f - p.d_emb)];
          else if (f < p.d_emb + 6) val = p.xv[(c * p.V + a) * 3 + (f - p.d_emb - 3)];
          else if (f < p.d_emb + 9) val = zo[jt][f - p.d_emb - 6];
          else if (RFF && f < p.d_emb + 9 + p.d_rff) {
            // rff_position_encoder.py:57-62: sqrt(1/n) * [cos(x G), sin(x G)]; same arithmetic as build_input_kernel
            const float* px = p.xc + (c * p.V + a) * 3;
            const int nvec = p.d_rff / 2;
            const int j = f - p.d_emb - 9;
            const int col = j % nvec;
            const float ip = px[0] * p.rff[0 * nvec + col] + px[1] * p.rff[1 * nvec + col] + px[2] * p.rff[2 * nvec + col];
            val = sqrtf(1.0f / nvec) * (j < nvec ? cosf(ip) : sinf(ip));
          }
        }
        const _Float16 hi = (_Float16)val;
        u[ks].h[jt][e] = hi;
        u[ks].l[jt][e] = (_Float16)(val - (float)hi);
      }
  }
  asm volatile("" : "+v"(u[0].h[0]), "+v"(u[1].h[0]));
  const unsigned long long t_pro2 = (p.debug & 16) ? __builtin_readcyclecounter() : 0ull;
  // zero the transposed tile once: its pad columns are multiplied by zero scores and must be finite (the encoder-stack
  // statement keeps the transposed copy in registers and writes all of it)
  if constexpr (!ENC)
    for (int i = lane; i < WAVE_LDS / 16; i += 64) ((f4*)xt_hi)[i] = (f4){0.f, 0.f, 0.f, 0.f};

  const unsigned long long t_pro3 = (p.debug & 16) ? __builtin_readcyclecounter() : 0ull;
  // ---- the weight pipeline (requested at the top of the kernel) must have its first stages in the LDS now ----
  pipe.start_wait();

  auto dump_x = [&](const f4 (&x)[8][NT], int stage) {
    if (!p.dump) return;
    float* d = p.dump + (int64_t)stage * p.n_rows * p.V * 128;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      if (tok_row[jt] < 0) continue;
      float* row = d + (tok_row[jt] * p.V + tok_atom[jt]) * 128;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) *(f4*)(row + 16 * ft + 4 * g) = x[ft][jt];
    }
  };

  // debug bit 4 (16): wave 0 of workgroup 0 writes s_memtime stamps of the section boundaries into the dump
  // buffer instead of activations (tools/profile_h3_sections.py)
  auto stamp = [&](int idx) {
    if ((p.debug & 16) && p.dump && blockIdx.x == 0 && wave == 0 && lane == 0)
      ((unsigned long long*)p.dump)[idx] = __builtin_readcyclecounter();
  };
  stamp(0);
  if ((p.debug & 16) && p.dump && blockIdx.x == 0 && wave == 0 && lane == 0) {
    ((unsigned long long*)p.dump)[60] = t_kernel_top;
    ((unsigned long long*)p.dump)[61] = t_pro1;
    ((unsigned long long*)p.dump)[62] = t_pro2;
    ((unsigned long long*)p.dump)[63] = t_pro3;
  }

  // ---- IN stage ----
  f4 x[8][NT];
  {
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) x[ot][jt] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (ASM_IO && !RFF) {
      // generated asm (tools/gen_h3_ffn_asm.py --shape=in): u in, x out through the wave-private LDS block
      char* priv = (char*)xt_hi;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          *(h8*)(priv + ((ks * NT + jt) * 2) * 1024 + lane * 16) = u[ks].h[jt];
          *(h8*)(priv + ((ks * NT + jt) * 2 + 1) * 1024 + lane * 16) = u[ks].l[jt];
        }
      int cur = __builtin_amdgcn_readfirstlane(pipe.cur);
      const char* gn = pipe.gnext;
      const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
      const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
      const int chunks = __builtin_amdgcn_readfirstlane(p.hid_chunks);
      if constexpr (NT == 4 && H1) {
        asm volatile(
#include "tw_h1n4_in_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h1n4_in_clobbers.inc"
        );
      } else if constexpr (NT == 4) {
        asm volatile(
#include "tw_h3n4_in_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h3n4_in_clobbers.inc"
        );
      } else if constexpr (R6) {
        asm volatile(
#include "tw_h1r_in_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h1r_in_clobbers.inc"
        );
      } else if constexpr (H1) {
        asm volatile(
#include "tw_h1_in_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h1_in_clobbers.inc"
        );
      } else
      asm volatile(
#include "tw_h3_in_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3_in_clobbers.inc"
      );
      pipe.cur = cur;
      pipe.gnext = gn;
#pragma unroll
      for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) x[ot][jt] = *(const f4*)(priv + (ot * NT + jt) * 1024 + lane * 16);
    } else {
      h3_mlp_chain<NT, KIN, 8, true>(u, x, pipe, p.hid_chunks, lane);
    }
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) x[ot][jt] = x[ot][jt] * sc_in2 + bb_in2[ot];
    }
  }
  // Padding tokens (4 of a wave's 48 with 22-atom molecules) carry zeros instead of whatever the biases and LayerNorms
  // make of them.  No real token ever sees them (their score rows and columns are zero); the point is power: the launch
  // is power-limited, all-zero operand columns switch less, and the chip answers with clock (-0.6 % per pass, A/B on
  // one box, for 48 multiplications per call).  tw_debug_set_flags bit 10 (1024) turns it off.
  unsigned padmask = 0;  // bit jt: this lane's token of tile jt is padding (one register across the layer loop)
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) padmask |= (tok_row[jt] < 0 ? 1u : 0u) << jt;
  if (p.debug & 1024) padmask = 0;
  unsigned pad_tiles = 0;  // wave-uniform: token tiles that hold a padding token at all (V = 22: the last one only)
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
    if (__builtin_amdgcn_ballot_w64((padmask >> jt) & 1u) != 0ull) pad_tiles |= 1u << jt;
  pad_tiles = __builtin_amdgcn_readfirstlane(pad_tiles);
  auto zero_pad = [&]() {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      if (!((pad_tiles >> jt) & 1u)) continue;
      const bool pad = (padmask >> jt) & 1u;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[ft][jt][r] = pad ? 0.f : x[ft][jt][r];
    }
  };
  zero_pad();
  if (!(p.debug & 16)) dump_x(x, 0);
  stamp(1);

  const char* sf_net = p.sfrag + (int64_t)(net * p.n_layers) * p.sf_variant_bytes +
                       (WIDE ? (int64_t)((p.sfrag_shared ? 0 : wg * 4) + wave) * p.H * ((int64_t)p.ng * NT * 2048)
                             : (p.sfrag_shared ? 0 : (int64_t)blk * p.H * NT * SF_BYTES));

  const float* sl = (const float*)(lds + SIDE_LDS_OFFSET);  // this layer's side block, staged in LDS
  // dense: which key tokens of the wave each of this lane's query tokens may attend to - the unmasked atoms of the
  // query's own molecule (src_key_padding_mask, transformer_block.py:57-68) - pre-shifted by 4 g so that the bit of
  // key 16 mt + 4 g + r (the S^T accumulator element this lane holds) sits at position 16 mt + r
  unsigned long long kvalid[NT];
  if constexpr (DENSE) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      unsigned long long m = 0ull;
      if (tok_row[jt] >= 0) {
        const int q = (16 * jt + i16) / p.V;
        const uint8_t* mk = p.masked + tok_cond[jt] * p.V;
        for (int a = 0; a < p.V; ++a) m |= mk[a] ? 0ull : (1ull << (q * p.V + a));
      }
      kvalid[jt] = m >> (4 * g);
    }
  }
  if constexpr (ENC) {
    // The whole encoder stack as ONE generated statement (tools/gen_h3_enc_asm.py): attention and FFN blocks as below,
    // and the residual / LayerNorm / split / transpose glue between them hand-scheduled too, with the residual folded
    // into the accumulators' start values.  x goes in as 24 register images through the wave-private block; what comes
    // back there are the split operand images of the out-MLP statement.  No activation dumps, no section stamps: the
    // launch code takes the per-section build for those.
    static_assert(SIDE_CHUNKS == (DENSE ? 5 : 3), "gen_h3_enc_asm.py SIDE_CHUNKS");
    // [cur] / [gn] are EARLY-CLOBBER read-write operands ("+&") in every statement of this file: the embedded blocks write
    // %[cur] at each block's exit and the blocks behind it go on reading %[ring] and the other inputs.  Without the '&' hipcc
    // may give an input of the same known value the same register - it did in the dense model's position-feature build
    // (r05), where the compiled in-MLP leaves cur == 0 == ring: both in s24, the second block's ring wrap-around then
    // selected slot `cur` instead of the ring base, and the nets returned garbage.
    char* priv = (char*)xt_hi;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) *(f4*)(priv + (ft * NT + jt) * 1024 + lane * 16) = x[ft][jt];
    int cur = __builtin_amdgcn_readfirstlane(pipe.cur);
    const char* gn = pipe.gnext;
    const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
    const unsigned sl_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)(lds + SIDE_LDS_OFFSET);
    const int heads = __builtin_amdgcn_readfirstlane(p.H);
    const int chunks = __builtin_amdgcn_readfirstlane(p.ff_chunks);
    const int layers = __builtin_amdgcn_readfirstlane(p.n_layers);
    const char* sfp = sf_net;
    const int64_t sfstride = p.sf_variant_bytes;
    const char* sidep = (const char*)(side + p.side_layers) + lane * 16;
    const int64_t sidestride = (int64_t)p.side_layer_size * 4;
    const float* scp = scales + 2;
    const float eps = p.eps;
    // (only read by the H3_ENC_EXPERIMENT=stamps build of the statement, tools/profile_h3_sections.py)
    const float* stamp_base = p.dump;
    int stampen = __builtin_amdgcn_readfirstlane(((p.debug & 16) && p.dump && blockIdx.x == 0 && wave == 0) ? 1 : 0);
    // Opaque from here on: hipcc (ROCm 7.2) otherwise remembers that the value is a zero-extended i1, carries it to the
    // statements below as a lane mask and re-materialises it with v_cndmask - into a VGPR, which it then hands to the "s"
    // operand (seen with the wide layout's single-MFMA statements: "s_cmp_eq_u32 v252, 0", invalid operand)
    asm volatile("" : "+s"(stampen));
    if constexpr (NT == 4 || WIDE || DENSE) {
      // 64-token build (tools/gen_h3_enc_asm.py --nt=4): the statement owns v0..v245 and keeps nothing of its own in VGPRs
      // across the embedded blocks, so its pointers arrive in SGPRs (all of them are wave-uniform).  The wide layout's
      // statements (--wide [--ng=3|6]: the six-group attention block owns v0..v239) take the same form.
      auto uniform64 = [](const void* ptr) {
        const uint64_t u = (uint64_t)(uintptr_t)ptr;
        return (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u) |
               ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32)) << 32);
      };
      const uint64_t sf_u = uniform64(sf_net), side_u = uniform64((const char*)(side + p.side_layers)), dump_u = uniform64(stamp_base);
      const uint64_t scales_u = uniform64(scp);
      const unsigned sfstride_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uint64_t)sfstride);
      const unsigned sfstride_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uint64_t)sfstride >> 32));
      const unsigned sidestride32 = (unsigned)__builtin_amdgcn_readfirstlane((int)sidestride);
      if constexpr (WIDE && NT == 4) {
        // paired layout (tools/gen_h3_enc_asm.py --nt=4 --pair): the 64-token statement with a four-group attention block
        if constexpr (H1) {
          asm volatile(
#include "tw_h1n4p_enc_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen)
              :
#include "tw_h1n4p_enc_clobbers.inc"
          );
        } else {
          asm volatile(
#include "tw_h3n4p_enc_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen)
              :
#include "tw_h3n4p_enc_clobbers.inc"
          );
        }
      } else if constexpr (DENSE && NT == 4) {
        // r06: the dense model on 64-token waves (tools/gen_h3_enc_asm.py --dense --nt=4).  One molecule per wave, so ONE pair of
        // key-mask words serves all four query tiles: token tile 0 is always whole (49+ atoms), its lanes hold the molecule's mask
        // (a padding query of the last tile then attends like a real one; its row is zeroed behind every LayerNorm anyway)
        const unsigned m0l = (unsigned)kvalid[0], m0h = (unsigned)(kvalid[0] >> 32);
        asm volatile(
#include "tw_h3n4d_enc_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks), [layers] "s"(layers), [side] "s"(side_u),
              [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
              [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [m0l] "v"(m0l), [m0h] "v"(m0h)
            :
#include "tw_h3n4d_enc_clobbers.inc"
        );
      } else if constexpr (DENSE) {
        // dense softmax model (tools/gen_h3_enc_asm.py --dense [--h1]): no score fragments; the key masks of the softmax
        const unsigned m0l = (unsigned)kvalid[0], m0h = (unsigned)(kvalid[0] >> 32), m1l = (unsigned)kvalid[1],
                       m1h = (unsigned)(kvalid[1] >> 32), m2l = (unsigned)kvalid[2], m2h = (unsigned)(kvalid[2] >> 32);
        if constexpr (H1) {
          asm volatile(
#include "tw_h1d_enc_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks), [layers] "s"(layers), [side] "s"(side_u),
                [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [m0l] "v"(m0l), [m0h] "v"(m0h), [m1l] "v"(m1l),
                [m1h] "v"(m1h), [m2l] "v"(m2l), [m2h] "v"(m2h)
              :
#include "tw_h1d_enc_clobbers.inc"
          );
        } else {
          asm volatile(
#include "tw_h3d_enc_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks), [layers] "s"(layers), [side] "s"(side_u),
                [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [m0l] "v"(m0l), [m0h] "v"(m0h), [m1l] "v"(m1l),
                [m1h] "v"(m1h), [m2l] "v"(m2l), [m2h] "v"(m2h)
              :
#include "tw_h3d_enc_clobbers.inc"
          );
        }
      } else if constexpr (WIDE) {
        // x went in through the wave-private blocks, which lie under the workgroup's shared X^T tile: the statement reads its
        // images, takes a barrier, and only then writes the first transposed copy (tools/gen_h3_enc_asm.py generate())
        const unsigned xt_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)(lds + RING * H3_STAGE_BYTES);
        const int win = __builtin_amdgcn_readfirstlane(wave == 0 ? p.win[0] : wave == 1 ? p.win[1] : wave == 2 ? p.win[2] : p.win[3]);
        if constexpr (H1) {
          if constexpr (NG6) {
            asm volatile(
#include "tw_h1w6_enc_asm.inc"
                : [cur] "+&s"(cur), [gn] "+&v"(gn)
                : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                  [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                  [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                  [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [xt] "s"(xt_lds), [win] "s"(win)
                :
#include "tw_h1w6_enc_clobbers.inc"
            );
          } else if (p.ng == H3W_NG3) {
            asm volatile(
#include "tw_h1w3_enc_asm.inc"
                : [cur] "+&s"(cur), [gn] "+&v"(gn)
                : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                  [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                  [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                  [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [xt] "s"(xt_lds), [win] "s"(win)
                :
#include "tw_h1w3_enc_clobbers.inc"
            );
          } else {
            asm volatile(
#include "tw_h1w_enc_asm.inc"
                : [cur] "+&s"(cur), [gn] "+&v"(gn)
                : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                  [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                  [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                  [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [xt] "s"(xt_lds), [win] "s"(win)
                :
#include "tw_h1w_enc_clobbers.inc"
            );
          }
        } else if constexpr (NG6) {
          asm volatile(
#include "tw_h3w6_enc_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [xt] "s"(xt_lds), [win] "s"(win)
              :
#include "tw_h3w6_enc_clobbers.inc"
          );
        } else if (p.ng == H3W_NG3) {
          asm volatile(
#include "tw_h3w3_enc_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [xt] "s"(xt_lds), [win] "s"(win)
              :
#include "tw_h3w3_enc_clobbers.inc"
          );
        } else {
          asm volatile(
#include "tw_h3w_enc_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
                [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
                [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
                [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen), [xt] "s"(xt_lds), [win] "s"(win)
              :
#include "tw_h3w_enc_clobbers.inc"
          );
        }
      } else
      if constexpr (H1) {
        asm volatile(
#include "tw_h1n4_enc_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
              [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
              [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
              [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen)
            :
#include "tw_h1n4_enc_clobbers.inc"
        );
      } else {
        asm volatile(
#include "tw_h3n4_enc_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
              [layers] "s"(layers), [sf] "s"(sf_u), [sfstride] "s"(sfstride_lo), [sfstridehi] "s"(sfstride_hi), [side] "s"(side_u),
              [sidestride] "s"(sidestride32), [sl] "s"(sl_lds), [scales] "s"(scales_u), [eps] "s"(eps), [padm] "v"(padmask),
              [padt] "s"(pad_tiles), [dump] "s"(dump_u), [stampen] "s"(stampen)
            :
#include "tw_h3n4_enc_clobbers.inc"
        );
      }
    } else
    if constexpr (R6) {
    if (p.windowed) {
      asm volatile(
#include "tw_h1r_encw_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
            [layers] "s"(layers), [sf] "v"(sfp), [sfstride] "s"(sfstride), [side] "v"(sidep), [sidestride] "s"(sidestride),
            [sl] "s"(sl_lds), [scales] "s"(scp), [eps] "s"(eps), [padm] "v"(padmask), [padt] "s"(pad_tiles),
            [dump] "v"(stamp_base), [stampen] "s"(stampen)
          :
#include "tw_h1r_enc_clobbers.inc"
      );
    } else {
      asm volatile(
#include "tw_h1r_enc_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
            [layers] "s"(layers), [sf] "v"(sfp), [sfstride] "s"(sfstride), [side] "v"(sidep), [sidestride] "s"(sidestride),
            [sl] "s"(sl_lds), [scales] "s"(scp), [eps] "s"(eps), [padm] "v"(padmask), [padt] "s"(pad_tiles),
            [dump] "v"(stamp_base), [stampen] "s"(stampen)
          :
#include "tw_h1r_enc_clobbers.inc"
      );
    }
    } else
    if constexpr (H1) {
    if (p.windowed) {
      asm volatile(
#include "tw_h1_encw_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
            [layers] "s"(layers), [sf] "v"(sfp), [sfstride] "s"(sfstride), [side] "v"(sidep), [sidestride] "s"(sidestride),
            [sl] "s"(sl_lds), [scales] "s"(scp), [eps] "s"(eps), [padm] "v"(padmask), [padt] "s"(pad_tiles),
            [dump] "v"(stamp_base), [stampen] "s"(stampen)
          :
#include "tw_h1_enc_clobbers.inc"
      );
    } else {
      asm volatile(
#include "tw_h1_enc_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
            [layers] "s"(layers), [sf] "v"(sfp), [sfstride] "s"(sfstride), [side] "v"(sidep), [sidestride] "s"(sidestride),
            [sl] "s"(sl_lds), [scales] "s"(scp), [eps] "s"(eps), [padm] "v"(padmask), [padt] "s"(pad_tiles),
            [dump] "v"(stamp_base), [stampen] "s"(stampen)
          :
#include "tw_h1_enc_clobbers.inc"
      );
    }
    } else {
    if (p.windowed) {
      asm volatile(
#include "tw_h3_encw_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
            [layers] "s"(layers), [sf] "v"(sfp), [sfstride] "s"(sfstride), [side] "v"(sidep), [sidestride] "s"(sidestride),
            [sl] "s"(sl_lds), [scales] "s"(scp), [eps] "s"(eps), [padm] "v"(padmask), [padt] "s"(pad_tiles),
            [dump] "v"(stamp_base), [stampen] "s"(stampen)
          :
#include "tw_h3_enc_clobbers.inc"
      );
    } else {
      asm volatile(
#include "tw_h3_enc_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [chunks] "s"(chunks),
            [layers] "s"(layers), [sf] "v"(sfp), [sfstride] "s"(sfstride), [side] "v"(sidep), [sidestride] "s"(sidestride),
            [sl] "s"(sl_lds), [scales] "s"(scp), [eps] "s"(eps), [padm] "v"(padmask), [padt] "s"(pad_tiles),
            [dump] "v"(stamp_base), [stampen] "s"(stampen)
          :
#include "tw_h3_enc_clobbers.inc"
      );
    }
    }
    pipe.cur = cur;
    pipe.gnext = gn;
    stamp(2 + 4 * (p.n_layers - 1) + 3);
  } else
  for (int l = 0; l < p.n_layers; ++l) {
    const char* sf_base = sf_net + l * p.sf_variant_bytes;  // the layer's own score fragments (chebyshev_kernel), else shared
    // Stage the layer's LayerNorm parameters, FFN output bias and the two output scales (2.6 KB) in LDS: wave 0
    // issues three 1 KiB LDS-DMA chunks and nobody waits for them here - they are older than every weight-stage
    // DMA of the layer, so the first stage hand-off (vmcnt wait + barrier) covers them long before the first use.
    // The barrier keeps the previous layer's last reads of the block ahead of the overwrite.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == 0) {
      const char* src = (const char*)(side + p.side_layers + (int64_t)l * p.side_layer_size) + lane * 16;
      char* dst = lds + SIDE_LDS_OFFSET;
#pragma unroll
      for (int i = 0; i < SIDE_CHUNKS; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    }
    // x -> transposed fp16 hi/lo tile in LDS (operand of the mixing; the dense variant has none)
    if constexpr (WIDE) {
      // the workgroup's shared tile [feature][192 tokens] over the four wave-private blocks (every wave has passed the
      // barrier above, so nobody still reads operands there); a barrier before anyone mixes against other waves' columns
      char* xts = lds + RING * H3_STAGE_BYTES;
      if (p.debug & 524288) {
        // r03's form, kept as the A/B reference (tw_debug_set_flags bit 19): 192 two-byte stores per lane and layer - the
        // section profile put 46 k cycles per layer here, 15 % of the launch (profiles/r04_wide_ng3_sections.txt)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
          for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = x[ft][jt][r];
              const _Float16 hi = (_Float16)v;
              const int off = (16 * ft + 4 * g + r) * H3W_XT_ROW + 2 * (slot0 + 16 * jt + i16);
              *(_Float16*)(xts + off) = hi;
              *(_Float16*)(xts + H3W_XT_LO + off) = (_Float16)(v - (float)hi);
            }
      } else {
        // Transposed through the matrix pipe like the 48-token kernel's tile (below): K = 16 MFMAs against the identity
        // return X[token 16 jt + 4 g + r][feature 16 ft + i16] with the FEATURE on the lane and four consecutive TOKENS in
        // its registers - exactly (products with 1.0, sums with zeros) - i.e. 8 contiguous bytes of the feature's row:
        // 48 (fast mode: 24) eight-byte stores per lane and layer.  Rows 16 ft + i16 at a stride of 104 dwords + 2 g dwords
        // cover all 64 banks twice per 64-lane store: conflict-free.
        h4 idw;
#pragma unroll
        for (int e = 0; e < 4; ++e) idw[e] = (i16 == 4 * g + e) ? (_Float16)1.f : (_Float16)0.f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            h4 xh, xl;
            split4(x[ft][jt], xh, xl);
            const int off = (16 * ft + i16) * H3W_XT_ROW + 2 * (slot0 + 16 * jt + 4 * g);
            const f4 th = mfma16(xh, idw, (f4){0.f, 0.f, 0.f, 0.f});
            *(u2*)(xts + off) = __builtin_bit_cast(u2, __builtin_convertvector(th, h4));
            if constexpr (!H1) {
              const f4 tl = mfma16(xl, idw, (f4){0.f, 0.f, 0.f, 0.f});
              *(u2*)(xts + H3W_XT_LO + off) = __builtin_bit_cast(u2, __builtin_convertvector(tl, h4));
            }
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else if constexpr (!DENSE) {
      // Transposed through the matrix pipe, not through 192 two-byte LDS stores per lane: the lane holds token i16's
      // features 16 ft + 4 g + r - as fp16 that is the A operand (row = token, k = 4 g + r) of a K = 16 MFMA; against the
      // identity as B the result D[4 g + r][i16] = X[token 16 jt + 4 g + r][feature 16 ft + i16] comes back with the
      // FEATURE on the lane and four TOKENS in its registers, exactly (products with 1.0, sums with zeros), i.e. as the A
      // operand of the mixing MFMA.  Images per (feature tile ft, part): [T0 | T1] 16 B per lane (K = 32 operand: tokens
      // 4 g + e and 16 + 4 g + e - the key order of the K index is free as long as the score fragments use the same one,
      // h3_score_frag_kernel), then T2 8 B per lane (K = 16 operand, tokens 32 + 4 g + e): 1536 B, hi at
      // (2 ft) * 1536, lo at (2 ft + 1) * 1536.  Every lane reads and writes its own 16 / 8 bytes: no bank conflicts.
      // (r01-r03: rows of 48 halfs per feature, 4.9 k cycles per layer for this block; pairing tokens per store
      // through DPP measured worse, profiles/r03_ab_glue.txt)
      static_assert(NT == 3 || NT == 4, "token tiles 0, 1 form a K = 32 operand; tile 2 a K = 16 one, or tiles 2, 3 a second K = 32");
      char* xt = (char*)xt_hi;
      h4 idb;
#pragma unroll
      for (int e = 0; e < 4; ++e) idb[e] = (i16 == 4 * g + e) ? (_Float16)1.f : (_Float16)0.f;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        u2 ph[NT], pl[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          h4 xh, xl;
          split4(x[ft][jt], xh, xl);
          const f4 th = mfma16(xh, idb, (f4){0.f, 0.f, 0.f, 0.f});
          const f4 tl = mfma16(xl, idb, (f4){0.f, 0.f, 0.f, 0.f});
          ph[jt] = __builtin_bit_cast(u2, __builtin_convertvector(th, h4));
          pl[jt] = __builtin_bit_cast(u2, __builtin_convertvector(tl, h4));
        }
        char* img = xt + ft * (2 * XT_IMG);
        *(u4*)(img + lane * 16) = (u4){ph[0][0], ph[0][1], ph[1][0], ph[1][1]};
        *(u4*)(img + XT_IMG + lane * 16) = (u4){pl[0][0], pl[0][1], pl[1][0], pl[1][1]};
        if constexpr (NT == 4) {  // [T2 | T3]: the second K = 32 operand
          *(u4*)(img + 1024 + lane * 16) = (u4){ph[2][0], ph[2][1], ph[3][0], ph[3][1]};
          *(u4*)(img + XT_IMG + 1024 + lane * 16) = (u4){pl[2][0], pl[2][1], pl[3][0], pl[3][1]};
        } else {
          *(u2*)(img + 1024 + lane * 8) = ph[2];
          *(u2*)(img + XT_IMG + 1024 + lane * 8) = pl[2];
        }
      }
    }

    f4 y[8][NT];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) y[ot][jt] = (f4){0.f, 0.f, 0.f, 0.f};

    stamp(40 + 4 * l + 0);
    if constexpr (DENSE && ASM_ATT) {
      // Hand-scheduled dense-softmax attention block (tools/gen_h3_dense_attn_asm.py): the split activations in through the
      // wave-private LDS block, y back the same way; biases and the in_proj scale from the layer's side block in LDS
      // (landed: the first stage hand-off inside the block lies between its DMA and the first read).
      static_assert(NT == 3, "key tiles 0, 1 form the K = 32 part of P.V, tile 2 the K = 16 part");
      BOp<NT> xb[4];
      to_bop<NT, 4>(x, xb);
      char* priv = (char*)xt_hi;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          *(h8*)(priv + ((ks * NT + jt) * 2) * 1024 + lane * 16) = xb[ks].h[jt];
          *(h8*)(priv + ((ks * NT + jt) * 2 + 1) * 1024 + lane * 16) = xb[ks].l[jt];
        }
      int cur = __builtin_amdgcn_readfirstlane(pipe.cur);
      const char* gn = pipe.gnext;
      const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
      const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
      const unsigned sl_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)(lds + SIDE_LDS_OFFSET);
      const unsigned m0l = (unsigned)kvalid[0], m0h = (unsigned)(kvalid[0] >> 32), m1l = (unsigned)kvalid[1],
                     m1h = (unsigned)(kvalid[1] >> 32), m2l = (unsigned)kvalid[2], m2h = (unsigned)(kvalid[2] >> 32);
      asm volatile(
#include "tw_h3_attnd_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [sl] "s"(sl_lds), [m0l] "v"(m0l), [m0h] "v"(m0h),
            [m1l] "v"(m1l), [m1h] "v"(m1h), [m2l] "v"(m2l), [m2h] "v"(m2h)
          :
#include "tw_h3_attnd_clobbers.inc"
      );
      pipe.cur = cur;
      pipe.gnext = gn;
      stamp(40 + 4 * l + 1);
#pragma unroll
      for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = *(const f4*)(priv + (ot * NT + jt) * 1024 + lane * 16);
    } else if constexpr (DENSE) {
      // compiled-C++ statement of the same block (tw_debug_set_flags bit 3), same stage order:
      //   q_0 k_0 | per head h: v_h, [head h], q_{h+1} k_{h+1}, after an odd h the out_proj k-step of the pair
      // (NT = 3: key tiles 0, 1 form the K = 32 part of P.V, tile 2 the K = 16 part; NT = 4: tiles 2, 3 a second K = 32 part)
      BOp<NT> xb[4];
      to_bop<NT, 4>(x, xb);
      if constexpr (NT == 4) {
        // 64-token waves: x (128 fp32 registers per lane, needed again only for the residual behind the block) waits in the
        // wave-private LDS block, which nothing uses while this compiled block runs - x + y + xb + the head's operands do not
        // fit 512 registers, and what hipcc spilled instead cost the block 75 % (section profile: 142 k cycles per layer)
        char* priv = (char*)xt_hi;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) *(f4*)(priv + (ft * NT + jt) * 1024 + lane * 16) = x[ft][jt];
      }
      constexpr float LOG2E = 1.44269504088896340736f;
      f4 qa[NT], ka[NT], va[NT];
      f4 oh[2][NT];
      auto stage_qk = [&](f4 (&acc)[NT]) {  // acc = W[16-row tile of the stage] . x^T
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) acc[jt] = (f4){0.f, 0.f, 0.f, 0.f};
        H3Tiles w;
        w.load(pipe.stage(), lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          w.ready(ks);
          mma3<NT>(w.hi(ks), w.lo(ks), xb[ks], acc);
          w.done(ks);
        }
        pipe.advance();
      };
      auto stage_v = [&]() {  // operands swapped: va[jt] = x[tokens of tile jt] . W_v[16 h ..]^T  (lane = feature, registers = tokens)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) va[jt] = (f4){0.f, 0.f, 0.f, 0.f};
        H3Tiles w;
        w.load(pipe.stage(), lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          w.ready(ks);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) va[jt] = mfma32(xb[ks].h[jt], w.hi(ks), va[jt]);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) va[jt] = mfma32(xb[ks].l[jt], w.hi(ks), va[jt]);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) va[jt] = mfma32(xb[ks].h[jt], w.lo(ks), va[jt]);
          w.done(ks);
        }
        pipe.advance();
      };
      stage_qk(qa);
      stage_qk(ka);
      for (int h = 0; h < p.H; ++h) {
        const int hh = h & 1;
        stage_v();
        {
          // scale, bias, 1 / sqrt(16) on q; fp16 hi / lo operands
          h4 qh[NT], ql[NT], kh[NT], kl[NT], vh2, vl2;
          h8 vh01, vl01, vh23, vl23;
          {
            const f4 bq = *(const f4*)(sl + H3D_INB + 16 * h + 4 * g);
            const f4 bk = *(const f4*)(sl + H3D_INB + 128 + 16 * h + 4 * g);
            const float bv = sl[H3D_INB + 256 + 16 * h + i16];
            const float sc_in = sl[640];  // (the side block has landed: three stage hand-offs lie behind us)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
              split4((qa[jt] * sc_in + bq) * 0.25f, qh[jt], ql[jt]);
              split4(ka[jt] * sc_in + bk, kh[jt], kl[jt]);
              va[jt] = va[jt] * sc_in + bv;
            }
            split8(va[0], va[1], vh01, vl01);
            if constexpr (NT == 4) split8(va[2], va[NT - 1], vh23, vl23);
            else split4(va[2], vh2, vl2);
          }
          if (TW_EXPERIMENT(p.debug & 2048)) {  // timing experiment (results WRONG): no scores / softmax / P.V - what the GEMM stages alone cost
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) oh[hh][jt] = qa[jt] + ka[jt] + va[jt];
          } else
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            // S^T[key tile mt][query tile jt]: this lane holds keys 16 mt + 4 g + r of its query 16 jt + i16
            f4 sc[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sc[mt] = mfma16(kh[mt], qh[jt], (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sc[mt] = mfma16(kh[mt], ql[jt], sc[mt]);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) sc[mt] = mfma16(kl[mt], qh[jt], sc[mt]);
            // keys outside the query's molecule and padded atoms drop out.  -3e4 rather than -inf: e^(-3e4 - max) is 0 for
            // any real row, and the rows of padding tokens - no keys at all - stay finite (uniform weights) instead of
            // turning into NaN that 0-weights would not remove from the P.V products of their neighbours
            constexpr float MASKED = -3.0e4f;
            float mx = MASKED;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                sc[mt][r] = ((kvalid[jt] >> (16 * mt + r)) & 1ull) ? sc[mt][r] : MASKED;
                mx = fmaxf(mx, sc[mt][r]);
              }
            mx = h3_quad_max(mx) * LOG2E;
            float sum = 0.f;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                sc[mt][r] = __builtin_amdgcn_exp2f(fmaf(sc[mt][r], LOG2E, -mx));
                sum += sc[mt][r];
              }
            sum = h3_quad_sum(sum);
            h8 ph01, pl01;
            split8(sc[0], sc[1], ph01, pl01);
            if constexpr (NT == 4) {
              // 64 keys = two K = 32 operands of the same shape: two accumulation chains, summed on the VALU
              h8 ph23, pl23;
              split8(sc[2], sc[NT - 1], ph23, pl23);
              f4 oa = mfma32(vh01, ph01, (f4){0.f, 0.f, 0.f, 0.f});
              f4 ob = mfma32(vh23, ph23, (f4){0.f, 0.f, 0.f, 0.f});
              oa = mfma32(vh01, pl01, oa);
              ob = mfma32(vh23, pl23, ob);
              oa = mfma32(vl01, ph01, oa);
              ob = mfma32(vl23, ph23, ob);
              oh[hh][jt] = (oa + ob) * __builtin_amdgcn_rcpf(sum);
            } else {
            h4 ph2, pl2;
            split4(sc[2], ph2, pl2);
            // O^T[feature][query]; the K = 32 and the K = 16 part in separate accumulators (mixed-shape chains, see above)
            f4 o32 = mfma32(vh01, ph01, (f4){0.f, 0.f, 0.f, 0.f});
            f4 o16 = mfma16(vh2, ph2, (f4){0.f, 0.f, 0.f, 0.f});
            o32 = mfma32(vh01, pl01, o32);
            o16 = mfma16(vh2, pl2, o16);
            o32 = mfma32(vl01, ph01, o32);
            o16 = mfma16(vl2, ph2, o16);
            oh[hh][jt] = (o32 + o16) * __builtin_amdgcn_rcpf(sum);
            }
          }
        }
        if (h + 1 < p.H) {
          stage_qk(qa);
          stage_qk(ka);
        }
        if (hh == 1) {
          // y += W_o[:, 32 hp .. 32 hp + 31] . O(heads 2 hp, 2 hp + 1)
          BOp<NT> ob;
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) split8(oh[0][jt], oh[1][jt], ob.h[jt], ob.l[jt]);
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            H3Tiles w;
            w.load(pipe.stage(), lane);
#pragma unroll
            for (int oo = 0; oo < 4; ++oo) {
              w.ready(oo);
              mma3<NT>(w.hi(oo), w.lo(oo), ob, y[4 * half + oo]);
              w.done(oo);
            }
            pipe.advance();
          }
        }
      }
      if constexpr (NT == 4) {
        const char* priv = (const char*)xt_hi;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) x[ft][jt] = *(const f4*)(priv + (ft * NT + jt) * 1024 + lane * 16);
      }
      stamp(40 + 4 * l + 1);
    } else if constexpr (ASM_ATT) {
      // Hand-scheduled attention block (tools/gen_h3_attn_asm.py): all heads, mixing + Wc GEMM; reads the
      // transposed copy of x written above, returns y through the same wave-private LDS block.
      char* priv = (char*)xt_hi;
      int cur = __builtin_amdgcn_readfirstlane(pipe.cur);
      const char* gn = pipe.gnext;
      const char* sfp = sf_base;
      const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
      const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
      const int heads = __builtin_amdgcn_readfirstlane(p.H);
      if constexpr (WIDE) {
        // wide layout (tools/gen_h3_attn_wide_asm.py): mixing against the wave's key window of the shared X^T tile
        const unsigned xt_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)(lds + RING * H3_STAGE_BYTES);
        const int win = __builtin_amdgcn_readfirstlane(wave == 0 ? p.win[0] : wave == 1 ? p.win[1] : wave == 2 ? p.win[2] : p.win[3]);
        // (p.ng is launch-uniform: three-group windows for 65 .. 96 atoms at the 96-slot stride, five otherwise)
        if constexpr (H1) {
          if constexpr (NG6) {
          asm volatile(
#include "tw_h1_attns6_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp), [xt] "s"(xt_lds),
                [win] "s"(win)
              :
#include "tw_h1_attns6_clobbers.inc"
          );
          } else if (p.ng == H3W_NG3) {
          asm volatile(
#include "tw_h1_attns3_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp), [xt] "s"(xt_lds),
                [win] "s"(win)
              :
#include "tw_h1_attns3_clobbers.inc"
          );
          } else {
          asm volatile(
#include "tw_h1_attns_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp), [xt] "s"(xt_lds),
                [win] "s"(win)
              :
#include "tw_h1_attns_clobbers.inc"
          );
          }
        } else if constexpr (NG6) {
        asm volatile(
#include "tw_h3_attns6_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp), [xt] "s"(xt_lds),
              [win] "s"(win)
            :
#include "tw_h3_attns6_clobbers.inc"
        );
        } else if (p.ng == H3W_NG3) {
        asm volatile(
#include "tw_h3_attns3_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp), [xt] "s"(xt_lds),
              [win] "s"(win)
            :
#include "tw_h3_attns3_clobbers.inc"
        );
        } else {
        asm volatile(
#include "tw_h3_attns_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp), [xt] "s"(xt_lds),
              [win] "s"(win)
            :
#include "tw_h3_attns_clobbers.inc"
        );
        }
      } else if constexpr (NT == 4 && H1) {
        asm volatile(
#include "tw_h1n4_attn_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp)
            :
#include "tw_h1n4_attn_clobbers.inc"
        );
      } else if constexpr (NT == 4) {
        asm volatile(
#include "tw_h3n4_attn_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp)
            :
#include "tw_h3n4_attn_clobbers.inc"
        );
      } else if (p.windowed) {
        // two or more molecules per wave: 24 instead of 36 mixing MFMAs per k-step (gen_h3_attn_asm.py --mode=windowed)
        asm volatile(
#include "tw_h3_attnw_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp)
            :
#include "tw_h3_attnw_clobbers.inc"
        );
      } else {
        asm volatile(
#include "tw_h3_attn_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [heads] "s"(heads), [sf] "v"(sfp)
            :
#include "tw_h3_attn_clobbers.inc"
        );
      }
      pipe.cur = cur;
      pipe.gnext = gn;
      stamp(40 + 4 * l + 1);
#pragma unroll
      for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = *(const f4*)(priv + (ot * NT + jt) * 1024 + lane * 16);
    } else
    for (int h = 0; h < p.H; ++h) {
      // score fragments of this head (B operand of the mixing MFMA)
      // mixing: xm = (A_h X)^T, produced directly as the split B operand of the Wc GEMM
      BOp<NT> xm[4];
      if constexpr (NT == 4) {
        // 64-token waves: keys 0-31 and 32-63 are two K = 32 operands of the same shape, one accumulation chain
        u4 r[16];
        h3_load_sf4(sf_base + (int64_t)(h * NT) * SF_BYTES + lane * 16, r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          f4 acc[2][NT];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const char* img = (const char*)xt_hi + (2 * ks + t) * (2 * XT_IMG);
            const h8 a0h = *(const h8*)(img + lane * 16);
            const h8 a0l = *(const h8*)(img + XT_IMG + lane * 16);
            const h8 a1h = *(const h8*)(img + 1024 + lane * 16);
            const h8 a1l = *(const h8*)(img + XT_IMG + 1024 + lane * 16);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a0h, __builtin_bit_cast(h8, r[4 * jt]), (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a1h, __builtin_bit_cast(h8, r[4 * jt + 2]), acc[t][jt]);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a0h, __builtin_bit_cast(h8, r[4 * jt + 1]), acc[t][jt]);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a1h, __builtin_bit_cast(h8, r[4 * jt + 3]), acc[t][jt]);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a0l, __builtin_bit_cast(h8, r[4 * jt]), acc[t][jt]);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a1l, __builtin_bit_cast(h8, r[4 * jt + 2]), acc[t][jt]);
          }
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) split8(acc[0][jt], acc[1][jt], xm[ks].h[jt], xm[ks].l[jt]);
        }
      } else {
      u4 r0h[3], r0l[3];
      u2 r1h[3], r1l[3];
      {
        const char* sp = sf_base + (int64_t)(h * NT) * H3_SF_BYTES;
        h3_load_sf3(sp + lane * 16, sp + lane * 16, r0h, r0l, r1h, r1l);
      }
      h8 s0h[3], s0l[3];
      h4 s1h[3], s1l[3];
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) {
        s0h[jt] = __builtin_bit_cast(h8, r0h[jt]);
        s0l[jt] = __builtin_bit_cast(h8, r0l[jt]);
        s1h[jt] = __builtin_bit_cast(h4, r1h[jt]);
        s1l[jt] = __builtin_bit_cast(h4, r1l[jt]);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // The K=32 body and the K=16 tail accumulate in SEPARATE registers and are summed on the VALU.
        // A dependent chain that alternates v_mfma_f32_16x16x32_f16 and v_mfma_f32_16x16x16_f16 on one
        // accumulator needs >= 5 wait states between the two opcodes on gfx950 (SrcC forwarding only
        // works between MFMAs of the same shape); hipcc (ROCm 7.2) does not insert them and the chain
        // silently drops terms (tools/probe/mfma_chain_builtin.hip reproduces it: 192 instead of 384).
        f4 acc[2][NT], tail[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* img = (const char*)xt_hi + (2 * ks + t) * (2 * H3_XT_IMG);
          const h8 a0h = *(const h8*)(img + lane * 16);
          const h8 a0l = *(const h8*)(img + H3_XT_IMG + lane * 16);
          const h4 a1h = *(const h4*)(img + 1024 + lane * 8);
          const h4 a1l = *(const h4*)(img + H3_XT_IMG + 1024 + lane * 8);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a0h, s0h[jt], (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) tail[t][jt] = mfma16(a1h, s1h[jt], (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a0h, s0l[jt], acc[t][jt]);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) tail[t][jt] = mfma16(a1h, s1l[jt], tail[t][jt]);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[t][jt] = mfma32(a0l, s0h[jt], acc[t][jt]);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) tail[t][jt] = mfma16(a1l, s1h[jt], tail[t][jt]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[t][jt] = acc[t][jt] + tail[t][jt];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) split8(acc[0][jt], acc[1][jt], xm[ks].h[jt], xm[ks].l[jt]);
      }
      }
      // y += Wc_h . xm : eight stages, ks-major (k-step ks, output tiles 4 half .. 4 half + 3)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const char* st = pipe.stage();
          H3Tiles w;
          w.load(st, lane);
#pragma unroll
          for (int oo = 0; oo < 4; ++oo) {
            w.ready(oo);
            mma3<NT>(w.hi(oo), w.lo(oo), xm[ks], y[4 * half + oo]);
            w.done(oo);
          }
          pipe.advance();
        }
    }
    if constexpr (DENSE) {
      const float sc = sl[642];  // out_proj scale and bias
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) {
        const f4 bo = *(const f4*)(sl + H3D_OUTB + 16 * ot + 4 * g);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = y[ot][jt] * sc + bo;
      }
    } else {
      const float sc = sl[640];
#pragma unroll
      for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = y[ot][jt] * sc;
    }
    if (p.debug & 4) dump_x(y, l + 1);
    stamp(2 + 4 * l + 0);
    h3_add_layernorm<NT>(x, y, sl + 4 * g, sl + 128 + 4 * g, p.eps);
    zero_pad();
    stamp(2 + 4 * l + 1);

    // FFN
    {
      BOp<NT> xb[4];
      to_bop<NT, 4>(x, xb);
#pragma unroll
      for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = (f4){0.f, 0.f, 0.f, 0.f};
      if constexpr (ASM) {
        // Hand-scheduled chunk loop (tools/gen_h3_ffn_asm.py).  Operands travel through the wave-private LDS
        // block (free between the mixing of this layer and the transposed copy of the next): xb in as 24
        // x 1 KiB register images, y out the same way; the asm statement owns v0-v203 / a0-a119.
        char* priv = (char*)xt_hi;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            *(h8*)(priv + ((ks * NT + jt) * 2) * 1024 + lane * 16) = xb[ks].h[jt];
            *(h8*)(priv + ((ks * NT + jt) * 2 + 1) * 1024 + lane * 16) = xb[ks].l[jt];
          }
        stamp(40 + 4 * l + 2);
        int cur = __builtin_amdgcn_readfirstlane(pipe.cur);
        const char* gn = pipe.gnext;
        const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
        const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
        const int chunks = __builtin_amdgcn_readfirstlane(p.ff_chunks);
        if constexpr (NT == 4 && H1) {
          asm volatile(
#include "tw_h1n4_ffn_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
              :
#include "tw_h1n4_ffn_clobbers.inc"
          );
        } else if constexpr (NT == 4) {
          asm volatile(
#include "tw_h3n4_ffn_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
              :
#include "tw_h3n4_ffn_clobbers.inc"
          );
        } else if constexpr (H1) {
          asm volatile(
#include "tw_h1_ffn_asm.inc"
              : [cur] "+&s"(cur), [gn] "+&v"(gn)
              : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
              :
#include "tw_h1_ffn_clobbers.inc"
          );
        } else
        asm volatile(
#include "tw_h3_ffn_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h3_ffn_clobbers.inc"
        );
        pipe.cur = cur;
        pipe.gnext = gn;
        stamp(40 + 4 * l + 3);
#pragma unroll
        for (int ot = 0; ot < 8; ++ot)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) y[ot][jt] = *(const f4*)(priv + (ot * NT + jt) * 1024 + lane * 16);
      } else {
        h3_mlp_chain<NT, 4, 8, false>(xb, y, pipe, p.ff_chunks, lane);
      }
      const float sc = sl[641];
      f4 bb[8];
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) bb[ot] = *(const f4*)(sl + 256 + 4 * g + 16 * ot);
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = y[ot][jt] * sc + bb[ot];
      }
    }
    stamp(2 + 4 * l + 2);
    h3_add_layernorm<NT>(x, y, sl + 384 + 4 * g, sl + 512 + 4 * g, p.eps);
    zero_pad();
    if (!(p.debug & (4 | 16))) dump_x(x, l + 1);
    stamp(2 + 4 * l + 3);
  }

  // ---- OUT stage ----
  f4 o[1][NT];
  {
    BOp<NT> xb[4];
    if constexpr (!ENC) to_bop<NT, 4>(x, xb);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) o[0][jt] = (f4){0.f, 0.f, 0.f, 0.f};
    if constexpr (ASM_IO) {
      // generated asm (tools/gen_h3_ffn_asm.py --shape=out)
      char* priv = (char*)xt_hi;
      if constexpr (!ENC) {  // (ENC: the encoder-stack statement left the split operand images there)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            *(h8*)(priv + ((ks * NT + jt) * 2) * 1024 + lane * 16) = xb[ks].h[jt];
            *(h8*)(priv + ((ks * NT + jt) * 2 + 1) * 1024 + lane * 16) = xb[ks].l[jt];
          }
      }
      int cur = __builtin_amdgcn_readfirstlane(pipe.cur);
      const char* gn = pipe.gnext;
      const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
      const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
      const int chunks = __builtin_amdgcn_readfirstlane(p.hid_chunks);
      if constexpr (NT == 4 && H1) {
        asm volatile(
#include "tw_h1n4_out_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h1n4_out_clobbers.inc"
        );
      } else if constexpr (NT == 4) {
        asm volatile(
#include "tw_h3n4_out_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h3n4_out_clobbers.inc"
        );
      } else if constexpr (R6) {
        asm volatile(
#include "tw_h1r_out_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h1r_out_clobbers.inc"
        );
      } else if constexpr (H1) {
        asm volatile(
#include "tw_h1_out_asm.inc"
            : [cur] "+&s"(cur), [gn] "+&v"(gn)
            : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
            :
#include "tw_h1_out_clobbers.inc"
        );
      } else
      asm volatile(
#include "tw_h3_out_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3_out_clobbers.inc"
      );
      pipe.cur = cur;
      pipe.gnext = gn;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) o[0][jt] = *(const f4*)(priv + jt * 1024 + lane * 16);
    } else {
      h3_mlp_chain<NT, 4, 1, true>(xb, o, pipe, p.hid_chunks, lane);
    }
    if constexpr (ENC && DENSE && NT == 4) {
      // (see below: nothing per-lane may stay live across the encoder-stack statement of this build - the bias is loaded here)
      int lane_b;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_b));
      const f4 bb_late = *(const f4*)(side + p.side_out2b + 4 * (lane_b >> 4));
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) o[0][jt] = o[0][jt] * sc_out2 + bb_late;
    } else {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) o[0][jt] = o[0][jt] * sc_out2 + bb_out2;
    }
  }
  stamp(2 + 4 * p.n_layers);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // drain the over-fetched stages before the workgroup retires its LDS
  if constexpr (ENC && DENSE && NT == 4) {
    // r06: this build's statement owns v0..v245 AND all 256 AGPRs (y + the split activations), so the compiler has ten VGPRs
    // and no AGPR to carry values across it: the token bookkeeping of the epilogue (eight + four registers per lane in the other
    // builds) is recomputed here from a lane id the compiler cannot connect with the prologue's - ScratchSize stays 0.
    int lane_b;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_b));
    if ((lane_b >> 4) == 0) {
      float* outp = p.out[net];
      const unsigned inv_v2 = (65536u + (unsigned)p.P - 1u) / (unsigned)p.P;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int t = 16 * jt + (lane_b & 15);
        const int q = (int)(((unsigned)t * inv_v2) >> 16);
        const int64_t n = row0 + q;
        const int atom = t - q * p.P;
        if (!(active && q < p.mpw && n < p.n_rows && atom < p.V)) continue;
        float* dst = outp + (n * p.V + atom) * 3;
        dst[0] = o[0][jt][0];
        dst[1] = o[0][jt][1];
        dst[2] = o[0][jt][2];
      }
    }
  } else
  if (g == 0) {
    float* outp = p.out[net];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      if (tok_row[jt] < 0) continue;
      float* dst = outp + (tok_row[jt] * p.V + tok_atom[jt]) * 3;
      dst[0] = o[0][jt][0];
      dst[1] = o[0][jt][1];
      dst[2] = o[0][jt][2];
      if (p.dump) {
        float* dd = p.dump + (int64_t)(p.n_layers + 1) * p.n_rows * p.V * 128 + (tok_row[jt] * p.V + tok_atom[jt]) * 3;
        dd[0] = o[0][jt][0]; dd[1] = o[0][jt][1]; dd[2] = o[0][jt][2];
      }
    }
  }
}

// ================================================================================================
// r06 - TW_PATH_SIMPLE_H3 (molecules no fused layout takes): the FFN half of an encoder layer on a FLAT token list,
//   h <- LayerNorm2(h + W2 relu(W1 h + b1) + b2)          (custom_transformer_block.py:64-72 / transformer_block.py:62-72)
// with the fused kernels' own machinery: the layer's FFN stages of the split-fp16 stream through the LDS ring, the generated chunk
// loop (tw_h3_ffn_asm.inc: 64 chunks x 144 MFMAs per wave), the 2048-wide hidden layer never leaving the chip.  The per-op form
// wrote and re-read it through HBM (403 MB per call at 192 atoms x 256 rows).  A workgroup = 4 waves x 48 consecutive tokens;
// nothing here knows about molecules.
// ================================================================================================
struct H3FfnParams {
  const char* stages;      // first FFN stage of (coupling, net, layer) in the tw_flow_pack_h3 stream
  const float* side;       // the layer's side block: n1w n1b b2 n2w n2b [128 each], slot 641 = W2's scale
  float* h;                // [n_tokens, 128] in / out
  int64_t n_tokens;
  int ff_chunks;           // chunks of 32 hidden units THIS workgroup walks (all of them unless split > 1)
  float eps;
  int split;               // small launches: `split` workgroups per token tile, each over its share of the hidden layer ...
  float* parts;            // ... writing W2-scaled partial sums [split][n_tokens, 128] here; h3_ffn_parts_ln_kernel finishes the layer
};

// NT = 3: 48-token waves on the five-slot ring; NT = 4: 64-token waves on the three-slot ring (tw_h3n4_ffn_asm.inc) - one launch
// is whole rounds of one workgroup per CU, so the host picks the token count per workgroup that wastes less of the last round.
template <int NT>
__global__ void __launch_bounds__(256) h3_ffn_tokens_kernel(H3FfnParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int RING = NT == 4 ? H3N4_RING : H3_RING;
  constexpr int WAVE_LDS = NT == 4 ? H3N4_WAVE_LDS : H3_WAVE_LDS;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i16 = lane & 15;
  const int part = __builtin_amdgcn_readfirstlane(p.split > 1 ? (int)(blockIdx.x % p.split) : 0);
  const int64_t tile = p.split > 1 ? blockIdx.x / p.split : blockIdx.x;
  H3Pipe pipe;
  pipe.gnext = p.stages + (int64_t)part * p.ff_chunks * 4 * H3_STAGE_BYTES + lane * 16;   // (split: the parts' own streams, back to back)
  pipe.lds = lds;
  pipe.cur = 0;
  pipe.wave = wave;
  pipe.debug = 0;
  pipe.ring = RING;
  pipe.start_issue();
  char* priv = lds + RING * H3_STAGE_BYTES + wave * WAVE_LDS;
  const int64_t t0 = (tile * 4 + wave) * (16 * NT);
  f4 x[8][NT], y[8][NT];
  auto load_x = [&]() {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int64_t t = t0 + 16 * jt + i16;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft)
        x[ft][jt] = t < p.n_tokens ? *(const f4*)(p.h + t * 128 + 16 * ft + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  load_x();
  {
    BOp<NT> xb[4];
    to_bop<NT, 4>(x, xb);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        *(h8*)(priv + ((ks * NT + jt) * 2) * 1024 + lane * 16) = xb[ks].h[jt];
        *(h8*)(priv + ((ks * NT + jt) * 2 + 1) * 1024 + lane * 16) = xb[ks].l[jt];
      }
  }
  pipe.start_wait();   // vmcnt(0) + barrier: the first stages are in the ring
  {
    int cur = 0;
    const char* gn = pipe.gnext;
    const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
    const int chunks = __builtin_amdgcn_readfirstlane(p.ff_chunks);
    if constexpr (NT == 4) {
      asm volatile(
#include "tw_h3n4_ffn_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3n4_ffn_clobbers.inc"
      );
    } else {
      asm volatile(
#include "tw_h3_ffn_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3_ffn_clobbers.inc"
      );
    }
  }
  const float sc = p.side[641];
#pragma unroll
  for (int ot = 0; ot < 8; ++ot)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) y[ot][jt] = *(const f4*)(priv + (ot * NT + jt) * 1024 + lane * 16);
  if (p.split > 1) {
    float* dst = p.parts + (int64_t)part * p.n_tokens * 128;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int64_t t = t0 + 16 * jt + i16;
      if (t >= p.n_tokens) continue;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) *(f4*)(dst + t * 128 + 16 * ft + 4 * g) = y[ft][jt] * sc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // (64-token waves: the layer input is read again here instead of living in 128 registers across the statement)
  if constexpr (NT == 4) load_x();
#pragma unroll
  for (int ot = 0; ot < 8; ++ot) {
    const f4 bb = *(const f4*)(p.side + 256 + 4 * g + 16 * ot);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) y[ot][jt] = y[ot][jt] * sc + bb;
  }
  h3_add_layernorm<NT>(x, y, p.side + 384 + 4 * g, p.side + 512 + 4 * g, p.eps);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int64_t t = t0 + 16 * jt + i16;
    if (t >= p.n_tokens) continue;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) *(f4*)(p.h + t * 128 + 16 * ft + 4 * g) = x[ft][jt];
  }
  // the statement's last hand-offs requested stages past this FFN (the stream has slack for that): they land before the LDS goes
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ... and behind a split launch:  h <- LayerNorm2(h + sum_p parts[p] + b2), one wave per token
__global__ void __launch_bounds__(256) h3_ffn_parts_ln_kernel(H3FfnParams p) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= p.n_tokens) return;
  const int lane = threadIdx.x & 63;
  float* row = p.h + t * 128;
  float v0 = row[lane] + p.side[256 + lane], v1 = row[64 + lane] + p.side[320 + lane];
  for (int q = 0; q < p.split; ++q) {
    const float* pr = p.parts + ((int64_t)q * p.n_tokens + t) * 128;
    v0 += pr[lane];
    v1 += pr[64 + lane];
  }
  float sum = v0 + v1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum * (1.f / 128.f);
  const float d0 = v0 - mean, d1 = v1 - mean;
  float var = d0 * d0 + d1 * d1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = 1.0f / sqrtf(var * (1.f / 128.f) + p.eps);
  row[lane] = d0 * rstd * p.side[384 + lane] + p.side[512 + lane];
  row[64 + lane] = d1 * rstd * p.side[448 + lane] + p.side[576 + lane];
}

// The in-MLP and the out-MLP of a net block on the flat token list, the same way (tw_h3_in_asm.inc / tw_h3_out_asm.inc):
//   IN :  h = W2 silu(W0 u + b0) + b2     u [n_tokens, d_in <= 64] (build_input_kernel)  ->  h [n_tokens, 128]
//   OUT:  o = W2 silu(W0 h + b0) + b2     h [n_tokens, 128]                              ->  o [n_tokens, 3]
struct H3IoParams {
  const char* stages;   // first stage of the section in the tw_flow_pack_h3 stream
  const float* bias2;   // second-layer bias in the net's side floats (in: 128, out: 16)
  const float* scale2;  // second-layer output scale (a 2^-s multiplier the pack kernels wrote)
  const float* in;      // IN: u [n_tokens, d_in];  OUT: h [n_tokens, 128]
  float* out;           // IN: h [n_tokens, 128];   OUT: o [n_tokens, 3]
  int64_t n_tokens;
  int d_in, chunks;
};

template <bool OUT, int NT>
__global__ void __launch_bounds__(256) h3_io_tokens_kernel(H3IoParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int RING = NT == 4 ? H3N4_RING : H3_RING;
  constexpr int WAVE_LDS = NT == 4 ? H3N4_WAVE_LDS : H3_WAVE_LDS;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i16 = lane & 15;
  H3Pipe pipe;
  pipe.gnext = p.stages + lane * 16;
  pipe.lds = lds;
  pipe.cur = 0;
  pipe.wave = wave;
  pipe.debug = 0;
  pipe.ring = RING;
  pipe.start_issue();
  char* priv = lds + RING * H3_STAGE_BYTES + wave * WAVE_LDS;
  const int64_t t0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * NT);
  const float sc = *p.scale2;
  if constexpr (OUT) {
    f4 x[8][NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int64_t t = t0 + 16 * jt + i16;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft)
        x[ft][jt] = t < p.n_tokens ? *(const f4*)(p.in + t * 128 + 16 * ft + 4 * g) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    BOp<NT> xb[4];
    to_bop<NT, 4>(x, xb);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        *(h8*)(priv + ((ks * NT + jt) * 2) * 1024 + lane * 16) = xb[ks].h[jt];
        *(h8*)(priv + ((ks * NT + jt) * 2 + 1) * 1024 + lane * 16) = xb[ks].l[jt];
      }
  } else {
    // u in B-operand element order: k-step ks, element e <-> column 32 ks + 16 (e / 4) + 4 g + e % 4 (columns >= d_in: zero)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int64_t t = t0 + 16 * jt + i16;
        f4 a = (f4){0.f, 0.f, 0.f, 0.f}, b = (f4){0.f, 0.f, 0.f, 0.f};
        if (t < p.n_tokens) {
          const float* row = p.in + t * p.d_in;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c0 = 32 * ks + 4 * g + r, c1 = c0 + 16;
            a[r] = c0 < p.d_in ? row[c0] : 0.f;
            b[r] = c1 < p.d_in ? row[c1] : 0.f;
          }
        }
        h8 hi, lo;
        split8(a, b, hi, lo);
        *(h8*)(priv + ((ks * NT + jt) * 2) * 1024 + lane * 16) = hi;
        *(h8*)(priv + ((ks * NT + jt) * 2 + 1) * 1024 + lane * 16) = lo;
      }
  }
  pipe.start_wait();
  {
    int cur = 0;
    const char* gn = pipe.gnext;
    const unsigned ring = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)lds;
    const unsigned priv_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)priv;
    const int chunks = __builtin_amdgcn_readfirstlane(p.chunks);
    if constexpr (OUT && NT == 4) {
      asm volatile(
#include "tw_h3n4_out_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3n4_out_clobbers.inc"
      );
    } else if constexpr (OUT) {
      asm volatile(
#include "tw_h3_out_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3_out_clobbers.inc"
      );
    } else if constexpr (NT == 4) {
      asm volatile(
#include "tw_h3n4_in_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3n4_in_clobbers.inc"
      );
    } else {
      asm volatile(
#include "tw_h3_in_asm.inc"
          : [cur] "+&s"(cur), [gn] "+&v"(gn)
          : [ring] "s"(ring), [wave] "s"(wave), [priv] "s"(priv_lds), [chunks] "s"(chunks)
          :
#include "tw_h3_in_clobbers.inc"
      );
    }
  }
  if constexpr (OUT) {
    const f4 bb = *(const f4*)(p.bias2 + 4 * g);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const f4 o = *(const f4*)(priv + jt * 1024 + lane * 16) * sc + bb;
      const int64_t t = t0 + 16 * jt + i16;
      if (g == 0 && t < p.n_tokens) {
        p.out[t * 3 + 0] = o[0];
        p.out[t * 3 + 1] = o[1];
        p.out[t * 3 + 2] = o[2];
      }
    }
  } else {
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
      const f4 bb = *(const f4*)(p.bias2 + 4 * g + 16 * ot);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int64_t t = t0 + 16 * jt + i16;
        if (t < p.n_tokens) *(f4*)(p.out + t * 128 + 16 * ot + 4 * g) = *(const f4*)(priv + (ot * NT + jt) * 1024 + lane * 16) * sc + bb;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-fetched stages land before the LDS goes
}

// ================================================================================================
// host side
// ================================================================================================
struct H3Ws {
  float *s_out, *t_out;
  float *s_out2, *t_out2, *zc_alt, *zv_alt;  // second s/t set and second copies of the two variables (PrevCoupling)
  char* sfrag;
  int64_t sf_variant_bytes;  // per-block layout: bytes of one fragment set over all blocks
  int64_t bytes;
};

// molecules per block and number of blocks for n_rows conformations: wave-blocks (<= 48 atoms) or workgroup blocks (wide)
static bool h3_layout(const tw_flow_desc& d, int V, int64_t n_rows, bool h1, FusedGeom* fg, H3Wide* wd, bool* wide) {
  *wide = h3_wide_choice(d, V, n_rows, h1);
  if (*wide) {
    h3_wide_geom(V, wd);
    fg->nt = wd->nt;
    fg->mpw = wd->mpwg;
    fg->tile_mask = 0;
    return true;
  }
  return fused_geom_nt(V, h3_nt4_choice(d, V, n_rows, h1) ? H3N4_NT : H3_NT, fg);
}

static H3Ws h3_ws(const tw_flow_desc& d, int64_t n_rows, int V, void* base, bool h1 = false, int force_layout = -1) {
  FusedGeom g;
  H3Wide wd;
  bool wide = false;
  h3_layout(d, V, n_rows, h1, &g, &wd, &wide);
  if (force_layout >= 0) {  // sizing only - 0: 48-token waves, 1: wide, 2: 64-token waves; an empty H3Ws where the layout does not exist
    wide = force_layout == 1;
    if (wide) {
      if (!h3_wide_geom(V, &wd, false)) return H3Ws{};  // (97 .. 128 atoms: the paired layout holds two per workgroup and needs less)
      g.nt = H3_NT;
      g.mpw = wd.mpwg;
    } else if (!fused_geom_nt(V, force_layout == 2 ? H3N4_NT : H3_NT, &g)) {
      return H3Ws{};
    }
  }
  H3Ws w;
  char* p = (char*)base;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  const int64_t nblocks = (n_rows + g.mpw - 1) / g.mpw;
  w.s_out = (float*)take(n_rows * V * 3 * 4);
  w.t_out = (float*)take(n_rows * V * 3 * 4);
  w.s_out2 = (float*)take(n_rows * V * 3 * 4);
  w.t_out2 = (float*)take(n_rows * V * 3 * 4);
  w.zc_alt = (float*)take(n_rows * V * 3 * 4);
  w.zv_alt = (float*)take(n_rows * V * 3 * 4);
  // chebyshev_kernel: one fragment set per (net, layer) of the coupling layer in flight
  // + one head of slack: the attention asm block prefetches the "next head" also after the last one
  const int64_t variants = d.cheb_order > 0 ? 2 * d.n_layers : 1;
  if (d.variant == 1) w.sf_variant_bytes = 0;  // dense: no score fragments
  else if (wide) w.sf_variant_bytes = nblocks * 4 * d.n_heads * h3w_frag_head(wd);
  else w.sf_variant_bytes = nblocks * d.n_heads * g.nt * (g.nt == 4 ? H3N4_SF_BYTES : H3_SF_BYTES);
  // (wide: sized for six-group windows whatever the launch takes - the workspace a caller allocated stays large enough when
  // tw_debug_set_flags moves between the two statements)
  w.sfrag = take(wide ? (variants * nblocks * 4 * d.n_heads + 1) * (int64_t)H3W_FRAG_HEAD(H3W_NG6)
                      : variants * w.sf_variant_bytes + g.nt * (g.nt == 4 ? H3N4_SF_BYTES : H3_SF_BYTES));
  w.bytes = p - (char*)base;
  return w;
}

std::atomic<int> g_debug_flags{0};
static bool h3_pair_enabled() {
  const int f = g_debug_flags;
  // (bit 3: the compiled-C++ statement exists for the 48-token layouts only - ADVICE r05)
  return !t_no_pair && !(f & (8 | 1048576)) && (!(f & (4 | 16 | 4096)) || (f & 8192));
}

// Per-tile key windows for the mixing (gen_h3_attn_asm.py --mode=windowed): every molecule that has a token in query
// tile 0 ends before token 32, and every molecule with a token in tile 2 starts at token 16 or later.  True for all
// multi-molecule wave layouts of 48 tokens (V <= 24); the compiled-C++ variant of the kernel keeps the full 48 keys.
static bool h3_windowed(const FusedGeom& fg, int V) {
  if (g_debug_flags & 8) return false;
  if (fg.mpw < 2) return false;
  const int end0 = (15 / V + 1) * V, start2 = (32 / V) * V;
  return end0 <= 32 && start2 >= 16;
}


int64_t h3_workspace_bytes(const tw_flow_desc& d, int64_t n_rows, int n_atoms) {
  // the largest of the layouts that exist for this size: which one a call takes depends on its own row count and on
  // tw_debug_set_flags (h3_wide_choice, h3_nt4_choice), and callers size one workspace for calls of up to n_rows rows
  int64_t b = 0;
  for (int layout = 0; layout < 3; ++layout)
    if (d.variant == 0 ? (layout == 0 ? h3_narrow_ok(d, n_atoms) : layout == 1 ? h3_wide_ok(n_atoms) : h3_nt4_ok(d, n_atoms, false))
                       : (layout == 0 ? !h3_nt4_ok(d, n_atoms, false) : layout == 2 && h3_nt4_ok(d, n_atoms, false))) {
      const int64_t x = h3_ws(d, n_rows, n_atoms, nullptr, false, layout).bytes;
      if (x > b) b = x;
    }
  return b;
}

static int h3_launch(const FlowArgs& a, const RawLayout& L, const FusedGeom& fg, int c, int net_sel, const float* z_other,
                     const char* sfrag, int64_t sf_variant_bytes, bool shared, float* s_out, float* t_out, float* dump,
                     const PrevCoupling& prev = PrevCoupling{}, bool dry = false) {
  const tw_flow_desc& d = *a.desc;
  const bool h1 = a.h1 != 0;
  const H3Geom g = h3_geom(d, h1);
  H3Params p;
  p.packed = (const char*)a.packed + (int64_t)(c * 2) * g.net_stride_bytes;
  p.net_stride_bytes = g.net_stride_bytes;
  p.stages = g.stages;
  p.side_in2b = g.side_in2b;
  p.side_layers = g.side_layers;
  p.side_layer_size = g.side_layer_size;
  p.side_out2b = g.side_out2b;
  p.side_scales = g.side_scales;
  p.emb = a.raw + L.emb;
  p.types = a.atom_types;
  p.xc = a.x_coords;
  p.xv = a.x_velocs;
  p.z_other = z_other;
  p.sfrag = sfrag;
  p.sfrag_shared = shared ? 1 : 0;
  p.sf_variant_bytes = sf_variant_bytes;
  H3Wide wd{};
  const bool wide = h3_wide_choice(d, a.n_atoms, a.n_rows, h1);
  if (wide) h3_wide_geom(a.n_atoms, &wd);
  p.d_rff = d.variant == 1 ? d.d_rff : 0;
  p.rff = p.d_rff > 0 ? a.raw + L.chain + (int64_t)c * L.coupling_size + L.rff : nullptr;
  p.windowed = (!wide && h3_windowed(fg, a.n_atoms)) ? 1 : 0;
  for (int i = 0; i < 4; ++i) p.win[i] = wide ? wd.win[i] : 0;
  p.out[0] = s_out;
  p.out[1] = t_out;
  p.dump = dump;
  p.n_rows = a.n_rows;
  p.n_cond = a.n_cond;
  p.V = a.n_atoms;
  p.P = wide ? wd.stride : a.n_atoms;
  p.ng = wide ? wd.ng : 0;
  p.mpw = fg.mpw;
  p.nblocks = (int)((a.n_rows + fg.mpw - 1) / fg.mpw);
  p.H = d.n_heads;
  p.n_layers = d.n_layers;
  p.ff_chunks = g.ff_chunks;
  p.hid_chunks = g.hid_chunks;
  p.d_emb = d.d_emb;
  p.eps = d.ln_eps;
  p.net_sel = net_sel;
  p.debug = g_debug_flags;
  p.masked = a.masked;
  p.prev = prev;
  const int wgs_per_net = wide ? p.nblocks : (p.nblocks + 3) / 4;  // wide: one block per workgroup
  unsigned grid = net_sel < 0 ? 8u * (unsigned)((wgs_per_net + 3) / 4) : (unsigned)wgs_per_net;
  int prc;
  if (!dry && (prc = profile_mark(a.stream, true))) return prc;
  // One launch form for every instantiation: raise the kernel's dynamic-LDS limit once per device, note its name for
  // tw_last_netblock_kernel (bench.py reports - and looks its PMC traffic up by - the instantiation that really ran), launch.
  // Template arguments: <NT, ASM, DENSE, WIDE, RFF, ENC, H1, NG6>, all eight spelled out so that the name is rocprofv3's.
#define H3_STR2(...) #__VA_ARGS__
#define H3_STR(...) H3_STR2(__VA_ARGS__)
#define H3_LAUNCH(LDS, ...)                                                                                        \
  do {                                                                                                             \
    note_netblock_kernel("tw::netblock_h3_kernel<" H3_STR(__VA_ARGS__) ">");                                       \
    if (dry) return TW_OK; /* h3_selected_kernel: the name of the instantiation this call WOULD launch */          \
    static LdsLimit lim;                                                                                           \
    if ((prc = lim.ensure((const void*)netblock_h3_kernel<__VA_ARGS__>, (int)(LDS)))) return prc;                  \
    hipLaunchKernelGGL((netblock_h3_kernel<__VA_ARGS__>), dim3(grid), dim3(256), (LDS), a.stream, p);              \
  } while (0)
  // the encoder-stack statements (no compiled glue, no scratch) unless activations / section stamps between the sections are
  // asked for (bit 12: per-section build, bit 13: encoder stack anyway; bit 3: the compiled-C++ statement of the kernel)
  const bool per_section = ((dump != nullptr || (g_debug_flags & (4 | 16 | 4096))) && !(g_debug_flags & 8192)) || d.n_layers < 1;
  const bool cpp = (g_debug_flags & 8) != 0;
  if (wide && wd.nt == H3N4_NT) {
    // r05: the paired layout - one molecule of 97 .. 128 atoms per pair of 64-token waves (encoder-stack statement only)
    TW_REQUIRE(d.variant == 0 && !per_section && !cpp, "paired 64-token layout: the encoder-stack statement only");
    if (h1) H3_LAUNCH(H3N4_LDS_BYTES, 4, true, false, true, false, true, true, false);
    else H3_LAUNCH(H3N4_LDS_BYTES, 4, true, false, true, false, true, false, false);
  } else if (!wide && fg.nt == H3N4_NT && d.variant == 1) {
    // r05: the dense softmax model on one molecule of 49-64 atoms per wave (MLP sections asm, attention block compiled C++)
    // r06: the encoder-stack statement (tools/gen_h3_enc_asm.py --dense --nt=4; no compiled block, no scratch) unless activation
    // dumps / section stamps between the sections are asked for (the per-section build: attention block compiled C++)
    TW_REQUIRE(!h1 && d.d_rff == 0 && !cpp, "dense model on 64-token waves: the split-fp16 build without position features");
    if (per_section) H3_LAUNCH(H3D4_LDS_BYTES, 4, true, true, false, false, false, false, false);
    else H3_LAUNCH(H3D4_LDS_BYTES, 4, true, true, false, false, true, false, false);
  } else if (!wide && fg.nt == H3N4_NT) {
    TW_REQUIRE(d.variant == 0, "64-token waves: kernel attention");
    if (h1 && per_section) H3_LAUNCH(H3N4_LDS_BYTES, 4, true, false, false, false, false, true, false);
    else if (h1) H3_LAUNCH(H3N4_LDS_BYTES, 4, true, false, false, false, true, true, false);
    else if (cpp) H3_LAUNCH(H3N4_LDS_BYTES, 4, false, false, false, false, false, false, false);
    else if (per_section) H3_LAUNCH(H3N4_LDS_BYTES, 4, true, false, false, false, false, false, false);
    else H3_LAUNCH(H3N4_LDS_BYTES, 4, true, false, false, false, true, false, false);
  } else if (h1) {
    // single-MFMA build: the encoder-stack statements (section stamps compiled in; no activation dumps), or the per-section builds
    TW_REQUIRE(h1_supported(d, a.n_atoms) && d.n_layers >= 1, "single-MFMA path: unsupported configuration");
    if (d.variant == 1 && d.d_rff > 0) {
      // r05: position features - the in-MLP as compiled C++ in split form (192 input columns), everything behind it single-MFMA
      TW_REQUIRE(!per_section, "single-MFMA path with position features: the encoder-stack statement only");
      H3_LAUNCH(H3D_ENC_LDS_BYTES, 3, true, true, false, true, true, true, false);
    } else if (d.variant == 1) {
      if (per_section) H3_LAUNCH(H3D_LDS_BYTES, 3, true, true, false, false, false, true, false);
      else H3_LAUNCH(H3D_ENC_LDS_BYTES, 3, true, true, false, false, true, true, false);
    } else if (wide) {
      // r05: the wide layout's encoder stack as one statement (tools/gen_h3_enc_asm.py --wide [--ng=3|6] --h1)
      if (per_section && wd.ng == H3W_NG6) H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, false, true, true);
      else if (per_section) H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, false, true, false);
      else if (wd.ng == H3W_NG6) H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, true, true, true);
      else H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, true, true, false);
    } else {
      TW_REQUIRE(dump == nullptr || (g_debug_flags & 16), "single-MFMA path: no activation dumps (section stamps only)");
      H3_LAUNCH(H3_R6(3, false, false, false, true, true) ? H3_R6_LDS_BYTES : H3_LDS_BYTES, 3, true, false, false, false, true, true, false);
    }
  } else if (wide) {
    // r05: the wide layout's encoder stack as one statement (tools/gen_h3_enc_asm.py --wide [--ng=3|6]): no compiled glue, no scratch
    if (per_section && wd.ng == H3W_NG6) H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, false, false, true);
    else if (per_section) H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, false, false, false);
    else if (wd.ng == H3W_NG6) H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, true, false, true);
    else H3_LAUNCH(H3W_LDS_BYTES, 3, true, false, true, false, true, false, false);
  } else if (d.variant == 1 && d.d_rff > 0) {
    // the dense model's encoder stack as one statement (tools/gen_h3_enc_asm.py --dense); the in-MLP of the position features
    // stays compiled C++
    if (cpp) H3_LAUNCH(H3D_LDS_BYTES, 3, false, true, false, true, false, false, false);
    else if (per_section) H3_LAUNCH(H3D_LDS_BYTES, 3, true, true, false, true, false, false, false);
    else H3_LAUNCH(H3D_ENC_LDS_BYTES, 3, true, true, false, true, true, false, false);
  } else if (d.variant == 1) {
    if (cpp) H3_LAUNCH(H3D_LDS_BYTES, 3, false, true, false, false, false, false, false);
    else if (per_section) H3_LAUNCH(H3D_LDS_BYTES, 3, true, true, false, false, false, false, false);
    else H3_LAUNCH(H3D_ENC_LDS_BYTES, 3, true, true, false, false, true, false, false);
  } else if (cpp) {
    H3_LAUNCH(H3_LDS_BYTES, 3, false, false, false, false, false, false, false);
  } else if (per_section) {
    // activation dumps / section stamps live between the sections; bit 12 (4096): A/B switch for the encoder-stack build;
    // bit 13 (8192): the encoder-stack build even with a dump buffer (only the stamps / dumps outside the stack are written)
    H3_LAUNCH(H3_LDS_BYTES, 3, true, false, false, false, false, false, false);
  } else {
    H3_LAUNCH(H3_LDS_BYTES, 3, true, false, false, false, true, false, false);
  }
#undef H3_LAUNCH
#undef H3_STR
#undef H3_STR2
  TW_LAUNCH_CHECK();
  if ((prc = profile_mark(a.stream, false))) return prc;
  return TW_OK;
}

// Score fragments of coupling layer c.  With the Gaussian basis they do not depend on c (one call per flow pass);
// chebyshev_kernel needs them per coupling layer: 2 * n_layers variants, `*variant_bytes` apart (0 if one variant).
static int h3_score_frags(const FlowArgs& a, const RawLayout& L, const FusedGeom& fg, const H3Ws& w, bool shared, int c,
                          int64_t* variant_bytes) {
  const tw_flow_desc& d = *a.desc;
  const int V = a.n_atoms;
  const int64_t nblocks = shared ? 1 : (a.n_rows + fg.mpw - 1) / fg.mpw;
  const ScoreBasis basis = score_basis(d, L, a.raw, c);
  H3Wide wd;
  if (h3_wide_choice(d, V, a.n_rows, a.h1 != 0)) {
    h3_wide_geom(V, &wd);
    const int64_t vbw = basis.n_variants > 1 ? nblocks * 4 * d.n_heads * h3w_frag_head(wd) : 0;
    const float* lsw = a.raw + L.lengthscales + (a.reverse ? d.n_heads : 0);
    const size_t shmw = h3w_sf_lds_bytes(V, wd.mpwg);
    TW_REQUIRE(shmw <= H3_SF_LDS_MAX, "score fragments: %zu bytes of LDS for %d atoms", shmw, V);
    if (shmw > (size_t)64 * 1024) {
      static LdsLimit limw;
      int lrc;
      if ((lrc = limw.ensure((const void*)h3w_score_frag_kernel, (int)H3_SF_LDS_MAX))) return lrc;
    }
    H3WideWin win;
    for (int i = 0; i < 4; ++i) win.w[i] = wd.win[i];
    hipLaunchKernelGGL(h3w_score_frag_kernel, dim3((unsigned)nblocks, (unsigned)d.n_heads, (unsigned)basis.n_variants), dim3(512),
                       shmw, a.stream, a.x_coords, a.masked, lsw, d.n_heads, V, wd.stride, wd.ng, wd.mpwg, a.n_rows, a.n_cond, d.normalise, w.sfrag,
                       basis, vbw, V > 25 ? 1 : 0, win, wd.nt);
    TW_LAUNCH_CHECK();
    *variant_bytes = vbw;
    return TW_OK;
  }
  const int64_t vb = basis.n_variants > 1 ? nblocks * d.n_heads * fg.nt * (fg.nt == 4 ? H3N4_SF_BYTES : H3_SF_BYTES) : 0;
  const unsigned nv = (unsigned)basis.n_variants;
  const float* ls = a.raw + L.lengthscales + (a.reverse ? d.n_heads : 0);
  const int win = h3_windowed(fg, V) ? 1 : 0;
  const size_t shm = h3_sf_lds_bytes(d.n_heads, V, fg.mpw);
  TW_REQUIRE(shm <= H3_SF_LDS_MAX, "score fragments: %zu bytes of LDS for %d atoms x %d heads", shm, V, d.n_heads);
  if (shm > (size_t)64 * 1024) {
    static LdsLimit lim;
    int lrc;
    if ((lrc = lim.ensure((const void*)h3_score_frag_kernel, (int)H3_SF_LDS_MAX))) return lrc;
  }
  // a lone block (all proposals share x) is latency-bound: spread it over 16 waves
  hipLaunchKernelGGL(h3_score_frag_kernel, dim3((unsigned)nblocks, 1, nv), dim3(shared ? 1024 : 512), shm, a.stream, a.x_coords,
                     a.masked, ls, d.n_heads, V, fg.mpw, a.n_rows, a.n_cond, d.normalise, w.sfrag, basis, vb, win, V > 25 ? 1 : 0,
                     fg.nt);
  TW_LAUNCH_CHECK();
  *variant_bytes = vb;
  return TW_OK;
}

int flow_pass_h3(const FlowArgs& a) {
  const tw_flow_desc& d = *a.desc;
  NoPairScope no_pair(d.n_layers < 1);  // a model without encoder layers runs the per-section build: 48-token wide layout
  FusedGeom fg;
  H3Wide wdg;
  bool wide_layout = false;
  TW_REQUIRE(h3_layout(d, a.n_atoms, a.n_rows, a.h1 != 0, &fg, &wdg, &wide_layout) && !(wide_layout && d.variant != 0),
             "split-fp16 path: unsupported atom count %d", a.n_atoms);
  const RawLayout L = raw_layout(d);
  const H3Ws w = h3_ws(d, a.n_rows, a.n_atoms, a.ws, a.h1 != 0);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  const bool shared = a.n_cond == 1;
  int rc;
  int64_t vb = 0;
  const bool kernel_attention = d.variant == 0;  // the dense variant computes its scores inside the net-block launch
  if (kernel_attention && d.cheb_order == 0 && (rc = h3_score_frags(a, L, fg, w, shared, 0, &vb))) return rc;
  // The coupling update of layer i is applied by the launch of layer i + 1 (PrevCoupling): the variable it transforms
  // is that launch's conditioning input.  Both variables alternate between the caller's buffer and a workspace copy,
  // the (s, t) outputs between two sets: a launch reads what its predecessor wrote while it writes its own.
  float* caller[2] = {a.z_coords, a.z_velocs};
  float* cur[2] = {a.z_coords, a.z_velocs};
  float* alt[2] = {w.zc_alt, w.zv_alt};
  float* sbuf[2] = {w.s_out, w.s_out2};
  float* tbuf[2] = {w.t_out, w.t_out2};
  int* flag = flow_range_flag(d);
  TW_REQUIRE(flag != nullptr, "hipGetSymbolAddress(g_nonfinite) failed");
  PrevCoupling prev{};
  int tv = 0;
  if (g_debug_flags & 32) {  // A/B switch: every coupling update as its own launch (the r01 sequence)
    for (int i = 0; i < d.n_coupling; ++i) {
      const int c = a.reverse ? d.n_coupling - 1 - i : i;
      const bool positions = (c % 2) == d.pos_mod2;
      if (kernel_attention && d.cheb_order > 0 && (rc = h3_score_frags(a, L, fg, w, shared, c, &vb))) return rc;
      if ((rc = h3_launch(a, L, fg, c, -1, positions ? a.z_velocs : a.z_coords, w.sfrag, vb, shared, w.s_out, w.t_out, nullptr)))
        return rc;
      if ((rc = launch_coupling(w.s_out, w.t_out, a.masked, a.n_cond, positions ? a.z_coords : a.z_velocs, a.delta_logp,
                                a.n_rows, a.n_atoms, a.reverse, a.stream, nullptr, flag)))
        return rc;
    }
    return TW_OK;
  }
  for (int i = 0; i < d.n_coupling; ++i) {
    const int c = a.reverse ? d.n_coupling - 1 - i : i;
    const bool positions = (c % 2) == d.pos_mod2;
    tv = positions ? 0 : 1;     // the variable this layer transforms
    const int ov = 1 - tv;      // its conditioning input - the variable the previous layer transformed
    if (prev.s_raw) {
      prev.z_in = cur[ov];
      prev.z_out = alt[ov];
    }
    if (kernel_attention && d.cheb_order > 0 && (rc = h3_score_frags(a, L, fg, w, shared, c, &vb))) return rc;
    if ((rc = h3_launch(a, L, fg, c, -1, cur[ov], w.sfrag, vb, shared, sbuf[i & 1], tbuf[i & 1], nullptr, prev))) return rc;
    if (prev.s_raw) std::swap(cur[ov], alt[ov]);
    prev = PrevCoupling{sbuf[i & 1], tbuf[i & 1], nullptr, nullptr, a.delta_logp, flag, a.reverse};
  }
  // the last layer's update has no successor: its own small launch, straight into the caller's buffer
  if ((rc = launch_coupling(prev.s_raw, prev.t, a.masked, a.n_cond, caller[tv], a.delta_logp, a.n_rows, a.n_atoms, a.reverse,
                            a.stream, cur[tv], flag)))
    return rc;
  const int ov = 1 - tv;
  if (cur[ov] != caller[ov])
    TW_HIP_CHECK(hipMemcpyAsync(caller[ov], cur[ov], (size_t)a.n_rows * a.n_atoms * 3 * sizeof(float), hipMemcpyDeviceToDevice,
                                a.stream));
  return TW_OK;
}

// TW_PATH_SIMPLE_H3: can the FFN run through the split-fp16 stream (what tw_flow_pack_h3 packs: d_model 128, chunks of 32)?
bool h3_ffn_tokens_supported(const tw_flow_desc& d) { return h3_supported(d, 22); }

int h3_ffn_tokens(const tw_flow_desc& d, const void* packed, int coupling, int net, int layer, float* h, int64_t n_tokens,
                  hipStream_t stream, float* scratch, int64_t scratch_floats, const void* split_stages) {
  TW_REQUIRE(packed && h3_ffn_tokens_supported(d), "FFN on the split-fp16 stream: unsupported model or no stream");
  const H3Geom g = h3_geom(d);
  const int64_t att_stages = d.variant == 1 ? 4LL * g.H : 8LL * g.H;
  const int64_t first = (int64_t)(g.in_a_stages + 2) * g.hid_chunks + (int64_t)layer * (att_stages + 4LL * g.ff_chunks) + att_stages;
  const char* net_base = (const char*)packed + (int64_t)(coupling * 2 + net) * g.net_stride_bytes;
  H3FfnParams p;
  p.stages = net_base + first * H3_STAGE_BYTES;
  p.side = (const float*)(net_base + g.stages * H3_STAGE_BYTES) + g.side_layers + (int64_t)layer * g.side_layer_size;
  p.h = h;
  p.n_tokens = n_tokens;
  p.ff_chunks = g.ff_chunks;
  p.eps = d.ln_eps;
  p.split = 1;
  p.parts = scratch;
  // One workgroup per CU (LDS), so a launch is whole rounds of the chip: 48-token waves (192 tokens per workgroup) unless 64-token
  // waves (256 per workgroup, 4/3 the time per round) need fewer rounds' worth - 200 atoms x 256 rows: 267 workgroups = 2 rounds
  // against 200 = 4/3; 256 x 256: 342 = 2 rounds against 256 = 4/3.  Bits 29 / 30 force either (A/B, tests).
  const int64_t wg3 = (n_tokens + 191) / 192, wg4 = (n_tokens + 255) / 256;
  const int cus = H3_CUS;
  bool four = 4 * ((wg4 + cus - 1) / cus) < 3 * ((wg3 + cus - 1) / cus);
  if (g_debug_flags & 536870912) four = false;
  if (g_debug_flags & 1073741824) four = true;
  int rc;
  // Launches that fill less than half the chip (691 atoms x 16 rows: 58 workgroups, each 84 us behind the weight stream): the hidden
  // layer over H3_FFN_SPLIT workgroups per token tile (their own streams: h3_ffn_split_pack), W2-scaled partial sums through `scratch`,
  // the layer finished by a small launch (bit 29: never)
  if (!four && scratch && split_stages && h3_ffn_split_bytes(d) && wg3 * 2 <= cus &&
      (int64_t)H3_FFN_SPLIT * n_tokens * 128 <= scratch_floats && !(g_debug_flags & 536870912)) {
    p.split = H3_FFN_SPLIT;
    p.ff_chunks = g.ff_chunks / H3_FFN_SPLIT;
    p.stages = (const char*)split_stages + (((int64_t)(coupling * 2 + net) * g.L + layer) * g.ff_chunks * 4) * H3_STAGE_BYTES;
  }
  const int64_t wgs = (four ? wg4 : wg3) * p.split;
  TW_REQUIRE(wgs < (int64_t)1 << 31, "FFN: %lld workgroups", (long long)wgs);
  if (four) {
    constexpr int lds = H3N4_RING * H3_STAGE_BYTES + 4 * H3N4_WAVE_LDS;
    static LdsLimit lim;
    if ((rc = lim.ensure((const void*)h3_ffn_tokens_kernel<4>, lds))) return rc;
    hipLaunchKernelGGL(h3_ffn_tokens_kernel<4>, dim3((unsigned)wgs), dim3(256), lds, stream, p);
  } else {
    constexpr int lds = H3_RING * H3_STAGE_BYTES + 4 * H3_WAVE_LDS;
    static LdsLimit lim;
    if ((rc = lim.ensure((const void*)h3_ffn_tokens_kernel<3>, lds))) return rc;
    hipLaunchKernelGGL(h3_ffn_tokens_kernel<3>, dim3((unsigned)wgs), dim3(256), lds, stream, p);
  }
  TW_LAUNCH_CHECK();
  if (p.split > 1) {
    hipLaunchKernelGGL(h3_ffn_parts_ln_kernel, dim3((unsigned)((n_tokens + 3) / 4)), dim3(256), 0, stream, p);
    TW_LAUNCH_CHECK();
  }
  return TW_OK;
}

// TW_PATH_SIMPLE_H3, small launches: the FFN stages of every (coupling, net, layer) once more as H3_FFN_SPLIT self-contained streams of
// ff_chunks / H3_FFN_SPLIT chunks each.  The stream is software-pipelined - A(0) | A(1) B(0) | ... | B(n - 1), A = the two W1 stages of a
// chunk, B = the two W2 stages - so a quarter of it is not a stream; its stages in the pipelined order of 16 chunks are.  Whole stages
// move (9 KiB each, aux included): a gather on the device behind tw_flow_pack_h3.
__global__ void __launch_bounds__(256) h3_ffn_split_gather_kernel(const char* __restrict__ packed, char* __restrict__ dst, int64_t net_stride,
                                                                  int64_t first_ffn_stage, int64_t layer_stages, int n_layers, int chunks) {
  // blockIdx.x = ((net_index * n_layers + l) * chunks + ch) * 4 + piece;  piece 0, 1: the A stages of chunk ch, 2, 3: its B stages
  int64_t b = blockIdx.x;
  const int piece = (int)(b % 4); b /= 4;
  const int ch = (int)(b % chunks); b /= chunks;
  const int l = (int)(b % n_layers);
  const int64_t ni = b / n_layers;
  const int per = chunks / H3_FFN_SPLIT, part = ch / per, c = ch % per;
  auto a_off = [](int k) -> int64_t { return k == 0 ? 0 : 2 + (int64_t)(k - 1) * 4; };
  auto b_off = [](int k, int n) -> int64_t { return 2 + (int64_t)k * 4 + (k < n - 1 ? 2 : 0); };
  const int64_t src_stage = first_ffn_stage + l * layer_stages + (piece < 2 ? a_off(ch) + piece : b_off(ch, chunks) + piece - 2);
  const int64_t dst_stage = ((ni * n_layers + l) * chunks + (int64_t)part * per) * 4 + (piece < 2 ? a_off(c) + piece : b_off(c, per) + piece - 2);
  const uint4* sp = (const uint4*)(packed + ni * net_stride + src_stage * H3_STAGE_BYTES);
  uint4* dp = (uint4*)(dst + dst_stage * H3_STAGE_BYTES);
  for (int i = threadIdx.x; i < H3_STAGE_BYTES / 16; i += 256) dp[i] = sp[i];
}

int64_t h3_ffn_split_bytes(const tw_flow_desc& d) {
  if (!h3_ffn_tokens_supported(d)) return 0;
  const H3Geom g = h3_geom(d);
  if (g.ff_chunks % H3_FFN_SPLIT) return 0;
  return ((int64_t)d.n_coupling * 2 * g.L * g.ff_chunks * 4 + H3_RING + 1) * H3_STAGE_BYTES;   // + the statements' over-fetch
}

int h3_ffn_split_pack(const tw_flow_desc& d, const void* packed, void* dst, hipStream_t stream) {
  if (!h3_ffn_split_bytes(d)) return TW_OK;
  const H3Geom g = h3_geom(d);
  const int64_t att_stages = d.variant == 1 ? 4LL * g.H : 8LL * g.H;
  const int64_t blocks = (int64_t)d.n_coupling * 2 * g.L * g.ff_chunks * 4;
  TW_REQUIRE(blocks < (int64_t)1 << 31, "FFN split pack: %lld stages", (long long)blocks);
  TW_HIP_CHECK(hipMemsetAsync((char*)dst + blocks * H3_STAGE_BYTES, 0, (H3_RING + 1) * H3_STAGE_BYTES, stream));
  hipLaunchKernelGGL(h3_ffn_split_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const char*)packed, (char*)dst,
                     g.net_stride_bytes, (int64_t)(g.in_a_stages + 2) * g.hid_chunks + att_stages, att_stages + 4LL * g.ff_chunks, g.L,
                     g.ff_chunks);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

// in-MLP (out == false) / out-MLP of (coupling, net) on the flat token list; d_in <= 64 (no position features)
bool h3_io_tokens_supported(const tw_flow_desc& d) {
  return h3_ffn_tokens_supported(d) && d.d_emb + 9 <= 64 && !(d.variant == 1 && d.d_rff > 0);
}

int h3_io_tokens(const tw_flow_desc& d, const void* packed, int coupling, int net, bool out, const float* in, float* res, int d_in,
                 int64_t n_tokens, hipStream_t stream) {
  TW_REQUIRE(packed && h3_io_tokens_supported(d) && d_in <= 64, "in / out MLP on the split-fp16 stream: unsupported model or no stream");
  const H3Geom g = h3_geom(d);
  const int64_t att_stages = d.variant == 1 ? 4LL * g.H : 8LL * g.H;
  const int64_t first = out ? (int64_t)(g.in_a_stages + 2) * g.hid_chunks + (int64_t)g.L * (att_stages + 4LL * g.ff_chunks) : 0;
  const char* net_base = (const char*)packed + (int64_t)(coupling * 2 + net) * g.net_stride_bytes;
  const float* side = (const float*)(net_base + g.stages * H3_STAGE_BYTES);
  H3IoParams p;
  p.stages = net_base + first * H3_STAGE_BYTES;
  p.bias2 = side + (out ? g.side_out2b : g.side_in2b);
  p.scale2 = side + g.side_scales + (out ? 2 + 3 * g.L + 1 : 1);   // scales: in0, in2, per layer (wc, w1, w2), out0, out2
  p.in = in;
  p.out = res;
  p.n_tokens = n_tokens;
  p.d_in = d_in;
  p.chunks = g.hid_chunks;
  // (48- or 64-token waves: whichever wastes less of the last round of the chip, as h3_ffn_tokens)
  const int64_t wg3 = (n_tokens + 191) / 192, wg4 = (n_tokens + 255) / 256;
  bool four = 4 * ((wg4 + H3_CUS - 1) / H3_CUS) < 3 * ((wg3 + H3_CUS - 1) / H3_CUS);
  if (g_debug_flags & 536870912) four = false;
  if (g_debug_flags & 1073741824) four = true;
  const int64_t wgs = four ? wg4 : wg3;
  TW_REQUIRE(wgs < (int64_t)1 << 31, "in / out MLP: %lld workgroups", (long long)wgs);
  constexpr int lds3 = H3_RING * H3_STAGE_BYTES + 4 * H3_WAVE_LDS, lds4 = H3N4_RING * H3_STAGE_BYTES + 4 * H3N4_WAVE_LDS;
  static LdsLimit lim[4];
  const void* fn[4] = {(const void*)h3_io_tokens_kernel<false, 3>, (const void*)h3_io_tokens_kernel<false, 4>,
                       (const void*)h3_io_tokens_kernel<true, 3>, (const void*)h3_io_tokens_kernel<true, 4>};
  const int which = (out ? 2 : 0) + (four ? 1 : 0);
  const int lds = four ? lds4 : lds3;
  int rc;
  if ((rc = lim[which].ensure(fn[which], lds))) return rc;
  switch (which) {
    case 0: hipLaunchKernelGGL((h3_io_tokens_kernel<false, 3>), dim3((unsigned)wgs), dim3(256), lds, stream, p); break;
    case 1: hipLaunchKernelGGL((h3_io_tokens_kernel<false, 4>), dim3((unsigned)wgs), dim3(256), lds, stream, p); break;
    case 2: hipLaunchKernelGGL((h3_io_tokens_kernel<true, 3>), dim3((unsigned)wgs), dim3(256), lds, stream, p); break;
    default: hipLaunchKernelGGL((h3_io_tokens_kernel<true, 4>), dim3((unsigned)wgs), dim3(256), lds, stream, p); break;
  }
  TW_LAUNCH_CHECK();
  return TW_OK;
}

// The instantiation a flow pass over n_rows x n_atoms would launch on the split-fp16 (h1 = false) or single-MFMA path, under the
// debug flags in force - the launch code's own branch, run dry (tw_flow_selected_kernel: the spill guard of the CPU suite asks
// this for every size instead of guessing which instantiations are "product").
int h3_selected_kernel(const tw_flow_desc& d, int n_atoms, int64_t n_rows, bool h1) {
  NoPairScope no_pair(d.n_layers < 1);
  FlowArgs a{};
  a.desc = &d;
  a.n_rows = n_rows;
  a.n_cond = 1;
  a.n_atoms = n_atoms;
  a.h1 = h1 ? 1 : 0;
  FusedGeom fg;
  H3Wide wdg;
  bool wide_layout = false;
  TW_REQUIRE(h3_layout(d, n_atoms, n_rows, h1, &fg, &wdg, &wide_layout) && !(wide_layout && d.variant != 0),
             "split-fp16 path: unsupported atom count %d", n_atoms);
  const RawLayout L = raw_layout(d);
  return h3_launch(a, L, fg, 0, -1, nullptr, nullptr, 0, true, nullptr, nullptr, nullptr, PrevCoupling{}, true);
}

int debug_netblock_h3(const FlowArgs& a, int c, int net, const float* z_other, float* dump) {
  const tw_flow_desc& d = *a.desc;
  NoPairScope no_pair;  // activation dumps / section stamps live in the per-section builds, which the paired layout has none of
  FusedGeom fg;
  H3Wide wdg;
  bool wide_layout = false;
  TW_REQUIRE(h3_layout(d, a.n_atoms, a.n_rows, a.h1 != 0, &fg, &wdg, &wide_layout) && !(wide_layout && d.variant != 0),
             "split-fp16 path: unsupported atom count %d", a.n_atoms);
  const RawLayout L = raw_layout(d);
  const H3Ws w = h3_ws(d, a.n_rows, a.n_atoms, a.ws, a.h1 != 0);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  const bool shared = a.n_cond == 1;
  int rc;
  int64_t vb = 0;
  if (d.variant == 0 && (rc = h3_score_frags(a, L, fg, w, shared, c, &vb))) return rc;
  return h3_launch(a, L, fg, c, net, z_other, w.sfrag, vb, shared, w.s_out, w.t_out, dump);
}

}  // namespace tw
