// Fused f32-MFMA net-block kernel for the kernel-attention flow (gfx950 / CDNA4).
//
// One launch evaluates BOTH coupling nets (scale_transformer, shift_transformer) of one coupling
// layer for every conformation: in_mlp -> L x [kernel self-attention, +res, LN, FFN, +res, LN] ->
// out_mlp (custom_transformer_block.py:46-82 and everything below it).  Design:
//
//  * transposed formulation  Y^T = W . X^T  on v_mfma_f32_16x16x4_f32 (exact fp32): weights are the
//    MFMA A operand, streamed from L2 as pre-packed 1 KiB fragment tiles in consumption order;
//    activations are the B operand and NEVER leave registers: the MFMA D layout of one stage
//    (lane = token, lane-group/reg = feature) is exactly the B layout the next stage needs once the
//    weight tiles are packed with the matching k-permutation.
//  * one WAVE owns 16*NT tokens = MPW whole molecules; no inter-wave communication, no barriers.
//  * the FFN (69 % of the FLOPs) is chained in registers: the 2048-wide hidden layer is produced
//    and consumed 32 units at a time and never materialised.
//  * kernel attention is folded: sum_h A_h X (W_o,h W_v,h)^T, the 128x128 products precomputed in
//    fp64 at pack time; the per-head mixing A_h X is an MFMA against block-diagonal score
//    fragments, with X transposed through a wave-private LDS tile.
//  * blockIdx -> (net, block) is XCD-aware: XCDs 0-3 stream the scale net, 4-7 the shift net.
#include <stdlib.h>
#include <utility>
#include <vector>

#include "tw_common.h"
#include "tw_nb_f32.h"

namespace tw {

bool fused_geom_nt(int V, int nt, FusedGeom* g) {
  if (V <= 0 || nt <= 0 || V > 16 * nt) return false;
  g->nt = nt;
  g->mpw = (16 * nt) / V;
  int mask = 0;
  for (int q = 0; q < g->mpw; ++q) {
    int t0 = (q * V) / 16, t1 = ((q + 1) * V - 1) / 16;
    for (int a = t0; a <= t1; ++a)
      for (int b = t0; b <= t1; ++b) mask |= 1 << (a * nt + b);
  }
  g->tile_mask = mask;
  return true;
}

// the better filled of 3 and 4 token tiles per wave (what the f32 kernels run)
bool fused_geom(int V, FusedGeom* g) {
  if (V <= 0 || V > 64) return false;
  int best_nt = 0, best_num = -1, best_den = 1;
  for (int nt = 3; nt <= 4; ++nt) {
    int mpw = (16 * nt) / V;
    if (mpw == 0) continue;
    int num = mpw * V, den = 16 * nt;  // utilisation num/den
    if (best_nt == 0 || (int64_t)num * best_den > (int64_t)best_num * den) {
      best_nt = nt; best_num = num; best_den = den;
    }
  }
  if (best_nt == 0) return false;
  return fused_geom_nt(V, best_nt, g);
}

// ---- weight stream geometry (floats), shared by the packer and the kernel ------------------------
// per net:  IN stage  : hid_chunks x 24 tiles  (W0 chunk 2x3, W2 chunk 8x2, 2 pad)
//           per layer : H x 64 tiles (folded W_o,h W_v,h)  then ff_chunks x 32 tiles (W1 2x8, W2 8x2)
//           OUT stage : hid_chunks x 24 tiles  (W0 chunk 2x8, W2 chunk 1x2, 6 pad)
//           side      : in0_b[hid] in2_b[128] { n1w n1b [128] b1[ff] b2[128] n2w n2b [128] } out0_b[hid] out2_b[16]
struct StreamGeom {
  int hid_chunks, ff_chunks, H, L;
  int64_t in_tiles, layer_tiles, out_tiles, tiles;
  int64_t side_in0b, side_in2b, side_layers, side_layer_size, side_out0b, side_out2b, side_size;
};

static StreamGeom stream_geom(const tw_flow_desc& d) {
  StreamGeom s;
  s.hid_chunks = d.d_hidden / 32;
  s.ff_chunks = d.d_ff / 32;
  s.H = d.n_heads;
  s.L = d.n_layers;
  s.in_tiles = (int64_t)s.hid_chunks * 24;
  s.layer_tiles = (int64_t)s.H * 64 + (int64_t)s.ff_chunks * 32;
  s.out_tiles = (int64_t)s.hid_chunks * 24;
  s.tiles = s.in_tiles + s.L * s.layer_tiles + s.out_tiles;
  int64_t o = 0;
  s.side_in0b = o; o += d.d_hidden;
  s.side_in2b = o; o += 128;
  s.side_layers = o;
  s.side_layer_size = 128 * 5 + d.d_ff;
  o += s.L * s.side_layer_size;
  s.side_out0b = o; o += d.d_hidden;
  s.side_out2b = o; o += 16;
  s.side_size = (o + 63) / 64 * 64 + 64;  // slack: the bias prefetch reads one chunk ahead
  return s;
}

PackedLayout packed_layout(const tw_flow_desc& d) {
  if (d.variant == 1) return dense_packed_layout(d);
  StreamGeom s = stream_geom(d);
  PackedLayout p;
  p.tiles_per_net = s.tiles;
  p.side_per_net = s.side_size;
  p.net_stride = (s.tiles * TILE_F + s.side_size + 255) / 256 * 256;
  p.total = p.net_stride * 2 * d.n_coupling + (int64_t)(RING + 1) * TILE_F;  // ring prefetch overrun
  return p;
}

// ================================================================================================
// packing kernels
// ================================================================================================
// folded attention weight of head h: Wc[o][i] = sum_k Wo[o][h*128+k] * Wv[h*128+k][i], fp64 accumulate
__global__ void pack_fold_kernel(const float* __restrict__ wv, const float* __restrict__ wo, int H, int h,
                                 float* __restrict__ dst) {
  const int ot = blockIdx.x, ft = blockIdx.y, lane = threadIdx.x;
  const int o_row = 16 * ot + (lane & 15);
  float* o = dst + ((int64_t)(ot * 8 + ft) * 64 + lane) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i_col = 16 * ft + 4 * (lane >> 4) + r;
    double acc = 0.0;
    for (int k = 0; k < 128; ++k)
      acc += (double)wo[(int64_t)o_row * (H * 128) + h * 128 + k] * (double)wv[(int64_t)(h * 128 + k) * 128 + i_col];
    o[r] = (float)acc;
  }
}

int pack_weights(const tw_flow_desc& d, const float* raw, float* packed, hipStream_t s) {
  if (d.variant == 1) return dense_pack_weights(d, raw, packed, s);
  const RawLayout L = raw_layout(d);
  const StreamGeom g = stream_geom(d);
  const PackedLayout P = packed_layout(d);
  TW_HIP_CHECK(hipMemsetAsync(packed, 0, P.total * sizeof(float), s));
  int rc;
  for (int c = 0; c < d.n_coupling; ++c)
    for (int net = 0; net < 2; ++net) {
      const float* nb = raw + net_base(L, c, net);
      float* pn = packed + (int64_t)(c * 2 + net) * P.net_stride;
      float* t = pn;  // tile cursor
      // IN stage: in_mlp.0 [hid, d_in] (K padded to 48), in_mlp.2 [128, hid]
      for (int ch = 0; ch < g.hid_chunks; ++ch) {
        if ((rc = pack_block(nb + L.net.in0_w, L.d_in, d.d_hidden, L.d_in, 32 * ch, 0, 2, 3, t, s))) return rc;
        if ((rc = pack_block(nb + L.net.in2_w, d.d_hidden, 128, d.d_hidden, 0, 32 * ch, 8, 2, t + 6 * TILE_F, s))) return rc;
        t += 24 * TILE_F;
      }
      for (int l = 0; l < d.n_layers; ++l) {
        const float* lb = nb + L.net.layers + (int64_t)l * L.layer.size;
        for (int h = 0; h < d.n_heads; ++h) {
          hipLaunchKernelGGL(pack_fold_kernel, dim3(8, 8), dim3(64), 0, s, lb + L.layer.wv, lb + L.layer.wo, d.n_heads, h, t);
          TW_LAUNCH_CHECK();
          t += 64 * TILE_F;
        }
        for (int ch = 0; ch < g.ff_chunks; ++ch) {
          if ((rc = pack_block(lb + L.layer.w1, 128, d.d_ff, 128, 32 * ch, 0, 2, 8, t, s))) return rc;
          if ((rc = pack_block(lb + L.layer.w2, d.d_ff, 128, d.d_ff, 0, 32 * ch, 8, 2, t + 16 * TILE_F, s))) return rc;
          t += 32 * TILE_F;
        }
      }
      for (int ch = 0; ch < g.hid_chunks; ++ch) {
        if ((rc = pack_block(nb + L.net.out0_w, 128, d.d_hidden, 128, 32 * ch, 0, 2, 8, t, s))) return rc;
        if ((rc = pack_block(nb + L.net.out2_w, d.d_hidden, 3, d.d_hidden, 0, 32 * ch, 1, 2, t + 16 * TILE_F, s))) return rc;
        t += 24 * TILE_F;
      }
      // side arrays
      float* sd = pn + g.tiles * TILE_F;
      if ((rc = copy_pad(nb + L.net.in0_b, d.d_hidden, sd + g.side_in0b, d.d_hidden, s))) return rc;
      if ((rc = copy_pad(nb + L.net.in2_b, 128, sd + g.side_in2b, 128, s))) return rc;
      for (int l = 0; l < d.n_layers; ++l) {
        const float* lb = nb + L.net.layers + (int64_t)l * L.layer.size;
        float* sl = sd + g.side_layers + (int64_t)l * g.side_layer_size;
        if ((rc = copy_pad(lb + L.layer.n1w, 128, sl, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.n1b, 128, sl + 128, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.b1, d.d_ff, sl + 256, d.d_ff, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.b2, 128, sl + 256 + d.d_ff, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.n2w, 128, sl + 384 + d.d_ff, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.n2b, 128, sl + 512 + d.d_ff, 128, s))) return rc;
      }
      if ((rc = copy_pad(nb + L.net.out0_b, d.d_hidden, sd + g.side_out0b, d.d_hidden, s))) return rc;
      if ((rc = copy_pad(nb + L.net.out2_b, 3, sd + g.side_out2b, 16, s))) return rc;
    }
  return TW_OK;
}

// ================================================================================================
// block-diagonal score fragments
// sfrag[blk][h][jt*NT+mt][lane] (float4 over s'=0..3) = S[query 16jt+(lane&15)][key 16mt+4s'+(lane>>4)]
// where S is the 16NT x 16NT block-diagonal matrix of the block's molecules' normalised RBF scores
// (kernel_attention.py:69-121); zero outside the molecules and for masked keys.
// ================================================================================================
__device__ __forceinline__ float pair_dist(const float* x, int q, int m, int use_mm) {
  float qx = x[3 * q], qy = x[3 * q + 1], qz = x[3 * q + 2];
  float mx = x[3 * m], my = x[3 * m + 1], mz = x[3 * m + 2];
  if (!use_mm) {
    float dx = qx - mx, dy = qy - my, dz = qz - mz;
    return sqrtf(dx * dx + dy * dy + dz * dz);
  }
  return tw_cdist_mm(qx, qy, qz, mx, my, mz);
}

__global__ void score_frag_kernel(const float* __restrict__ x, const uint8_t* __restrict__ masked,
                                  const float* __restrict__ ls, int H, int V, int mpw, int nt, int64_t n_rows,
                                  int64_t n_cond, int normalise, int use_mm, float* __restrict__ sfrag, ScoreBasis basis,
                                  int64_t variant_floats) {
  extern __shared__ float sm[];
  float* xs = sm;                        // [mpw][V*3]
  float* dist = xs + mpw * V * 3;        // [mpw][V*V]
  float* denom = dist + mpw * V * V;     // [mpw][H][V]
  uint8_t* msk = (uint8_t*)(denom + mpw * H * V);  // [mpw][V]
  const int64_t blk = blockIdx.x;
  for (int i = threadIdx.x; i < mpw * V * 3; i += blockDim.x) {
    const int q = i / (V * 3);
    int64_t n = blk * mpw + q;
    if (n >= n_rows) n = n_rows - 1;
    xs[i] = x[(n % n_cond) * V * 3 + i % (V * 3)];
  }
  for (int i = threadIdx.x; i < mpw * V; i += blockDim.x) {
    const int q = i / V;
    int64_t n = blk * mpw + q;
    if (n >= n_rows) n = n_rows - 1;
    msk[i] = masked[(n % n_cond) * V + i % V];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < mpw * V * V; i += blockDim.x) {
    const int q = i / (V * V), r = i % (V * V);
    dist[i] = pair_dist(xs + q * V * 3, r / V, r % V, use_mm);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < mpw * H * V; i += blockDim.x) {
    const int q = i / (H * V), h = (i / V) % H, a = i % V;
    const float l = ls[h];
    float cmean;
    const float* cf = basis_coeffs(basis, blockIdx.y, h, &cmean);  // grid.y = basis variant (chebyshev_kernel: net x layer)
    double sum = 0.0;
    for (int m = 0; m < V; ++m) {
      float sc = dist[(q * V + a) * V + m] / l;
      float e = msk[q * V + m] ? 0.f : basis_value(sc, cf, basis.order, cmean);
      sum += (double)fabsf(e);
    }
    denom[i] = (float)sum + 1e-5f;
  }
  __syncthreads();
  const int ntt = nt * nt;
  const int total = H * ntt * 64 * 4;
  float* out = sfrag + blockIdx.y * variant_floats + blk * (int64_t)total;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int sp = i & 3, lane = (i >> 2) & 63, tile = (i >> 8) % ntt, h = i / (ntt * 256);
    const int jt = tile / nt, mt = tile % nt;
    const int tq = 16 * jt + (lane & 15), tk = 16 * mt + 4 * sp + (lane >> 4);
    float val = 0.f;
    const int mq = tq / V, mk = tk / V;
    if (mq == mk && mq < mpw) {
      const int a = tq % V, m = tk % V;
      if (!msk[mq * V + m]) {
        float cmean;
        const float* cf = basis_coeffs(basis, blockIdx.y, h, &cmean);
        float sc = dist[(mq * V + a) * V + m] / ls[h];
        float e = basis_value(sc, cf, basis.order, cmean);
        val = normalise ? e / denom[(mq * H + h) * V + a] : e;
      }
    }
    out[i] = val;
  }
}

// ================================================================================================
// the net-block kernel
// ================================================================================================
struct NBParams {
  const float* packed;   // stream of net 0 of this coupling layer
  int64_t net_stride;    // floats between the two nets
  int64_t tiles_per_net;
  int64_t side_in0b, side_in2b, side_layers, side_layer_size, side_out0b, side_out2b;
  const float* emb;
  const int32_t* types;
  const float* xc;
  const float* xv;
  const float* z_other;
  const float* sfrag;
  int sfrag_shared;
  int64_t sf_variant_floats;  // chebyshev_kernel: floats between the fragment sets of (net, layer) variants; else 0
  float* out[2];
  float* dump;
  int64_t n_rows, n_cond;
  int V, mpw, nblocks, tile_mask;
  int H, n_layers, ff_chunks, hid_chunks, d_emb, d_ff;
  float eps;
  int net_sel;  // -1: both nets with the XCD split; 0/1: only that net
};

template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
netblock_kernel(const NBParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i16 = lane & 15;

  // XCD-aware mapping: block b runs on XCD b % 8 (observed; only speed depends on it)
  int net, wg;
  if (p.net_sel < 0) {
    const int xcd = blockIdx.x & 7;
    net = xcd >> 2;
    wg = (blockIdx.x >> 3) * 4 + (xcd & 3);
  } else {
    net = p.net_sel;
    wg = blockIdx.x;
  }
  const int blk = wg * 4 + wave;
  if (blk >= p.nblocks) return;

  float* xs = lds + wave * (16 * NT * XS);
  const float* net_base = p.packed + (int64_t)net * p.net_stride;
  const float* side = net_base + p.tiles_per_net * TILE_F;
  const float* wp = net_base + lane * 4;  // this lane's float4 inside tile 0
  f4 ring[RING];
#pragma unroll
  for (int i = 0; i < RING; ++i) ring[i] = *(const f4*)(wp + (int64_t)i * TILE_F);
  // invariant: wp = start of the current body; ring[k] holds tile k of it; RING_LOAD(T) fetches tile T + RING

  // ---- token bookkeeping -------------------------------------------------------------------
  int64_t tok_row[NT];  // conformation index of this lane's token in tile jt (or -1)
  int tok_atom[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int t = 16 * jt + i16;
    const int q = t / p.V;
    const int64_t n = (int64_t)blk * p.mpw + q;
    const bool ok = q < p.mpw && n < p.n_rows;
    tok_row[jt] = ok ? n : -1;
    tok_atom[jt] = t - q * p.V;
  }

  // ---- input features u = [emb(type), x_coords, x_velocs, z_other] padded to 48 -------------
  f4 u[3][NT];
  if (p.d_emb == 32) {
    // the configured shape: two 16-byte loads of the embedding row, columns 32 + 4 g.. = [xc 3 | xv 3 | z 3 | 0...] picked
    // by lane group (same path as the split-fp16 kernel's prologue, tw_netblock_h3.hip; the general loop below walks every
    // element through runtime comparisons)
    const bool cond_shared = p.n_cond == 1, cond_per_row = p.n_cond >= p.n_rows;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int64_t n = tok_row[jt];
      const int64_t c = n < 0 || cond_shared ? 0 : (cond_per_row ? n : n % p.n_cond);
      const int64_t ca = c * p.V + (n < 0 ? 0 : tok_atom[jt]);
      const int ty = n < 0 ? 0 : p.types[ca];
      const float* er = p.emb + ty * 32 + 4 * g;
      const int64_t zi = n < 0 ? 0 : (n * p.V + tok_atom[jt]) * 3;
      const float c0 = p.xc[ca * 3], c1 = p.xc[ca * 3 + 1], c2 = p.xc[ca * 3 + 2];
      const float v0 = p.xv[ca * 3], v1 = p.xv[ca * 3 + 1], v2 = p.xv[ca * 3 + 2];
      const float z0 = p.z_other[zi], z1 = p.z_other[zi + 1], z2 = p.z_other[zi + 2];
      f4 e0, e1, m;
#pragma unroll
      for (int r = 0; r < 4; ++r) { e0[r] = er[r]; e1[r] = er[16 + r]; }
      m[0] = g == 0 ? c0 : g == 1 ? v1 : g == 2 ? z2 : 0.f;
      m[1] = g == 0 ? c1 : g == 1 ? v2 : 0.f;
      m[2] = g == 0 ? c2 : g == 1 ? z0 : 0.f;
      m[3] = g == 0 ? v0 : g == 1 ? z1 : 0.f;
      if (n < 0) e0 = e1 = m = (f4){0.f, 0.f, 0.f, 0.f};
      u[0][jt] = e0;
      u[1][jt] = e1;
      u[2][jt] = m;
    }
  } else
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int64_t n = tok_row[jt];
    const int64_t c = n < 0 ? 0 : n % p.n_cond;
    const int a = tok_atom[jt];
    const int ty = n < 0 ? 0 : p.types[c * p.V + a];
#pragma unroll
    for (int ft = 0; ft < 3; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * ft + 4 * g + r;
        float val = 0.f;
        if (n >= 0) {
          if (f < p.d_emb) val = p.emb[ty * p.d_emb + f];
          else if (f < p.d_emb + 3) val = p.xc[(c * p.V + a) * 3 + (f - p.d_emb)];
          else if (f < p.d_emb + 6) val = p.xv[(c * p.V + a) * 3 + (f - p.d_emb - 3)];
          else if (f < p.d_emb + 9) val = p.z_other[(n * p.V + a) * 3 + (f - p.d_emb - 6)];
        }
        u[ft][jt][r] = val;
      }
  }

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

  // ---- IN stage: x = in_mlp.2(silu(in_mlp.0(u))) ----------------------------------------------
  f4 x[8][NT];
  {
    const float* b2 = side + p.side_in2b + 4 * g;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
      const f4 bb = *(const f4*)(b2 + 16 * ot);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) x[ot][jt] = bb;
    }
    mlp_chain<NT, 3, 8, 24, true>(u, x, wp, ring, side + p.side_in0b + 4 * g, p.hid_chunks);
  }
  dump_x(x, 0);

  const float* sf_net = p.sfrag + (int64_t)(net * p.n_layers) * p.sf_variant_floats +
                        (p.sfrag_shared ? 0 : (int64_t)blk * p.H * NT * NT * TILE_F) + lane * 4;

  // ---- encoder layers ---------------------------------------------------------------------------
  for (int l = 0; l < p.n_layers; ++l) {
    const float* sf_base = sf_net + l * p.sf_variant_floats;  // the layer's own score fragments (chebyshev_kernel), else shared
    const float* sl = side + p.side_layers + (int64_t)l * p.side_layer_size;
    // x -> LDS (token-major) so the mixing MFMA can read it as an A operand
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) *(f4*)(xs + (16 * jt + i16) * XS + 16 * ft + 4 * g) = x[ft][jt];

    f4 y[8][NT];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) y[ot][jt] = (f4){0.f, 0.f, 0.f, 0.f};

    for (int h = 0; h < p.H; ++h) {
      // score fragments of this head: sf[jt][mt]
      f4 sf[NT][NT];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
          sf[jt][mt] = *(const f4*)(sf_base + (int64_t)((h * NT + jt) * NT + mt) * TILE_F);
      // mixing: xm[ft][jt] = (A_h X)^T tile
      f4 xm[8][NT];
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        float af[NT][4];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) af[mt][sp] = xs[(16 * mt + 4 * sp + g) * XS + 16 * ft + i16];
        f4 acc[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) acc[jt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int sp = 0; sp < 4; ++sp)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) acc[jt] = mfma4(af[mt][sp], sf[jt][mt][sp], acc[jt]);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) xm[ft][jt] = acc[jt];
      }
      // y += Wc_h . xm   (64 tiles)
#pragma unroll
      for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
          const int T = ot * 8 + ft;
          const f4 a = ring[T % RING];
          tile_mma<NT>(a, xm[ft], y[ot]);
          RING_LOAD(T);
          TW_PIN();
        }
      wp += (int64_t)64 * TILE_F;
    }
    // x = LN1(x + attn); x was not kept in registers across the head loop: reload it from LDS
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) x[ft][jt] = *(const f4*)(xs + (16 * jt + i16) * XS + 16 * ft + 4 * g);
    add_layernorm<NT>(x, y, sl + 4 * g, sl + 128 + 4 * g, p.eps);

    // FFN: y = b2 + W2 relu(W1 x + b1);  x = LN2(x + y)
    {
      const float* b2 = sl + 256 + p.d_ff + 4 * g;
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) {
        const f4 bb = *(const f4*)(b2 + 16 * ot);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = bb;
      }
      mlp_chain<NT, 8, 8, 32, false>(x, y, wp, ring, sl + 256 + 4 * g, p.ff_chunks);
    }
    add_layernorm<NT>(x, y, sl + 384 + p.d_ff + 4 * g, sl + 512 + p.d_ff + 4 * g, p.eps);
    dump_x(x, l + 1);
  }

  // ---- OUT stage: out = out_mlp.2(silu(out_mlp.0(x))) -> 3 values per token ---------------------
  f4 o[1][NT];
  {
    const f4 bb = *(const f4*)(side + p.side_out2b + 4 * g);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) o[0][jt] = bb;
    mlp_chain<NT, 8, 1, 24, true>(x, o, wp, ring, side + p.side_out0b + 4 * g, p.hid_chunks);
  }
  float* outp = p.out[net];
  if (g == 0) {
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
// host side
// ================================================================================================
// optional timing of the net-block launches (bench.py's roofline leg): HIP events on the launch stream
static bool g_profile = false;
static int g_profile_stride = 1;   // TW_PROFILE_STRIDE: bracket every n-th launch only (an event record costs ~1-2 us of stream time)
static int64_t g_profile_seen = 0; // launches since profile_begin
static bool g_profile_open = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_events;
static size_t g_events_used = 0;

int profile_begin() {
  const char* st = getenv("TW_PROFILE_STRIDE");
  g_profile_stride = st && atoi(st) > 0 ? atoi(st) : 1;
  g_profile_seen = 0;
  g_profile_open = false;
  g_profile = true;
  g_events_used = 0;
  return TW_OK;
}

static thread_local const char* g_last_netblock = "";
void note_netblock_kernel(const char* name) { g_last_netblock = name; }
const char* last_netblock_kernel() { return g_last_netblock; }

// begin=true records the start event of a new launch, begin=false the stop event of the last one
int profile_mark(hipStream_t s, bool begin) {
  if (!g_profile) return TW_OK;
  if (begin) {
    g_profile_open = (g_profile_seen++ % g_profile_stride) == 0;
    if (!g_profile_open) return TW_OK;
    if (g_events_used == g_events.size()) {
      hipEvent_t e0, e1;
      TW_HIP_CHECK(hipEventCreate(&e0));
      TW_HIP_CHECK(hipEventCreate(&e1));
      g_events.emplace_back(e0, e1);
    }
    TW_HIP_CHECK(hipEventRecord(g_events[g_events_used].first, s));
    ++g_events_used;
  } else {
    if (!g_profile_open) return TW_OK;
    TW_HIP_CHECK(hipEventRecord(g_events[g_events_used - 1].second, s));
  }
  return TW_OK;
}

int profile_end(double* total_ms, int64_t* launches) {
  g_profile = false;
  double ms = 0.0;
  for (size_t i = 0; i < g_events_used; ++i) {
    TW_HIP_CHECK(hipEventSynchronize(g_events[i].second));
    float t = 0.f;
    TW_HIP_CHECK(hipEventElapsedTime(&t, g_events[i].first, g_events[i].second));
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int64_t)g_events_used;
  g_events_used = 0;
  return TW_OK;
}

struct FusedWs {
  float *s_out, *t_out, *sfrag;
  int64_t nblk_scores, bytes;
};

static FusedWs fused_ws(const tw_flow_desc& d, int64_t n_rows, int V, int64_t n_cond, void* base) {
  FusedGeom g;
  fused_geom(V, &g);
  FusedWs w;
  char* p = (char*)base;
  auto take = [&](int64_t floats) {
    float* r = (float*)p;
    p += ((floats * 4 + 255) / 256) * 256;
    return r;
  };
  const int64_t nblocks = (n_rows + g.mpw - 1) / g.mpw;
  w.s_out = take(n_rows * V * 3);
  w.t_out = take(n_rows * V * 3);
  w.nblk_scores = nblocks;  // sized for the per-block case; the shared case uses 1
  // chebyshev_kernel: one fragment set per (net, layer) of the coupling layer in flight
  w.sfrag = take((d.cheb_order > 0 ? 2 * d.n_layers : 1) * nblocks * d.n_heads * g.nt * g.nt * TILE_F);
  (void)n_cond;
  w.bytes = p - (char*)base;
  return w;
}

int64_t fused_workspace_bytes(const tw_flow_desc& d, int64_t n_rows, int n_atoms) {
  if (d.variant == 1) return dense_fused_workspace_bytes(d, n_rows, n_atoms);
  return fused_ws(d, n_rows, n_atoms, n_rows, nullptr).bytes;
}

// Score fragments of coupling layer c.  With the Gaussian basis they do not depend on c (one call per flow pass);
// chebyshev_kernel needs them per coupling layer: 2 * n_layers variants, `*variant_floats` apart (0 if one variant).
static int launch_score_frags(const FlowArgs& a, const RawLayout& L, const FusedGeom& g, float* sfrag, bool shared, int c,
                              int64_t* variant_floats) {
  const tw_flow_desc& d = *a.desc;
  const int V = a.n_atoms;
  const int64_t nblocks = shared ? 1 : (a.n_rows + g.mpw - 1) / g.mpw;
  const ScoreBasis basis = score_basis(d, L, a.raw, c);
  const int64_t vf = basis.n_variants > 1 ? nblocks * d.n_heads * g.nt * g.nt * TILE_F : 0;
  size_t shm = (size_t)(g.mpw * V * 3 + g.mpw * V * V + g.mpw * d.n_heads * V) * 4 + (size_t)g.mpw * V;
  hipLaunchKernelGGL(score_frag_kernel, dim3((unsigned)nblocks, (unsigned)basis.n_variants), dim3(256), shm, a.stream, a.x_coords,
                     a.masked, a.raw + L.lengthscales + (a.reverse ? d.n_heads : 0), d.n_heads, V, g.mpw, g.nt, a.n_rows, a.n_cond,
                     d.normalise, V > 25, sfrag, basis, vf);
  TW_LAUNCH_CHECK();
  *variant_floats = vf;
  return TW_OK;
}

static int launch_netblock(const FlowArgs& a, const RawLayout& L, const FusedGeom& g, int c, int net_sel,
                           const float* z_other, const float* sfrag, int64_t sf_variant_floats, bool shared, float* s_out,
                           float* t_out, float* dump) {
  const tw_flow_desc& d = *a.desc;
  const StreamGeom sg = stream_geom(d);
  const PackedLayout P = packed_layout(d);
  NBParams p;
  p.packed = a.packed + (int64_t)(c * 2) * P.net_stride;
  p.net_stride = P.net_stride;
  p.tiles_per_net = sg.tiles;
  p.side_in0b = sg.side_in0b;
  p.side_in2b = sg.side_in2b;
  p.side_layers = sg.side_layers;
  p.side_layer_size = sg.side_layer_size;
  p.side_out0b = sg.side_out0b;
  p.side_out2b = sg.side_out2b;
  p.emb = a.raw + L.emb;
  p.types = a.atom_types;
  p.xc = a.x_coords;
  p.xv = a.x_velocs;
  p.z_other = z_other;
  p.sfrag = sfrag;
  p.sfrag_shared = shared ? 1 : 0;
  p.sf_variant_floats = sf_variant_floats;
  p.out[0] = s_out;
  p.out[1] = t_out;
  p.dump = dump;
  p.n_rows = a.n_rows;
  p.n_cond = a.n_cond;
  p.V = a.n_atoms;
  p.mpw = g.mpw;
  p.nblocks = (int)((a.n_rows + g.mpw - 1) / g.mpw);
  p.tile_mask = g.tile_mask;
  p.H = d.n_heads;
  p.n_layers = d.n_layers;
  p.ff_chunks = sg.ff_chunks;
  p.hid_chunks = sg.hid_chunks;
  p.d_emb = d.d_emb;
  p.d_ff = d.d_ff;
  p.eps = d.ln_eps;
  p.net_sel = net_sel;
  const int wgs_per_net = (p.nblocks + 3) / 4;
  unsigned grid = net_sel < 0 ? 8u * (unsigned)((wgs_per_net + 3) / 4) : (unsigned)wgs_per_net;
  const size_t shm = (size_t)4 * 16 * g.nt * XS * sizeof(float);
  int prc;
  if ((prc = profile_mark(a.stream, true))) return prc;
  static LdsLimit lim3, lim4;
  if (g.nt == 3) {
    if ((prc = lim3.ensure((const void*)netblock_kernel<3>, (int)shm))) return prc;
    note_netblock_kernel("tw::netblock_kernel<3>");
    hipLaunchKernelGGL(netblock_kernel<3>, dim3(grid), dim3(256), shm, a.stream, p);
  } else {
    if ((prc = lim4.ensure((const void*)netblock_kernel<4>, (int)shm))) return prc;
    note_netblock_kernel("tw::netblock_kernel<4>");
    hipLaunchKernelGGL(netblock_kernel<4>, dim3(grid), dim3(256), shm, a.stream, p);
  }
  TW_LAUNCH_CHECK();
  if ((prc = profile_mark(a.stream, false))) return prc;
  return TW_OK;
}

int flow_pass_fused(const FlowArgs& a) {
  const tw_flow_desc& d = *a.desc;
  if (d.variant == 1) return flow_pass_fused_dense(a);
  FusedGeom g;
  TW_REQUIRE(fused_geom(a.n_atoms, &g), "fused path: unsupported atom count %d", a.n_atoms);
  const RawLayout L = raw_layout(d);
  const FusedWs w = fused_ws(d, a.n_rows, a.n_atoms, a.n_cond, a.ws);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  const bool shared = a.n_cond == 1;  // every conformation is conditioned on the same x: one fragment set
  int rc;
  int64_t vf = 0;
  if (d.cheb_order == 0 && (rc = launch_score_frags(a, L, g, w.sfrag, shared, 0, &vf))) return rc;
  for (int i = 0; i < d.n_coupling; ++i) {
    const int c = a.reverse ? d.n_coupling - 1 - i : i;
    const bool positions = (c % 2) == d.pos_mod2;
    const float* z_other = positions ? a.z_velocs : a.z_coords;
    float* z_t = positions ? a.z_coords : a.z_velocs;
    if (d.cheb_order > 0 && (rc = launch_score_frags(a, L, g, w.sfrag, shared, c, &vf))) return rc;
    if ((rc = launch_netblock(a, L, g, c, -1, z_other, w.sfrag, vf, shared, w.s_out, w.t_out, nullptr))) return rc;
    if ((rc = launch_coupling(w.s_out, w.t_out, a.masked, a.n_cond, z_t, a.delta_logp, a.n_rows, a.n_atoms, a.reverse,
                              a.stream, nullptr, a.desc->range_flag)))
      return rc;
  }
  return TW_OK;
}

int debug_netblock_fused(const FlowArgs& a, int c, int net, const float* z_other, float* dump) {
  const tw_flow_desc& d = *a.desc;
  if (d.variant == 1) return debug_netblock_fused_dense(a, c, net, z_other, dump);
  FusedGeom g;
  TW_REQUIRE(fused_geom(a.n_atoms, &g), "fused path: unsupported atom count %d", a.n_atoms);
  const RawLayout L = raw_layout(d);
  const FusedWs w = fused_ws(d, a.n_rows, a.n_atoms, a.n_cond, a.ws);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  const bool shared = a.n_cond == 1;
  int rc;
  int64_t vf = 0;
  if ((rc = launch_score_frags(a, L, g, w.sfrag, shared, c, &vf))) return rc;
  return launch_netblock(a, L, g, c, net, z_other, w.sfrag, vf, shared, w.s_out, w.t_out, dump);
}

}  // namespace tw
