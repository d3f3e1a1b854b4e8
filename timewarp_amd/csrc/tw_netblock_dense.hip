// Fused f32-MFMA net-block kernel for the dense softmax-attention flow (`transformer_nvp`, gfx950 / CDNA4).
//
// One launch evaluates BOTH coupling nets of one coupling layer of TransformerCouplingLayer
// (modules/transformer_nvp.py:13-97): in_mlp -> L x nn.TransformerEncoderLayer(d_model 128, 8 heads, relu,
// post-norm, batch_first, src_key_padding_mask; modules/layers/transformer_block.py:18-72) -> out_mlp.
// Same skeleton as tw_netblock.hip (transposed formulation on v_mfma_f32_16x16x4_f32, activations chained in
// registers in MFMA D/B layout, one wave = whole molecules, weights streamed as 1 KiB A-fragment tiles through an
// 8-deep register ring, FFN chained 32 hidden units at a time); what differs is the attention block:
//
//   per head h (dh = 16 = one MFMA tile):
//     q_h, k_h, v_h = three 16-feature output tiles of in_proj (24 weight tiles, bias as the accumulator's initial value,
//                     q scaled by 1/sqrt(dh))                                                      -> 36 MFMA groups
//     the three tiles go to a wave-private LDS block [token][16]; every lane then does the softmax attention of ITS
//     tokens (the D/B layout gives lane (g, i16) token 16 jt + i16 and features 4g..4g+3): scores against the keys of
//     the token's own molecule (the four lanes of a token split the keys and park the scores in LDS), then
//     exp / sum / P.V for its four features; padded keys are -inf as in nn.MultiheadAttention.  fp32 on the VALU.
//     The result IS the B operand tile `ft = h` of out_proj:  y[ot] += W_out(ot, h) . o_h              -> 8 tiles
//   y += out_proj.bias;  x = LN1(x + y);  FFN as in the kernel variant;  x = LN2(x + y)
//
// Weight stream per net (tiles in consumption order): IN hid_chunks x 24 | per layer: 8 heads x (24 + 8) then
// ff_chunks x 32 | OUT hid_chunks x 24;  side floats: in0_b in2_b { in_proj_b[384] out_proj_b[128] n1w n1b b1[ff] b2
// n2w n2b } out0_b out2_b.  Supported: d_model 128, 8 heads, molecules of <= 64 atoms, input width <= 48 (no position
// features, configs/transformer_nvp.yaml) or 32 + 9 + 128 random Fourier features (configs/transformer_nvp_posenc.yaml: cos / sin
// of the conditioning positions computed in the prologue, eleven input tiles); other dense configurations use the per-op path.
#include "tw_common.h"
#include "tw_nb_f32.h"

namespace tw {

#define DH 16       // head dimension = one 16-wide MFMA tile
#define QS 20       // LDS row stride (floats) of the q / k / v tiles: conflict-free 16-byte writes and reads
#define PS 65       // LDS row stride (floats) of the score tile [token][key <= 64]

// input width of the in-MLP in 16-feature tiles: 3 (atom embedding + 9, no position features: transformer_nvp.yaml) or 11
// (32 + 9 + 128 random Fourier features of the conditioning positions: transformer_nvp_posenc.yaml); 0 = not supported
static int dense_in_tiles(const tw_flow_desc& d) {
  if (d.d_rff == 0 && d.d_emb + 9 <= 48) return 3;
  if (d.d_rff == 128 && d.d_emb == 32) return 11;
  return 0;
}
// weight tiles of one in-MLP chunk: W0 chunk 2 x FT_IN, W2 chunk 8 x 2, padded to a multiple of the ring depth
static int dense_in_body(int ft_in) { return (2 * ft_in + 16 + RING - 1) / RING * RING; }

bool dense_fused_supported(const tw_flow_desc& d, int n_atoms) {
  FusedGeom g;
  return d.variant == 1 && d.d_model == 128 && d.n_heads == 8 && d.d_hidden % 32 == 0 && d.d_ff % 32 == 0 &&
         d.d_emb % 4 == 0 && dense_in_tiles(d) > 0 && fused_geom(n_atoms, &g);
}

struct DenseGeom {
  int hid_chunks, ff_chunks, L;
  int64_t in_tiles, layer_tiles, out_tiles, tiles;
  int64_t side_in0b, side_in2b, side_layers, side_layer_size, side_out0b, side_out2b, side_size;
  // offsets inside one layer's side block
  int64_t l_inb, l_outb, l_n1w, l_n1b, l_b1, l_b2, l_n2w, l_n2b;
};

static DenseGeom dense_geom(const tw_flow_desc& d) {
  DenseGeom s;
  s.hid_chunks = d.d_hidden / 32;
  s.ff_chunks = d.d_ff / 32;
  s.L = d.n_layers;
  s.in_tiles = (int64_t)s.hid_chunks * dense_in_body(dense_in_tiles(d));
  s.layer_tiles = (int64_t)d.n_heads * 32 + (int64_t)s.ff_chunks * 32;
  s.out_tiles = (int64_t)s.hid_chunks * 24;
  s.tiles = s.in_tiles + s.L * s.layer_tiles + s.out_tiles;
  int64_t q = 0;
  s.l_inb = q; q += 384;
  s.l_outb = q; q += 128;
  s.l_n1w = q; q += 128;
  s.l_n1b = q; q += 128;
  s.l_b1 = q; q += d.d_ff;
  s.l_b2 = q; q += 128;
  s.l_n2w = q; q += 128;
  s.l_n2b = q; q += 128;
  s.side_layer_size = q;
  int64_t o = 0;
  s.side_in0b = o; o += d.d_hidden;
  s.side_in2b = o; o += 128;
  s.side_layers = o;
  o += s.L * s.side_layer_size;
  s.side_out0b = o; o += d.d_hidden;
  s.side_out2b = o; o += 16;
  s.side_size = (o + 63) / 64 * 64 + 64;  // slack: the bias prefetch of mlp_chain reads one chunk ahead
  return s;
}

PackedLayout dense_packed_layout(const tw_flow_desc& d) {
  DenseGeom s = dense_geom(d);
  PackedLayout p;
  p.tiles_per_net = s.tiles;
  p.side_per_net = s.side_size;
  p.net_stride = (s.tiles * TILE_F + s.side_size + 255) / 256 * 256;
  p.total = p.net_stride * 2 * d.n_coupling + (int64_t)(RING + 1) * TILE_F;  // ring prefetch overrun
  return p;
}

int dense_pack_weights(const tw_flow_desc& d, const float* raw, float* packed, hipStream_t s) {
  const RawLayout L = raw_layout(d);
  const DenseGeom g = dense_geom(d);
  const PackedLayout P = dense_packed_layout(d);
  TW_HIP_CHECK(hipMemsetAsync(packed, 0, P.total * sizeof(float), s));
  int rc;
  for (int c = 0; c < d.n_coupling; ++c)
    for (int net = 0; net < 2; ++net) {
      const float* nb = raw + net_base(L, c, net);
      float* pn = packed + (int64_t)(c * 2 + net) * P.net_stride;
      float* t = pn;  // tile cursor
      const int ft_in = dense_in_tiles(d), in_body = dense_in_body(ft_in);
      for (int ch = 0; ch < g.hid_chunks; ++ch) {
        if ((rc = pack_block(nb + L.net.in0_w, L.d_in, d.d_hidden, L.d_in, 32 * ch, 0, 2, ft_in, t, s))) return rc;
        if ((rc = pack_block(nb + L.net.in2_w, d.d_hidden, 128, d.d_hidden, 0, 32 * ch, 8, 2, t + 2 * ft_in * TILE_F, s))) return rc;
        t += in_body * TILE_F;
      }
      for (int l = 0; l < d.n_layers; ++l) {
        const float* lb = nb + L.net.layers + (int64_t)l * L.layer.size;
        for (int h = 0; h < d.n_heads; ++h) {
          // q_h, k_h, v_h: rows part * 128 + 16 h .. + 15 of in_proj_weight [384, 128], all 8 k-tiles each
          for (int part = 0; part < 3; ++part)
            if ((rc = pack_block(lb + L.layer.in_w, 128, 384, 128, part * 128 + DH * h, 0, 1, 8, t + part * 8 * TILE_F, s))) return rc;
          // out_proj columns 16 h .. 16 h + 15 for all 8 output tiles
          if ((rc = pack_block(lb + L.layer.out_w, 128, 128, 128, 0, DH * h, 8, 1, t + 24 * TILE_F, s))) return rc;
          t += 32 * TILE_F;
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
      float* sd = pn + g.tiles * TILE_F;
      if ((rc = copy_pad(nb + L.net.in0_b, d.d_hidden, sd + g.side_in0b, d.d_hidden, s))) return rc;
      if ((rc = copy_pad(nb + L.net.in2_b, 128, sd + g.side_in2b, 128, s))) return rc;
      for (int l = 0; l < d.n_layers; ++l) {
        const float* lb = nb + L.net.layers + (int64_t)l * L.layer.size;
        float* sl = sd + g.side_layers + (int64_t)l * g.side_layer_size;
        if ((rc = copy_pad(lb + L.layer.in_b, 384, sl + g.l_inb, 384, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.out_b, 128, sl + g.l_outb, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.n1w, 128, sl + g.l_n1w, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.n1b, 128, sl + g.l_n1b, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.b1, d.d_ff, sl + g.l_b1, d.d_ff, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.b2, 128, sl + g.l_b2, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.n2w, 128, sl + g.l_n2w, 128, s))) return rc;
        if ((rc = copy_pad(lb + L.layer.n2b, 128, sl + g.l_n2b, 128, s))) return rc;
      }
      if ((rc = copy_pad(nb + L.net.out0_b, d.d_hidden, sd + g.side_out0b, d.d_hidden, s))) return rc;
      if ((rc = copy_pad(nb + L.net.out2_b, 3, sd + g.side_out2b, 16, s))) return rc;
    }
  return TW_OK;
}

// ================================================================================================
// the kernel
// ================================================================================================
struct DNParams {
  const float* packed;
  int64_t net_stride, tiles_per_net;
  int64_t side_in0b, side_in2b, side_layers, side_layer_size, side_out0b, side_out2b;
  int64_t l_inb, l_outb, l_n1w, l_n1b, l_b1, l_b2, l_n2w, l_n2b;
  const float* emb;
  const int32_t* types;
  const float* xc;
  const float* xv;
  const float* z_other;
  const float* rff;  // [3, d_rff / 2] Gaussian vectors of this coupling layer's position encoder (d_rff > 0)
  int d_rff;
  const uint8_t* masked;
  float* out[2];
  float* dump;
  int64_t n_rows, n_cond;
  int V, mpw, nblocks;
  int H, n_layers, ff_chunks, hid_chunks, d_emb;
  float eps;
  int net_sel;
  int debug;  // timing experiments (tw_debug_set_flags): 64 = no softmax section (o_h = v_h), 128 = also no LDS round trip
};

template <int NT, int FT_IN = 3>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
netblock_dense_kernel(const DNParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i16 = lane & 15;

  int net, wg;
  if (p.net_sel < 0) {  // XCD-aware mapping: XCDs 0-3 stream the scale net, 4-7 the shift net (speed only)
    const int xcd = blockIdx.x & 7;
    net = xcd >> 2;
    wg = (blockIdx.x >> 3) * 4 + (xcd & 3);
  } else {
    net = p.net_sel;
    wg = blockIdx.x;
  }
  const int blk = wg * 4 + wave;
  if (blk >= p.nblocks) return;

  // wave-private q / k / v tiles of the head in flight: [16 NT tokens][QS]
  float* qs = lds + wave * (16 * NT * (3 * QS + PS));
  float* ks = qs + 16 * NT * QS;
  float* vs = ks + 16 * NT * QS;
  float* ps = vs + 16 * NT * QS;
  const float* net_base = p.packed + (int64_t)net * p.net_stride;
  const float* side = net_base + p.tiles_per_net * TILE_F;
  const float* wp = net_base + lane * 4;
  f4 ring[RING];
#pragma unroll
  for (int i = 0; i < RING; ++i) ring[i] = *(const f4*)(wp + (int64_t)i * TILE_F);

  // ---- token bookkeeping: row, atom, first token of the token's molecule, bit mask of its usable keys ----
  int64_t tok_row[NT];
  int tok_atom[NT], tok_mol0[NT];
  unsigned long long keymask[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int t = 16 * jt + i16;
    const int q = t / p.V;
    const int64_t n = (int64_t)blk * p.mpw + q;
    const bool ok = q < p.mpw && n < p.n_rows;
    tok_row[jt] = ok ? n : -1;
    tok_atom[jt] = t - q * p.V;
    tok_mol0[jt] = q * p.V;
    unsigned long long m = 0ull;
    if (ok) {
      const uint8_t* mk = p.masked + (n % p.n_cond) * p.V;
      for (int a = 0; a < p.V; ++a) m |= mk[a] ? 0ull : (1ull << a);
    }
    keymask[jt] = m;
  }

  // ---- input features u = [emb(type), x_coords, x_velocs, z_other (, rff(x_coords))] padded to 16 FT_IN -------------
  f4 u[FT_IN][NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int64_t n = tok_row[jt];
    const int64_t c = n < 0 ? 0 : n % p.n_cond;
    const int a = tok_atom[jt];
    const int ty = n < 0 ? 0 : p.types[c * p.V + a];
    const float* px = p.xc + (c * p.V + a) * 3;
#pragma unroll
    for (int ft = 0; ft < FT_IN; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * ft + 4 * g + r;
        float val = 0.f;
        if (n >= 0) {
          if (f < p.d_emb) val = p.emb[ty * p.d_emb + f];
          else if (f < p.d_emb + 3) val = px[f - p.d_emb];
          else if (f < p.d_emb + 6) val = p.xv[(c * p.V + a) * 3 + (f - p.d_emb - 3)];
          else if (f < p.d_emb + 9) val = p.z_other[(n * p.V + a) * 3 + (f - p.d_emb - 6)];
          else if (FT_IN > 3 && f < p.d_emb + 9 + p.d_rff) {
            // rff_position_encoder.py:57-62: sqrt(1/n) * [cos(x G), sin(x G)]; same arithmetic as build_input_kernel
            const int nvec = p.d_rff / 2;
            const int j = f - p.d_emb - 9;
            const int col = j % nvec;
            const float ip = px[0] * p.rff[0 * nvec + col] + px[1] * p.rff[1 * nvec + col] + px[2] * p.rff[2 * nvec + col];
            val = sqrtf(1.0f / nvec) * (j < nvec ? cosf(ip) : sinf(ip));
          }
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

  // ---- IN stage ----
  f4 x[8][NT];
  {
    const float* b2 = side + p.side_in2b + 4 * g;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
      const f4 bb = *(const f4*)(b2 + 16 * ot);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) x[ot][jt] = bb;
    }
    mlp_chain<NT, FT_IN, 8, (2 * FT_IN + 16 + RING - 1) / RING * RING, true>(u, x, wp, ring, side + p.side_in0b + 4 * g, p.hid_chunks);
  }
  dump_x(x, 0);

  const float inv_sqrt_dh = 0.25f;  // 1 / sqrt(16)
  for (int l = 0; l < p.n_layers; ++l) {
    const float* sl = side + p.side_layers + (int64_t)l * p.side_layer_size;
    f4 y[8][NT];
    {
      const float* bo = sl + p.l_outb + 4 * g;
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) {
        const f4 bb = *(const f4*)(bo + 16 * ot);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = bb;
      }
    }
    for (int h = 0; h < p.H; ++h) {
      // q_h, k_h, v_h = in_proj tiles (part, h); bias = initial accumulator
      f4 qkv[3][NT];
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        const f4 bb = *(const f4*)(sl + p.l_inb + part * 128 + DH * h + 4 * g);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) qkv[part][jt] = bb;
      }
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
          const int T = part * 8 + ft;
          const f4 a = ring[T % RING];
          tile_mma<NT>(a, x[ft], qkv[part]);
          RING_LOAD(T);
          TW_PIN();
        }
      // to the wave-private LDS tiles, [token][feature]
      if (!TW_EXPERIMENT(p.debug & 128))
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int row = (16 * jt + i16) * QS + 4 * g;
        *(f4*)(qs + row) = qkv[0][jt] * inv_sqrt_dh;
        *(f4*)(ks + row) = qkv[1][jt];
        *(f4*)(vs + row) = qkv[2][jt];
      }
      // Softmax attention of this lane's tokens over the keys of their molecule.  The four lanes that share a token
      // (g = 0..3) split the score pass - lane g takes keys g, g + 4, ... - and leave the scores in the wave-private
      // tile ps[token][key]; the second pass (all keys, this lane's four output features) reads them back.  One wave
      // wrote what it reads: LDS operations of a wave complete in order.  Padded keys get -inf (nn.MultiheadAttention).
      f4 oh[NT];
      float mx[NT];
      if (TW_EXPERIMENT(p.debug & 64)) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) oh[jt] = qkv[2][jt] + qkv[0][jt] * qkv[1][jt];
      } else {
      // pass 1: scores of the keys this lane group owns; the NT tokens of the lane are independent chains
      {
        f4 q4[NT][4];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          mx[jt] = -INFINITY;
#pragma unroll
          for (int i = 0; i < 4; ++i) q4[jt][i] = *(const f4*)(qs + (16 * jt + i16) * QS + 4 * i);
        }
        for (int m = g; m < p.V; m += 4) {
          f4 k4[NT][4];
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int i = 0; i < 4; ++i) k4[jt][i] = *(const f4*)(ks + (tok_mol0[jt] + m) * QS + 4 * i);
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            float a0 = q4[jt][0][0] * k4[jt][0][0], a1 = q4[jt][1][0] * k4[jt][1][0];
            float a2 = q4[jt][2][0] * k4[jt][2][0], a3 = q4[jt][3][0] * k4[jt][3][0];
#pragma unroll
            for (int r = 1; r < 4; ++r) {
              a0 = fmaf(q4[jt][0][r], k4[jt][0][r], a0);
              a1 = fmaf(q4[jt][1][r], k4[jt][1][r], a1);
              a2 = fmaf(q4[jt][2][r], k4[jt][2][r], a2);
              a3 = fmaf(q4[jt][3][r], k4[jt][3][r], a3);
            }
            const float sc = ((keymask[jt] >> m) & 1ull) ? (a0 + a1) + (a2 + a3) : -INFINITY;
            ps[(16 * jt + i16) * PS + m] = sc;
            mx[jt] = fmaxf(mx[jt], sc);
          }
        }
      }
      // pass 2: e^(s - max) (hardware exp2, 1 ulp; 0 for padded keys), sum, P.V for this lane's four features; keys in
      // blocks of four so the LDS reads of a block are in flight together
      {
        float sum[NT], mxl[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          float m_ = fmaxf(mx[jt], __shfl_xor(mx[jt], 16));
          m_ = fmaxf(m_, __shfl_xor(m_, 32));
          mxl[jt] = m_ * 1.44269504088896340736f;
          sum[jt] = 0.f;
          oh[jt] = (f4){0.f, 0.f, 0.f, 0.f};
        }
        for (int m0 = 0; m0 < p.V; m0 += 4) {
          float sc[NT][4];
          f4 vv[NT][4];
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int m = min(m0 + j, p.V - 1);
              sc[jt][j] = ps[(16 * jt + i16) * PS + m];
              vv[jt][j] = *(const f4*)(vs + (tok_mol0[jt] + m) * QS + 4 * g);
            }
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float e = __builtin_amdgcn_exp2f(fmaf(sc[jt][j], 1.44269504088896340736f, -mxl[jt]));
              e = (m0 + j < p.V) ? e : 0.f;
              sum[jt] += e;
              oh[jt] = oh[jt] + vv[jt][j] * e;
            }
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)  // tokens outside every molecule (tile padding) have no keys: keep them finite
          oh[jt] = keymask[jt] ? oh[jt] * (1.0f / sum[jt]) : (f4){0.f, 0.f, 0.f, 0.f};
      }
      }
      // y += W_out(:, head h) . o_h   (8 tiles)
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) {
        const int T = 24 + ot;
        const f4 a = ring[T % RING];
        tile_mma<NT>(a, oh, y[ot]);
        RING_LOAD(T);
        TW_PIN();
      }
      wp += (int64_t)32 * TILE_F;
    }
    add_layernorm<NT>(x, y, sl + p.l_n1w + 4 * g, sl + p.l_n1b + 4 * g, p.eps);

    {
      const float* b2 = sl + p.l_b2 + 4 * g;
#pragma unroll
      for (int ot = 0; ot < 8; ++ot) {
        const f4 bb = *(const f4*)(b2 + 16 * ot);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) y[ot][jt] = bb;
      }
      mlp_chain<NT, 8, 8, 32, false>(x, y, wp, ring, sl + p.l_b1 + 4 * g, p.ff_chunks);
    }
    add_layernorm<NT>(x, y, sl + p.l_n2w + 4 * g, sl + p.l_n2b + 4 * g, p.eps);
    dump_x(x, l + 1);
  }

  // ---- OUT stage ----
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
struct DenseWs {
  float *s_out, *t_out;
  int64_t bytes;
};

static DenseWs dense_ws(int64_t n_rows, int V, void* base) {
  DenseWs w;
  char* p = (char*)base;
  auto take = [&](int64_t floats) {
    float* r = (float*)p;
    p += ((floats * 4 + 255) / 256) * 256;
    return r;
  };
  w.s_out = take(n_rows * V * 3);
  w.t_out = take(n_rows * V * 3);
  w.bytes = p - (char*)base;
  return w;
}

int64_t dense_fused_workspace_bytes(const tw_flow_desc&, int64_t n_rows, int n_atoms) {
  return dense_ws(n_rows, n_atoms, nullptr).bytes;
}

static int dense_launch(const FlowArgs& a, const RawLayout& L, const FusedGeom& g, int c, int net_sel, const float* z_other,
                        float* s_out, float* t_out, float* dump) {
  const tw_flow_desc& d = *a.desc;
  const DenseGeom sg = dense_geom(d);
  const PackedLayout P = dense_packed_layout(d);
  DNParams p;
  p.packed = a.packed + (int64_t)(c * 2) * P.net_stride;
  p.net_stride = P.net_stride;
  p.tiles_per_net = sg.tiles;
  p.side_in0b = sg.side_in0b;
  p.side_in2b = sg.side_in2b;
  p.side_layers = sg.side_layers;
  p.side_layer_size = sg.side_layer_size;
  p.side_out0b = sg.side_out0b;
  p.side_out2b = sg.side_out2b;
  p.l_inb = sg.l_inb; p.l_outb = sg.l_outb; p.l_n1w = sg.l_n1w; p.l_n1b = sg.l_n1b;
  p.l_b1 = sg.l_b1; p.l_b2 = sg.l_b2; p.l_n2w = sg.l_n2w; p.l_n2b = sg.l_n2b;
  p.emb = a.raw + L.emb;
  p.types = a.atom_types;
  p.xc = a.x_coords;
  p.xv = a.x_velocs;
  p.z_other = z_other;
  p.rff = d.d_rff > 0 ? a.raw + L.chain + (int64_t)c * L.coupling_size + L.rff : nullptr;
  p.d_rff = d.d_rff;
  p.masked = a.masked;
  p.out[0] = s_out;
  p.out[1] = t_out;
  p.dump = dump;
  p.n_rows = a.n_rows;
  p.n_cond = a.n_cond;
  p.V = a.n_atoms;
  p.mpw = g.mpw;
  p.nblocks = (int)((a.n_rows + g.mpw - 1) / g.mpw);
  p.H = d.n_heads;
  p.n_layers = d.n_layers;
  p.ff_chunks = sg.ff_chunks;
  p.hid_chunks = sg.hid_chunks;
  p.d_emb = d.d_emb;
  p.eps = d.ln_eps;
  p.net_sel = net_sel;
  p.debug = g_debug_flags;
  const int wgs_per_net = (p.nblocks + 3) / 4;
  unsigned grid = net_sel < 0 ? 8u * (unsigned)((wgs_per_net + 3) / 4) : (unsigned)wgs_per_net;
  const size_t shm = (size_t)4 * 16 * g.nt * (3 * QS + PS) * sizeof(float);
  int prc;
  if ((prc = profile_mark(a.stream, true))) return prc;
  static LdsLimit lim3, lim4, lim3r, lim4r;  // > 64 KiB of dynamic LDS needs the kernel's limit raised, once per device
  const bool rff = dense_in_tiles(d) == 11;
  if (g.nt == 3 && !rff) {
    if ((prc = lim3.ensure((const void*)netblock_dense_kernel<3>, (int)shm))) return prc;
    note_netblock_kernel("tw::netblock_dense_kernel<3, 3>");
    hipLaunchKernelGGL(netblock_dense_kernel<3>, dim3(grid), dim3(256), shm, a.stream, p);
  } else if (!rff) {
    if ((prc = lim4.ensure((const void*)netblock_dense_kernel<4>, (int)shm))) return prc;
    note_netblock_kernel("tw::netblock_dense_kernel<4, 3>");
    hipLaunchKernelGGL(netblock_dense_kernel<4>, dim3(grid), dim3(256), shm, a.stream, p);
  } else if (g.nt == 3) {
    if ((prc = lim3r.ensure((const void*)netblock_dense_kernel<3, 11>, (int)shm))) return prc;
    note_netblock_kernel("tw::netblock_dense_kernel<3, 11>");
    hipLaunchKernelGGL((netblock_dense_kernel<3, 11>), dim3(grid), dim3(256), shm, a.stream, p);
  } else {
    if ((prc = lim4r.ensure((const void*)netblock_dense_kernel<4, 11>, (int)shm))) return prc;
    note_netblock_kernel("tw::netblock_dense_kernel<4, 11>");
    hipLaunchKernelGGL((netblock_dense_kernel<4, 11>), dim3(grid), dim3(256), shm, a.stream, p);
  }
  TW_LAUNCH_CHECK();
  if ((prc = profile_mark(a.stream, false))) return prc;
  return TW_OK;
}

int flow_pass_fused_dense(const FlowArgs& a) {
  const tw_flow_desc& d = *a.desc;
  FusedGeom g;
  TW_REQUIRE(fused_geom(a.n_atoms, &g), "fused dense path: unsupported atom count %d", a.n_atoms);
  const RawLayout L = raw_layout(d);
  const DenseWs w = dense_ws(a.n_rows, a.n_atoms, a.ws);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  int rc;
  for (int i = 0; i < d.n_coupling; ++i) {
    const int c = a.reverse ? d.n_coupling - 1 - i : i;
    const bool positions = (c % 2) == d.pos_mod2;
    const float* z_other = positions ? a.z_velocs : a.z_coords;
    float* z_t = positions ? a.z_coords : a.z_velocs;
    if ((rc = dense_launch(a, L, g, c, -1, z_other, w.s_out, w.t_out, nullptr))) return rc;
    if ((rc = launch_coupling(w.s_out, w.t_out, a.masked, a.n_cond, z_t, a.delta_logp, a.n_rows, a.n_atoms, a.reverse,
                              a.stream, nullptr, a.desc->range_flag)))
      return rc;
  }
  return TW_OK;
}

int debug_netblock_fused_dense(const FlowArgs& a, int c, int net, const float* z_other, float* dump) {
  const tw_flow_desc& d = *a.desc;
  FusedGeom g;
  TW_REQUIRE(fused_geom(a.n_atoms, &g), "fused dense path: unsupported atom count %d", a.n_atoms);
  const RawLayout L = raw_layout(d);
  const DenseWs w = dense_ws(a.n_rows, a.n_atoms, a.ws);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  return dense_launch(a, L, g, c, net, z_other, w.s_out, w.t_out, dump);
}

}  // namespace tw
