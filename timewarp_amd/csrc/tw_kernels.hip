// Glue kernels (scores, centring, coupling, prior log-prob, kinetic energy, MH accept, chirality)
// and the SIMPLE flow path: one plain HIP kernel per reference torch op.  The simple path is the
// always-available HIP implementation (all variants; kernel attention on molecules of any size since r05 - above 64 / ~200 atoms the
// mixing / the scores take tiled kernels instead of one V x V tile in the LDS; the dense softmax variant up to ~190 atoms,
// larger ones are refused with a message, TW_LDS_LIMIT); the fused f32-MFMA path
// (tw_netblock.hip) is the fast one for the kernel variant.
#include <stdarg.h>

#include "tw_common.h"

namespace tw {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

RawLayout raw_layout(const tw_flow_desc& d) {
  RawLayout L;
  const int64_t dm = d.d_model, ff = d.d_ff, hid = d.d_hidden, H = d.n_heads;
  L.d_in = d.d_emb + 9 + (d.variant == 1 ? d.d_rff : 0);
  int64_t o = 0;
  L.emb = o; o += (int64_t)d.n_elements * d.d_emb;
  L.lengthscales = o; o += (d.variant == 0 ? 2 * H : 0);  // [0]: used by the forward pass, [1]: by the reverse pass
  L.prior = o; o += 2;
  L.chain = o;
  // layer
  LayerOff& y = L.layer;
  int64_t q = 0;
  y.wv = y.wo = y.in_w = y.in_b = y.out_w = y.out_b = -1;
  y.cheb = -1;
  if (d.variant == 0) {
    y.wv = q; q += H * dm * dm;
    y.wo = q; q += dm * H * dm;
    if (d.cheb_order > 0) { y.cheb = q; q += H * d.cheb_order; }
  } else {
    y.in_w = q; q += 3 * dm * dm;
    y.in_b = q; q += 3 * dm;
    y.out_w = q; q += dm * dm;
    y.out_b = q; q += dm;
  }
  y.w1 = q; q += ff * dm;
  y.b1 = q; q += ff;
  y.w2 = q; q += dm * ff;
  y.b2 = q; q += dm;
  y.n1w = q; q += dm;
  y.n1b = q; q += dm;
  y.n2w = q; q += dm;
  y.n2b = q; q += dm;
  y.size = q;
  // net
  NetOff& n = L.net;
  q = 0;
  n.in0_w = q; q += hid * L.d_in;
  n.in0_b = q; q += hid;
  n.in2_w = q; q += dm * hid;
  n.in2_b = q; q += dm;
  n.layers = q; q += (int64_t)d.n_layers * y.size;
  n.out0_w = q; q += hid * dm;
  n.out0_b = q; q += hid;
  n.out2_w = q; q += 3 * hid;
  n.out2_b = q; q += 3;
  n.size = q;
  // coupling
  q = 0;
  L.rff = q; q += (d.variant == 1 ? 3 * (d.d_rff / 2) : 0);
  L.nets = q; q += 2 * n.size;
  L.coupling_size = q;
  L.total = L.chain + (int64_t)d.n_coupling * L.coupling_size;
  return L;
}

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Fragment-order layout of the split fp16 MFMA operands of the folded attention (S, x^T, Wc): a [rows (padded to 16), keys or features
// (padded to 32)] matrix is stored as [k-tile of 32][row tile of 16][lane 16 g + i16][8 halves] - the 16 rows x 32 k of one 16 x 16 x 32
// operand fragment are ONE contiguous KiB in which lane (i16 = row % 16, g = (k % 32) / 8) owns bytes 16 lane .. 16 lane + 15.  An
// LDS-DMA instruction (lane l moves 16 bytes to LDS base + 16 l) then reads a KiB linearly and the fragment read is ds_read_b128 at
// base + 16 lane.  (Against the row-major [row][32 k] tile it replaced: the same kernel time, 148 us per call at 256 atoms x 256 rows -
// the layout is kept for its addressing, one add per piece.)
__host__ __device__ __forceinline__ int64_t frag_off(int64_t row_tiles, int row, int k) {
  return (((int64_t)(k >> 5) * row_tiles + (row >> 4)) * 64 + (((k & 31) >> 3) * 16 + (row & 15))) * 8 + (k & 7);
}

#define AH_SSCALE 1024.0f   // the scores (<= 1 in magnitude) x 2^10 before their fp16 split: lo halves stay normal

// ------------------------------------------------------------------------------------------------
// compute_kernel_attention_scores  (kernel_attention.py:69-121)
// one workgroup per conditioning row; out [B,H,V,V]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pair_distance(const float* x, int q, int m, int use_mm) {
  float qx = x[3 * q], qy = x[3 * q + 1], qz = x[3 * q + 2];
  float mx = x[3 * m], my = x[3 * m + 1], mz = x[3 * m + 2];
  if (!use_mm) {
    float dx = qx - mx, dy = qy - my, dz = qz - mz;
    return sqrtf(dx * dx + dy * dy + dz * dz);
  }
  return tw_cdist_mm(qx, qy, qz, mx, my, mz);
}

__global__ void scores_kernel(const float* __restrict__ x, const uint8_t* __restrict__ masked,
                              const float* __restrict__ ls, int H, int V, int normalise, int use_mm,
                              float* __restrict__ out, const float* __restrict__ coeffs, int order, int force_zero) {
  extern __shared__ float sm[];
  float* xs = sm;            // [V*3]
  float* dist = sm + 3 * V;  // [V*V]
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) xs[i] = x[b * 3 * V + i];
  __syncthreads();
  for (int i = threadIdx.x; i < V * V; i += blockDim.x) dist[i] = pair_distance(xs, i / V, i % V, use_mm);
  __syncthreads();
  // one thread per (h, q) row
  for (int r = threadIdx.x; r < H * V; r += blockDim.x) {
    const int h = r / V, q = r % V;
    const float l = ls[h];
    const float* cf = order > 0 ? coeffs + (int64_t)h * order : nullptr;
    float cmean = 0.f;
    if (order > 0 && force_zero) {
      for (int c = 0; c < order; ++c) cmean += cf[c];
      cmean /= (float)order;
    }
    double sum = 0.0;  // exact sum, rounded once: the neutral statement of torch's (vectorised, non-sequential) reduction
    for (int m = 0; m < V; ++m) {
      float sc = dist[q * V + m] / l;
      float e = masked[b * V + m] ? 0.f : basis_value(sc, cf, order, cmean);
      sum += (double)fabsf(e);
    }
    const float denom = (float)sum + 1e-5f;
    float* o = out + ((b * H + h) * V + q) * (int64_t)V;
    for (int m = 0; m < V; ++m) {
      float sc = dist[q * V + m] / l;
      float e = masked[b * V + m] ? 0.f : basis_value(sc, cf, order, cmean);
      o[m] = normalise ? e / denom : e;
    }
  }
}

// Large molecules (r05 / r06): the same rows without the V x V distance tile - the coordinates (12 V bytes) are all the LDS a block
// holds.  ONE WAVE per (head, query) row, lanes over the keys: every basis value is computed once and stays in the lane's registers
// (up to 16 per lane = 1024 atoms; beyond that they are recomputed) between the row sum and the division, reads and writes of a row are
// coalesced, and a single conditioning state already is H V waves (the thread-per-row form this replaces took as long as one thread
// needs for 2 V basis values whatever the launch: 210 us for one state of 256 atoms, 1.3 ms for 256 states, its stores 4 bytes per
// lane V floats apart).  The row sum: per-lane partial sums in double in key order, then a butterfly over the wave - the same exact
// sum rounded once up to the order of the double additions (scores_kernel adds in key order: equal floats except where the
// exact sum lies within 2^-29 of a rounding boundary; tests/test_flow_gpu.py::test_per_op_path_large_molecules holds them to one
// float ulp).  grid (B, blocks of 16 rows).
#define SCORES_ROWS_PER_BLOCK 16
#define SCORES_KEEP 16
__global__ void __launch_bounds__(64 * SCORES_ROWS_PER_BLOCK) scores_rows_kernel(
    const float* __restrict__ x, const uint8_t* __restrict__ masked, const float* __restrict__ ls, int H, int V, int normalise, int use_mm,
    float* __restrict__ out, const float* __restrict__ coeffs, int order, int force_zero, _Float16* __restrict__ hi,
    _Float16* __restrict__ lo) {
  // hi / lo (optional, instead of out): the scores x 2^10 as split fp16 in fragment order, keys padded to Vp = 32 ceil(V / 32) with
  // zeros - what split_scores_kernel makes of `out`, without the fp32 round trip (padded QUERY rows stay unwritten: a query is a
  // column of the mixing's B operand, and columns past the molecule are dropped)
  extern __shared__ float sm[];
  float* xs = sm;  // [V*3]
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) xs[i] = x[b * 3 * V + i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.y * SCORES_ROWS_PER_BLOCK + (threadIdx.x >> 6);  // one wave per (h, q) row
  if (r >= H * V) return;
  const int h = r / V, q = r % V;
  const float l = ls[h];
  const float* cf = order > 0 ? coeffs + (int64_t)h * order : nullptr;
  float cmean = 0.f;
  if (order > 0 && force_zero) {
    for (int c = 0; c < order; ++c) cmean += cf[c];
    cmean /= (float)order;
  }
  auto value = [&](int m) -> float {
    const float sc = pair_distance(xs, q, m, use_mm) / l;
    return masked[b * V + m] ? 0.f : basis_value(sc, cf, order, cmean);
  };
  float keep[SCORES_KEEP];
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < SCORES_KEEP; ++k) {
    const int m = lane + 64 * k;
    keep[k] = m < V ? value(m) : 0.f;
    sum += (double)fabsf(keep[k]);
  }
  for (int m = lane + 64 * SCORES_KEEP; m < V; m += 64) sum += (double)fabsf(value(m));
  sum = wave_sum(sum);
  const float denom = (float)sum + 1e-5f;
  const int Vp = (V + 31) / 32 * 32;
  const int64_t mat = (b * H + h) * (int64_t)Vp * Vp;
  auto put = [&](int m, float v) {
    if (hi) {
      const float sv = v * AH_SSCALE;
      const _Float16 hh = (_Float16)sv;
      const int64_t off = mat + frag_off(Vp / 16, q, m);
      hi[off] = hh;
      lo[off] = (_Float16)(sv - (float)hh);
    } else {
      out[((b * H + h) * V + q) * (int64_t)V + m] = v;
    }
  };
#pragma unroll
  for (int k = 0; k < SCORES_KEEP; ++k) {
    const int m = lane + 64 * k;
    if (m < V) put(m, normalise ? keep[k] / denom : keep[k]);
    else if (hi && m < Vp) put(m, 0.f);
  }
  for (int m = lane + 64 * SCORES_KEEP; m < Vp; m += 64) {
    if (m >= V) {
      if (hi) put(m, 0.f);
      continue;
    }
    const float e = value(m);
    put(m, normalise ? e / denom : e);
  }
}

// The small-molecule per-op kernels hold one molecule's V x V score / distance matrix in LDS.  Up to 64 KiB of dynamic LDS
// launches as is; up to the CU's 160 KiB after raising the kernel's limit; beyond that the kernel-attention flow takes the
// tiled kernels (scores_rows_kernel, attend_mfma_kernel: any V), the dense flow sdpa_rows_kernel.
#define TW_LDS_LIMIT(kernel, bytes, V)                                                                        \
  do {                                                                                                        \
    TW_REQUIRE((bytes) <= (size_t)160 * 1024,                                                                 \
               "n_atoms = %d needs %zu bytes of LDS per molecule on the per-op path (V x V scores), the CU has 163840", \
               (int)(V), (size_t)(bytes));                                                                    \
    if ((bytes) > (size_t)64 * 1024)                                                                          \
      TW_HIP_CHECK(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
  } while (0)

// s_hi / s_lo (optional; the row-wise kernel only - scores_split_direct): the split fp16 operand of the folded mixing instead of `out`
bool scores_split_direct(int V) { return V > 160 && !(g_debug_flags & 2097152); }
int launch_scores(const float* x, const uint8_t* masked, const float* ls, int H, int64_t B, int V,
                  int normalise, int use_mm, float* out, hipStream_t s, const float* coeffs, int order, int force_zero, _Float16* s_hi,
                  _Float16* s_lo) {
  if (B == 0) return TW_OK;
  TW_REQUIRE(!s_hi || scores_split_direct(V), "scores: split output asked of the tile kernel (%d atoms)", V);
  size_t shm = (size_t)(3 * V + V * V) * sizeof(float);
  // The row-wise kernel: no room for the distance tile (or bit 21) - and from 161 atoms on anyway: the tile kernel is one workgroup
  // per conditioning state, one thread per (head, query) row (200 atoms, one state: 590 us on one CU); the row-wise one a wave per row.
  if (shm > (size_t)160 * 1024 || V > 160 || (g_debug_flags & 2097152)) {
    hipLaunchKernelGGL(scores_rows_kernel, dim3((unsigned)B, (unsigned)((H * V + SCORES_ROWS_PER_BLOCK - 1) / SCORES_ROWS_PER_BLOCK)),
                       dim3(64 * SCORES_ROWS_PER_BLOCK), (size_t)3 * V * sizeof(float), s, x, masked, ls, H, V, normalise, use_mm, out, coeffs,
                       order, force_zero, s_hi, s_lo);
    TW_LAUNCH_CHECK();
    return TW_OK;
  }
  TW_LDS_LIMIT(scores_kernel, shm, V);
  hipLaunchKernelGGL(scores_kernel, dim3((unsigned)B), dim3(256), shm, s, x, masked, ls, H, V, normalise,
                     use_mm, out, coeffs, order, force_zero);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

// ------------------------------------------------------------------------------------------------
// get_centre_of_mass (molecule_utils.py:15-29)
// ------------------------------------------------------------------------------------------------
__global__ void centre_kernel(const float* __restrict__ x, const uint8_t* __restrict__ masked,
                              float* __restrict__ xc, float* __restrict__ com, int V) {
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x;
  float sx = 0, sy = 0, sz = 0, cnt = 0;
  for (int a = lane; a < V; a += 64) {
    float keep = masked[n * V + a] ? 0.f : 1.f;
    sx += keep * x[(n * V + a) * 3 + 0];
    sy += keep * x[(n * V + a) * 3 + 1];
    sz += keep * x[(n * V + a) * 3 + 2];
    cnt += keep;
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); cnt = wave_sum(cnt);
  const float cx = sx / cnt, cy = sy / cnt, cz = sz / cnt;
  if (com && lane == 0) { com[n * 3] = cx; com[n * 3 + 1] = cy; com[n * 3 + 2] = cz; }
  if (xc)
    for (int a = lane; a < V; a += 64) {
      xc[(n * V + a) * 3 + 0] = x[(n * V + a) * 3 + 0] - cx;
      xc[(n * V + a) * 3 + 1] = x[(n * V + a) * 3 + 1] - cy;
      xc[(n * V + a) * 3 + 2] = x[(n * V + a) * 3 + 2] - cz;
    }
}

int launch_centre(const float* x, const uint8_t* masked, float* xc, float* com, int64_t n, int V,
                  hipStream_t s) {
  if (n == 0) return TW_OK;
  hipLaunchKernelGGL(centre_kernel, dim3((unsigned)n), dim3(64), 0, s, x, masked, xc, com, V);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

// ------------------------------------------------------------------------------------------------
// affine coupling + log-det (layers/nvp.py:89-183); one wave per row
//   forward: z' = z*exp(s)+t, logdet = +sum log(exp(s)); reverse: z' = (z-t)/exp(s), logdet = -sum
//   delta_logp -= logdet (nvp.py:86)
// ------------------------------------------------------------------------------------------------
// Sticky per-device flag: some coupling net returned a non-finite scale or shift.  The split-fp16 path keeps its
// operands in fp16 (|value| < 65504); a checkpoint whose activations leave that range shows up here as inf/NaN, and
// the host can tell the user to switch to the exact-f32 path instead of sampling with NaN log-densities
// (tw_flow_nonfinite; checked where the MH loop synchronises anyway).
__device__ int g_nonfinite = 0;

int nonfinite_flag(int reset, int* out) {
  int v = 0;
  TW_HIP_CHECK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_nonfinite), sizeof(int)));
  if (reset && v) {
    const int zero = 0;
    TW_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_nonfinite), &zero, sizeof(int)));
  }
  *out = v;
  return TW_OK;
}

int* nonfinite_flag_device_ptr() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_nonfinite)) != hipSuccess) return nullptr;
  return (int*)p;
}

__global__ void coupling_kernel(const float* __restrict__ s_raw, const float* __restrict__ t,
                                const uint8_t* __restrict__ masked, int64_t n_cond, const float* z_in, float* z,
                                float* __restrict__ delta_logp, int V, int reverse, int* __restrict__ flag) {
  const int64_t n = blockIdx.x;
  const int64_t c = n % n_cond;
  float acc = 0.f;
  bool bad = false;
  for (int i = threadIdx.x; i < 3 * V; i += 64) {
    const int64_t idx = n * 3 * V + i;
    const float scale = expf(s_raw[idx]);
    const float shift = t[idx];
    bad |= !(isfinite(s_raw[idx]) && isfinite(shift));
    const float keep = masked[c * V + i / 3] ? 0.f : 1.f;
    acc += logf(scale) * keep;
    z[idx] = reverse ? (z_in[idx] - shift) / scale : z_in[idx] * scale + shift;
  }
  if (bad) atomicOr(flag ? flag : &g_nonfinite, 1);
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    const float logdet = reverse ? -acc : acc;
    delta_logp[n] = delta_logp[n] - logdet;
  }
}

int launch_coupling(const float* s_raw, const float* t, const uint8_t* masked, int64_t n_cond, float* z,
                    float* delta_logp, int64_t n_rows, int V, int reverse, hipStream_t s, const float* z_in, int* flag) {
  if (n_rows == 0) return TW_OK;
  hipLaunchKernelGGL(coupling_kernel, dim3((unsigned)n_rows), dim3(64), 0, s, s_raw, t, masked, n_cond, z_in ? z_in : z, z,
                     delta_logp, V, reverse, flag);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

// ------------------------------------------------------------------------------------------------
// prior log-density + final assembly (flow.py:191-203, 303-334)
// Normal(0, e^ls).log_prob(z) = -z^2/(2 var) - log(scale) - log(sqrt(2 pi))
// ------------------------------------------------------------------------------------------------
__global__ void prior_logp_kernel(const float* __restrict__ zc, const float* __restrict__ zv,
                                  const uint8_t* __restrict__ masked, int64_t n_cond,
                                  const float* __restrict__ prior, const float* __restrict__ delta,
                                  float sign, float* __restrict__ out, int V) {
  const int64_t n = blockIdx.x;
  const int64_t c = n % n_cond;
  const float sc = expf(prior[0]), sv = expf(prior[1]);
  const float var_c = sc * sc, var_v = sv * sv;
  const float lsc = logf(sc), lsv = logf(sv);
  const float half_log_2pi = 0.91893853320467274178f;
  float ac = 0.f, av = 0.f;
  for (int i = threadIdx.x; i < 3 * V; i += 64) {
    const float keep = masked[c * V + i / 3] ? 0.f : 1.f;
    const float a = zc[n * 3 * V + i], b = zv[n * 3 * V + i];
    ac += keep * (-(a * a) / (2.f * var_c) - lsc - half_log_2pi);
    av += keep * (-(b * b) / (2.f * var_v) - lsv - half_log_2pi);
  }
  ac = wave_sum(ac);
  av = wave_sum(av);
  if (threadIdx.x == 0) out[n] = (ac + av) + sign * delta[n];
}

// y = (x_centred + com) + residual  (flow.py:303-310)
__global__ void uncentre_add_kernel(const float* __restrict__ xc, const float* __restrict__ com,
                                    const float* __restrict__ resid, int64_t n_cond, float* __restrict__ y,
                                    int V, int displacement, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t n = i / (3 * V);
  const int r = (int)(i % (3 * V));
  const int64_t c = n % n_cond;
  float base = xc[c * 3 * V + r] + com[c * 3 + r % 3];
  y[i] = displacement ? base + resid[i] : resid[i];
}

__global__ void sub_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                           int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) o[i] = a[i] - b[i];
}

// ------------------------------------------------------------------------------------------------
// compute_kinetic_energy (evaluation_utils.py:416-436)
// ------------------------------------------------------------------------------------------------
__global__ void kinetic_kernel(const float* __restrict__ v, const float* __restrict__ masses, int random_velocs,
                               float kbT, float* __restrict__ out, int V) {
  const int64_t n = blockIdx.x;
  float acc = 0.f;
  for (int a = threadIdx.x; a < V; a += 64) {
    const float* p = v + (n * V + a) * 3;
    float s = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    acc += random_velocs ? s : masses[a] * s;
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[n] = random_velocs ? 0.5f * acc : 0.5f * acc / kbT;
}

// ------------------------------------------------------------------------------------------------
// MH accept step (evaluation_utils.py:659-677): single workgroup, first-true scan + state update
// ------------------------------------------------------------------------------------------------
// One block per chain c; proposal s of chain c lives at index s * n_chains + c of every per-proposal array
// (the row order of a [S, C] flow call); n_chains = 1 is the reference's single chain.
__global__ void mh_accept_kernel(const float* __restrict__ energy, const float* __restrict__ p_xy,
                                 const float* __restrict__ p_yx, const float* __restrict__ u,
                                 const float* __restrict__ yc, const float* __restrict__ yv,
                                 float* __restrict__ xc, float* __restrict__ xv, float* __restrict__ out_exp,
                                 float* __restrict__ out_pacc, uint8_t* __restrict__ out_acc,
                                 int32_t* __restrict__ result, int64_t S, int V, int64_t C) {
  __shared__ int first;
  const int64_t c = blockIdx.x;
  if (threadIdx.x == 0) first = 0x7fffffff;
  __syncthreads();
  int local = 0x7fffffff;
  for (int64_t s = threadIdx.x; s < S; s += blockDim.x) {
    const int64_t i = s * C + c;
    const float e = energy[i] + p_xy[i] - p_yx[i];
    // torch.min(1, exp(-e)) propagates NaN and `rand < NaN` is False (evaluation_utils.py:665-668): a proposal with
    // a non-finite exponent is rejected.  fminf alone would return 1 for a NaN exponent and accept it.
    const float ee = expf(-e);
    const float p = (ee != ee) ? ee : fminf(1.f, ee);
    const bool acc = u[i] < p;
    out_exp[i] = e;
    out_pacc[i] = p;
    out_acc[i] = acc ? 1 : 0;
    if (acc && (int)s < local) local = (int)s;
  }
  atomicMin(&first, local);
  __syncthreads();
  const int k = first;
  const bool any = k != 0x7fffffff;
  if (any && xc && xv)
    for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) {
      xc[c * 3 * V + i] = yc[((int64_t)k * C + c) * 3 * V + i];
      xv[c * 3 * V + i] = yv[((int64_t)k * C + c) * 3 * V + i];
    }
  if (threadIdx.x == 0) {
    result[4 * c + 0] = any ? k : (int)(S - 1);
    result[4 * c + 1] = any ? 1 : 0;
    result[4 * c + 2] = 0;
    result[4 * c + 3] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// chirality (utils/chirality.py:40-80): sign of (b-a) . ((c-a) x (d-a)) per centre
// ------------------------------------------------------------------------------------------------
__global__ void chirality_kernel(const float* __restrict__ coords, const int32_t* __restrict__ centres,
                                 const float* __restrict__ ref, int n_centres, uint8_t* __restrict__ changed,
                                 int64_t n_rows, int V) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_rows) return;
  const float* x = coords + n * 3 * V;
  bool ch = false;
  for (int c = 0; c < n_centres; ++c) {
    const int i0 = centres[4 * c], i1 = centres[4 * c + 1], i2 = centres[4 * c + 2], i3 = centres[4 * c + 3];
    // direction vectors from the first listed atom to the other three (chirality.py:52-60)
    float a[3], b[3], d[3];
    for (int k = 0; k < 3; ++k) {
      a[k] = x[3 * i1 + k] - x[3 * i0 + k];
      b[k] = x[3 * i2 + k] - x[3 * i0 + k];
      d[k] = x[3 * i3 + k] - x[3 * i0 + k];
    }
    // perm_sign = d0 . (d1 x d2)  (chirality.py:57-61)
    float cx = b[1] * d[2] - b[2] * d[1];
    float cy = b[2] * d[0] - b[0] * d[2];
    float cz = b[0] * d[1] - b[1] * d[0];
    float dot = a[0] * cx + a[1] * cy + a[2] * cz;
    float sgn = (dot > 0.f) ? 1.f : ((dot < 0.f) ? -1.f : 0.f);
    if (sgn != ref[c]) ch = true;
  }
  changed[n] = ch ? 1 : 0;
}

// ================================================================================================
// SIMPLE PATH kernels
// ================================================================================================

// u[n, a, :] = cat(emb[type], x_coords, x_velocs, z_other [, rff(x_coords)])
// (custom_transformer_nvp.py:64-71, transformer_nvp.py:76-88)
__global__ void build_input_kernel(const float* __restrict__ emb, const int32_t* __restrict__ types,
                                   const float* __restrict__ xc, const float* __restrict__ xv,
                                   const float* __restrict__ z_other, const float* __restrict__ rff_vec,
                                   int d_rff, int64_t n_cond, int V, int d_emb, int d_in, float* __restrict__ u,
                                   int64_t tokens) {
  const int64_t tok = blockIdx.x;
  if (tok >= tokens) return;
  const int64_t n = tok / V;
  const int a = (int)(tok % V);
  const int64_t c = n % n_cond;
  const int ty = types[c * V + a];
  const float* px = xc + (c * V + a) * 3;
  for (int f = threadIdx.x; f < d_in; f += blockDim.x) {
    float val;
    if (f < d_emb) val = emb[ty * d_emb + f];
    else if (f < d_emb + 3) val = px[f - d_emb];
    else if (f < d_emb + 6) val = xv[(c * V + a) * 3 + f - d_emb - 3];
    else if (f < d_emb + 9) val = z_other[tok * 3 + f - d_emb - 6];
    else {
      // rff_position_encoder.py:57-62: sqrt(1/n) * [cos(x G), sin(x G)]
      const int nvec = d_rff / 2;
      const int j = f - d_emb - 9;
      const int col = j % nvec;
      float ip = px[0] * rff_vec[0 * nvec + col] + px[1] * rff_vec[1 * nvec + col] + px[2] * rff_vec[2 * nvec + col];
      val = sqrtf(1.0f / nvec) * (j < nvec ? cosf(ip) : sinf(ip));
    }
    u[tok * d_in + f] = val;
  }
}

// Y[M,N] = act(X[M,K] W[N,K]^T + b) on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32
// accumulate).  Block = 4 waves = a 128 x 64 output tile; wave w owns rows 32w..32w+31 and all 64 columns (2 x 4
// 16 x 16 accumulators, each A / B fragment reused 4 / 2 times); X and W tiles of 16 k-values are staged through LDS,
// k-major with padded rows (fragment reads: lanes read consecutive m / n for a fixed k - conflict free).
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };
typedef float lin_f4 __attribute__((ext_vector_type(4)));
#define LIN_BM 128
#define LIN_BN 64
template <int ACT>
__global__ void __launch_bounds__(256) linear_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ Y,
                                                      int64_t M, int N, int K) {
  __shared__ float xs[16][LIN_BM + 4];
  __shared__ float wsh[16][LIN_BN + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * LIN_BM;
  const int n0 = blockIdx.x * LIN_BN;
  lin_f4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (lin_f4){0.f, 0.f, 0.f, 0.f};
  const bool k_vec = (K % 4) == 0;  // rows of X / W are 16-byte aligned: stage with float4 loads
  // One K = 16 slice of the tile is (128 + 64) rows x 4 float4 = 3 float4 per thread.  The NEXT slice's global loads are issued
  // before the current slice's MFMAs and parked in registers (r05: without that every slice paid a global round trip in the
  // open - 43 % of the f32 MFMA line at 192 atoms, where this kernel is 77 % of a per-op pass); same values, same order of
  // operations, bit-identical results.
  lin_f4 nxt[3];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int i = threadIdx.x + 256 * t;
      const bool is_x = i < LIN_BM * 4;
      const int r = (is_x ? i : i - LIN_BM * 4) / 4, kq = 4 * (i % 4);
      const int64_t row = is_x ? m0 + r : (int64_t)n0 + r;
      const bool row_ok = is_x ? row < M : row < N;
      const float* src = (is_x ? X : W) + row * K + k0 + kq;
      lin_f4 v = (lin_f4){0.f, 0.f, 0.f, 0.f};
      if (row_ok) {
        if (k_vec && k0 + kq + 3 < K) {
          v = *(const lin_f4*)src;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + kq + e < K) v[e] = src[e];
        }
      }
      nxt[t] = v;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int i = threadIdx.x + 256 * t;
      const bool is_x = i < LIN_BM * 4;
      const int r = (is_x ? i : i - LIN_BM * 4) / 4, kq = 4 * (i % 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (is_x) xs[kq + e][r] = nxt[t][e];
        else wsh[kq + e][r] = nxt[t][e];
      }
    }
    __syncthreads();
    if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // A fragment: lane (i16, g) holds X[m][k = 4 ks + g]; B fragment: W[n = 16 j + i16][k = 4 ks + g]
      float a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = xs[4 * ks + g][32 * wave + 16 * i + i16];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = wsh[4 * ks + g][16 * j + i16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D layout: lane (i16, g) holds rows 4 g + r (r = 0..3) of column i16
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 16 * j + i16;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 32 * wave + 16 * i + 4 * g + r;
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
        if (ACT == ACT_SILU) v = v / (1.f + expf(-v));
        Y[m * N + n] = v;
      }
  }
}

// The same GEMM for TW_PATH_SIMPLE_H3 (r06): every fp32 product as THREE v_mfma_f32_16x16x32_f16 on fp16 hi / lo splits
// (hi.hi + hi.lo + lo.hi, fp32 accumulate) - the arithmetic of the fused split-fp16 kernels, 16x the matrix rate of the fp32
// form for 3x the instructions.  X and W are split while they are staged: fp32 tile -> registers -> (hi, lo) fp16 tiles in the
// LDS, row-major [row][32 k] with rows padded to 40 halves (16-byte fragment reads of lanes (i16, g) then spread over the banks);
// W is multiplied by 2^8 before the split so that its lo halves stay normal (|w| ~ 0.1 -> lo ~ 6e-3; 2^-8 is applied to the
// fp32 sums, exact; |w| >= 256 overflows fp16 -> non-finite outputs -> the coupling step raises the range flag and the caller
// falls back to the exact kernels, as for the fused split-fp16 path).  Workgroup = 4 waves = a 128 x 128 output tile, wave
// (wm, wn) owns 64 x 64 (4 x 4 accumulators, 48 MFMAs per k-step of 32 against 16 fragment reads); the LDS tiles are double
// buffered (one barrier per k-step) and the NEXT k-step's global loads are in flight during the MFMAs.
typedef _Float16 lin_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 lin_h4 __attribute__((ext_vector_type(4)));
#define LH_BM 128
#define LH_BN 128
#define LH_ROW 40   // halves per LDS row: 32 + 8 of padding
#define LH_WSCALE 256.0f
template <int ACT>
__global__ void __launch_bounds__(256) linear_h3_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Y,
                                                         int64_t M, int N, int K) {
  // [buffer][X | W][hi | lo][row][k]: 80 KiB of dynamic LDS (launch_linear raises the kernel's limit)
  extern __shared__ __attribute__((aligned(16))) char lh_lds[];
  _Float16 (*tile)[2][2][LH_BM][LH_ROW] = (_Float16 (*)[2][2][LH_BM][LH_ROW])lh_lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * LH_BM;
  const int n0 = blockIdx.x * LH_BN;
  lin_f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (lin_f4){0.f, 0.f, 0.f, 0.f};
  const bool k_vec = (K % 4) == 0;
  // one k-step of the tile: (128 + 128) rows x 8 float4 = 8 float4 per thread: t = 0..3 rows of X, 4..7 rows of W
  lin_f4 nxt[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = threadIdx.x + 256 * (t & 3);
      const bool is_x = t < 4;
      const int r = i >> 3, kq = 4 * (i & 7);
      const int64_t row = is_x ? m0 + r : (int64_t)n0 + r;
      const bool row_ok = is_x ? row < M : row < N;
      const float* src = (is_x ? X : W) + row * K + k0 + kq;
      lin_f4 v = (lin_f4){0.f, 0.f, 0.f, 0.f};
      if (row_ok) {
        if (k_vec && k0 + kq + 3 < K) {
          v = *(const lin_f4*)src;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + kq + e < K) v[e] = src[e];
        }
      }
      nxt[t] = v;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = threadIdx.x + 256 * (t & 3);
      const int which = t < 4 ? 0 : 1;
      const int r = i >> 3, kq = 4 * (i & 7);
      lin_h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = which ? nxt[t][e] * LH_WSCALE : nxt[t][e];
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
      }
      *(lin_h4*)&tile[buf][which][0][r][kq] = hi;
      *(lin_h4*)&tile[buf][which][1][r][kq] = lo;
    }
  };
  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const bool more = k0 + 32 < K;
    if (more) fetch(k0 + 32);
    // fragments: lane (i16, g) holds k = 8 g .. 8 g + 7 of row i16 of a 16-row tile - the same k order for A and B
    lin_h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 64 * wm + 16 * i + i16;
      ah[i] = *(const lin_h8*)&tile[buf][0][0][r][8 * g];
      al[i] = *(const lin_h8*)&tile[buf][0][1][r][8 * g];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 64 * wn + 16 * j + i16;
      bh[j] = *(const lin_h8*)&tile[buf][1][0][r][8 * g];
      bl[j] = *(const lin_h8*)&tile[buf][1][1][r][8 * g];
    }
    // D = A B^T with A = X rows (tokens), B = W rows (output units): D[m][n], lane (i16, g) holds rows 4 g + r of column i16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
      }
    if (more) stage(buf ^ 1);   // the other buffer: its last readers passed the barrier of the previous k-step
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 64 * wn + 16 * j + i16;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 64 * wm + 16 * i + 4 * g + r;
        if (m >= M) continue;
        float v = acc[i][j][r] * (1.0f / LH_WSCALE) + bv;
        if (ACT == ACT_RELU) v = fmaxf(v, 0.f);
        if (ACT == ACT_SILU) v = v / (1.f + expf(-v));
        Y[m * N + n] = v;
      }
  }
}

// Wc[n][h D + k] = sum_j W_o[n][h D + j] W_v[h D + j][k]  (fp64 sums, as the fused kernels' pack does): the value and output
// projections of kernel attention folded per head (kernel_attention.py:124-156: out_proj(flatten(A_h (x W_v,h^T)))) = sum_h (A_h x) Wc_h^T
__global__ void fold_vo_kernel(const float* __restrict__ raw, int64_t first_net, int64_t coupling_size, int64_t net_size,
                               int64_t layers_off, int64_t layer_size, int64_t wv_off, int64_t wo_off, int n_layers, int H, int D,
                               float* __restrict__ out, _Float16* __restrict__ wcp_hi, _Float16* __restrict__ wcp_lo) {
  // blockIdx.x = ((c * 2 + net) * n_layers + l) * H + h;  one thread per (n, k) of the head's D x D block
  int64_t b = blockIdx.x;
  const int h = (int)(b % H); b /= H;
  const int l = (int)(b % n_layers); b /= n_layers;
  const int net = (int)(b % 2);
  const int64_t c = b / 2;
  const float* lb = raw + first_net + c * coupling_size + net * net_size + layers_off + (int64_t)l * layer_size;
  const float* wv = lb + wv_off;   // [H D, D]
  const float* wo = lb + wo_off;   // [D, H D]
  float* o = out + (((c * 2 + net) * n_layers + l) * (int64_t)D) * H * D;   // [D, H D]
  // ... and, for attend_fold_h3_kernel, as split fp16 (x 2^8) per head [H][D][D] with the k index in MFMA-operand order: position
  // 32 ks + 8 g + e holds feature 32 ks + 16 (e / 4) + 4 g + e % 4 - the order in which a lane's accumulator registers of the mixing
  // (feature tiles 2 ks, 2 ks + 1, rows 4 g + r) become its B operand, so the matching A operand is one 16-byte read
  _Float16* ph = wcp_hi ? wcp_hi + ((((c * 2 + net) * n_layers + l) * (int64_t)H + h) * D) * D : nullptr;
  _Float16* pl = wcp_lo ? wcp_lo + ((((c * 2 + net) * n_layers + l) * (int64_t)H + h) * D) * D : nullptr;
  for (int i = threadIdx.x; i < D * D; i += blockDim.x) {
    const int n = i / D, k = i % D;
    double acc = 0.0;
    for (int j = 0; j < D; ++j) acc += (double)wo[(int64_t)n * H * D + h * D + j] * (double)wv[(int64_t)(h * D + j) * D + k];
    o[(int64_t)n * H * D + h * D + k] = (float)acc;
    if (ph) {
      const int ks = k / 32, w = k % 32, half = w / 16, g = (w % 16) / 4, r = w % 4;
      const int pos = 32 * ks + 8 * g + 4 * half + r;
      const float v = (float)acc * LH_WSCALE;
      const _Float16 hh = (_Float16)v;
      ph[frag_off(D / 16, n, pos)] = hh;
      pl[frag_off(D / 16, n, pos)] = (_Float16)(v - (float)hh);
    }
  }
}

// floats behind the split-fp16 stream in the tw_flow_pack_simple_h3 buffer: Wc fp32 [c][net][l][D][H D], then its split, permuted
// fp16 copies (hi, lo: half as many floats each)
static int64_t fold_wc_floats(const tw_flow_desc& d) { return (int64_t)d.n_coupling * 2 * d.n_layers * d.d_model * d.n_heads * d.d_model; }
int64_t simple_h3_fold_floats(const tw_flow_desc& d) { return d.variant == 0 ? 2 * fold_wc_floats(d) : 0; }
int64_t simple_h3_split_offset(const tw_flow_desc& d) {
  return ((h3_packed_bytes(d, false) + 255) / 256 * 256 + simple_h3_fold_floats(d) * 4 + 255) / 256 * 256;
}

int simple_h3_fold(const tw_flow_desc& d, const float* raw, float* out, hipStream_t s) {
  if (d.variant != 0) return TW_OK;
  const RawLayout L = raw_layout(d);
  const int64_t blocks = (int64_t)d.n_coupling * 2 * d.n_layers * d.n_heads;
  _Float16* hi = (_Float16*)(out + fold_wc_floats(d));
  hipLaunchKernelGGL(fold_vo_kernel, dim3((unsigned)blocks), dim3(256), 0, s, raw, L.chain + L.nets, L.coupling_size, L.net.size,
                     L.net.layers, L.layer.size, L.layer.wv, L.layer.wo, d.n_layers, d.n_heads, d.d_model, out, hi, hi + fold_wc_floats(d));
  TW_LAUNCH_CHECK();
  return TW_OK;
}

static int launch_linear(const float* X, const float* W, const float* b, float* Y, int64_t M, int N, int K,
                         int act, hipStream_t s, bool split = false) {
  if (split && N >= 32) {   // (the 3-column head of the out-MLP stays on the fp32 form: one sixteenth of a 128-wide tile)
    dim3 gridh((N + LH_BN - 1) / LH_BN, (unsigned)((M + LH_BM - 1) / LH_BM));
    constexpr int lds = 2 * 2 * 2 * LH_BM * LH_ROW * (int)sizeof(_Float16);
    static LdsLimit lim[3];
    int lrc;
    if (act == ACT_NONE) {
      if ((lrc = lim[0].ensure((const void*)linear_h3_kernel<ACT_NONE>, lds))) return lrc;
      hipLaunchKernelGGL(linear_h3_kernel<ACT_NONE>, gridh, dim3(256), lds, s, X, W, b, Y, M, N, K);
    } else if (act == ACT_RELU) {
      if ((lrc = lim[1].ensure((const void*)linear_h3_kernel<ACT_RELU>, lds))) return lrc;
      hipLaunchKernelGGL(linear_h3_kernel<ACT_RELU>, gridh, dim3(256), lds, s, X, W, b, Y, M, N, K);
    } else {
      if ((lrc = lim[2].ensure((const void*)linear_h3_kernel<ACT_SILU>, lds))) return lrc;
      hipLaunchKernelGGL(linear_h3_kernel<ACT_SILU>, gridh, dim3(256), lds, s, X, W, b, Y, M, N, K);
    }
    TW_LAUNCH_CHECK();
    return TW_OK;
  }
  dim3 grid((N + LIN_BN - 1) / LIN_BN, (unsigned)((M + LIN_BM - 1) / LIN_BM));
  if (act == ACT_NONE) hipLaunchKernelGGL(linear_kernel<ACT_NONE>, grid, dim3(256), 0, s, X, W, b, Y, M, N, K);
  else if (act == ACT_RELU) hipLaunchKernelGGL(linear_kernel<ACT_RELU>, grid, dim3(256), 0, s, X, W, b, Y, M, N, K);
  else hipLaunchKernelGGL(linear_kernel<ACT_SILU>, grid, dim3(256), 0, s, X, W, b, Y, M, N, K);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

// attend + flatten_multihead (kernel_attention.py:124-156):
// att[n, q, h*D + d] = sum_m scores[n % B, h, q, m] * vals[n, m, h*D + d]; grid (n, h), block D threads
__global__ void attend_kernel(const float* __restrict__ scores, const float* __restrict__ vals,
                              float* __restrict__ att, int64_t n_cond, int H, int V, int D) {
  extern __shared__ float sc[];  // [V*V]
  const int64_t n = blockIdx.x;
  const int h = blockIdx.y;
  const int64_t c = n % n_cond;
  for (int i = threadIdx.x; i < V * V; i += blockDim.x) sc[i] = scores[((c * H + h) * V) * (int64_t)V + i];
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    for (int q = 0; q < V; ++q) {
      float acc = 0.f;
      for (int m = 0; m < V; ++m) acc = fmaf(sc[q * V + m], vals[((n * V + m) * H + h) * (int64_t)D + d], acc);
      att[((n * V + q) * H + h) * (int64_t)D + d] = acc;
    }
  }
}

// The mixing GEMM of TW_PATH_SIMPLE_H3 (r06): per (row n, head h)  att[q, d] = sum_m scores[q, m] * vals[m, d]  as split-fp16 MFMAs
// (three v_mfma_f32_16x16x32_f16 per fp32 product, as linear_h3_kernel).  A operand = score rows (m contiguous: staged like
// linear_h3_kernel's X, multiplied by 2^10 before the split - scores are in [0, 1], entries of a 200-atom row ~5e-3 would have
// subnormal lo halves otherwise); B operand = vals^T: lane (d, g) needs eight consecutive m of ONE feature, so the 32 x 128 slice of
// vals is transposed on its way into the LDS - each thread loads rows m, m + 1 of four features and writes (m, m + 1) half pairs.
// Workgroup = 128 queries x 128 features, wave (wm, wn) 64 x 64.  1-D grid over (n, h, query tile, feature tile).
// (AH_SSCALE: defined with frag_off above)
// `vrow` / `vhead`: floats between consecutive keys of `vals` and between heads - (H D, D) for the projected values [n, m, h, d];
// (D, 0) for the FOLDED form, where every head mixes the layer input x [n, m, d] itself and the per-head value and output
// projections are one 768 -> 128 GEMM behind the mixing (Wc_h = W_o,h W_v,h, tw_flow_pack_simple_h3).
__global__ void __launch_bounds__(256) attend_h3_kernel(const float* __restrict__ scores, const float* __restrict__ vals,
                                                         float* __restrict__ att, int64_t n_cond, int H, int V, int D,
                                                         int64_t vrow, int64_t vhead) {
  extern __shared__ __attribute__((aligned(16))) char lh_lds[];
  _Float16 (*tile)[2][2][LH_BM][LH_ROW] = (_Float16 (*)[2][2][LH_BM][LH_ROW])lh_lds;   // [buffer][S | vals^T][hi | lo][row][k]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (D + LH_BN - 1) / LH_BN, tiles_m = (V + LH_BM - 1) / LH_BM;
  int64_t blk = blockIdx.x;
  const int tn = (int)(blk % tiles_n); blk /= tiles_n;
  const int tm = (int)(blk % tiles_m); blk /= tiles_m;
  const int h = (int)(blk % H);
  const int64_t n = blk / H;
  const int64_t c = n % n_cond;
  const int q0 = tm * LH_BM, d0 = tn * LH_BN;
  const float* S = scores + ((c * H + h) * V) * (int64_t)V;
  const float* Vv = vals + n * V * vrow + h * vhead;
  const int64_t vstride = vrow;
  lin_f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (lin_f4){0.f, 0.f, 0.f, 0.f};
  // a k-step (32 keys): scores 128 q x 32 m = 4 float4 per thread (unaligned rows: scalar loads); vals 32 m x 128 d = 16 m-pairs x
  // 32 feature quads = 2 (pair, quad) items per thread, two float4 each
  float ns[4][4];
  lin_f4 nv[2][2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = threadIdx.x + 256 * t, r = i >> 3, kq = 4 * (i & 7);
#pragma unroll
      for (int e = 0; e < 4; ++e) ns[t][e] = (q0 + r < V && k0 + kq + e < V) ? S[(int64_t)(q0 + r) * V + k0 + kq + e] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = threadIdx.x + 256 * t, mp = i >> 5, dq = 4 * (i & 31);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int m = k0 + 2 * mp + u;
        lin_f4 v = (lin_f4){0.f, 0.f, 0.f, 0.f};
        if (m < V) {
          const float* src = Vv + (int64_t)m * vstride + d0 + dq;
          if (d0 + dq + 3 < D) v = *(const lin_f4*)src;
          else
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (d0 + dq + e < D) v[e] = src[e];
        }
        nv[t][u] = v;
      }
    }
  };
  typedef _Float16 lin_h2 __attribute__((ext_vector_type(2)));
  auto stage = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = threadIdx.x + 256 * t, r = i >> 3, kq = 4 * (i & 7);
      lin_h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = ns[t][e] * AH_SSCALE;
        const _Float16 hh = (_Float16)v;
        hi[e] = hh;
        lo[e] = (_Float16)(v - (float)hh);
      }
      *(lin_h4*)&tile[buf][0][0][r][kq] = hi;
      *(lin_h4*)&tile[buf][0][1][r][kq] = lo;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = threadIdx.x + 256 * t, mp = i >> 5, dq = 4 * (i & 31);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lin_h2 hi, lo;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float v = nv[t][u][e];
          const _Float16 hh = (_Float16)v;
          hi[u] = hh;
          lo[u] = (_Float16)(v - (float)hh);
        }
        *(lin_h2*)&tile[buf][1][0][dq + e][2 * mp] = hi;   // vals^T: row = feature, k = key
        *(lin_h2*)&tile[buf][1][1][dq + e][2 * mp] = lo;
      }
    }
  };
  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < V; k0 += 32) {
    const bool more = k0 + 32 < V;
    if (more) fetch(k0 + 32);
    lin_h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 64 * wm + 16 * i + i16;
      ah[i] = *(const lin_h8*)&tile[buf][0][0][r][8 * g];
      al[i] = *(const lin_h8*)&tile[buf][0][1][r][8 * g];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 64 * wn + 16 * j + i16;
      bh[j] = *(const lin_h8*)&tile[buf][1][0][r][8 * g];
      bl[j] = *(const lin_h8*)&tile[buf][1][1][r][8 * g];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
      }
    if (more) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int dd = d0 + 64 * wn + 16 * j + i16;
    if (dd >= D) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + 64 * wm + 16 * i + 4 * g + r;
        if (q < V) att[((n * V + q) * H + h) * (int64_t)D + dd] = acc[i][j][r] * (1.0f / AH_SSCALE);
      }
  }
}

// The folded form's operands prepared ONCE instead of per workgroup (attend_h3_kernel splits the score tile again for every row n
// that shares it and transposes the values on their way into the LDS):
//   split_scores_kernel   scores [B H, V, V] fp32 -> (hi, lo) fp16 [B H][Vp x Vp in fragment order (frag_off)], x 2^10, queries and
//                         keys padded with zeros to Vp = 32 ceil(V / 32) - once per flow pass (the scores are shared by every encoder
//                         layer and both nets)
//   xt_split_kernel       x [n, V, D] fp32 -> x^T (hi, lo) fp16 [n][D x Vp in fragment order] - once per (layer, net)
//   attend_h3p_kernel     att[n, q, h, :] = sum_m S_h[q, m] x[n, m, :]: both operands arrive as 16-byte fp16 vectors, k contiguous -
//                         no VALU work between the loads and the MFMAs
__global__ void split_scores_kernel(const float* __restrict__ s, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int64_t rows,
                                    int V, int Vp) {
  // rows = matrices x Vp (padded query rows included: zeros)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Vp) return;
  const int64_t r = i / Vp;
  const int k = (int)(i - r * Vp);
  const int64_t mat = r / Vp;
  const int q = (int)(r - mat * Vp);
  const float v = (k < V && q < V) ? s[(mat * V + q) * V + k] * AH_SSCALE : 0.f;
  const _Float16 h = (_Float16)v;
  const int64_t o = mat * Vp * Vp + frag_off(Vp / 16, q, k);
  hi[o] = h;
  lo[o] = (_Float16)(v - (float)h);
}

__global__ void __launch_bounds__(256) xt_split_kernel(const float* __restrict__ x, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                        int V, int Vp, int D) {
  // block (row n, key tile of 32, feature tile of 32): a 32 x 32 transpose through the LDS
  __shared__ float t[32][33];
  const int64_t n = blockIdx.x;
  const int m0 = blockIdx.y * 32, d0 = blockIdx.z * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per sweep
#pragma unroll
  for (int r = ty; r < 32; r += 8) t[r][tx] = (m0 + r < V && d0 + tx < D) ? x[(n * V + m0 + r) * D + d0 + tx] : 0.f;
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    if (d0 + r >= D) continue;
    const float v = t[tx][r];   // key m0 + tx of feature d0 + r
    const _Float16 h = (_Float16)v;
    const int64_t o = n * D * (int64_t)Vp + frag_off((D + 15) / 16, d0 + r, m0 + tx);
    hi[o] = h;
    lo[o] = (_Float16)(v - (float)h);
  }
}

__global__ void __launch_bounds__(256) attend_h3p_kernel(const _Float16* __restrict__ s_hi, const _Float16* __restrict__ s_lo,
                                                          const _Float16* __restrict__ xt_hi, const _Float16* __restrict__ xt_lo,
                                                          float* __restrict__ att, int64_t n_cond, int H, int V, int Vp, int D) {
  extern __shared__ __attribute__((aligned(16))) char lh_lds[];
  _Float16 (*tile)[2][2][LH_BM][LH_ROW] = (_Float16 (*)[2][2][LH_BM][LH_ROW])lh_lds;   // [buffer][S | x^T][hi | lo][row][k]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (V + LH_BM - 1) / LH_BM;
  int64_t blk = blockIdx.x;
  const int tm = (int)(blk % tiles_m); blk /= tiles_m;
  const int h = (int)(blk % H);
  const int64_t n = blk / H;
  const int64_t c = n % n_cond;
  const int q0 = tm * LH_BM;
  const _Float16* Sh = s_hi + ((c * H + h) * Vp) * (int64_t)Vp;
  const _Float16* Sl = s_lo + ((c * H + h) * Vp) * (int64_t)Vp;
  const _Float16* Xh = xt_hi + n * D * (int64_t)Vp;
  const _Float16* Xl = xt_lo + n * D * (int64_t)Vp;
  lin_f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (lin_f4){0.f, 0.f, 0.f, 0.f};
  // a k-step: four 128-row x 32-k fp16 tiles = 4 x 512 sixteen-byte pieces, 8 per thread (t = 0, 1: S hi; 2, 3: S lo; 4, 5: x^T hi; 6, 7: lo)
  lin_h8 nx[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = threadIdx.x + 256 * (t & 1), r = i >> 2, kq = 8 * (i & 3);
      lin_h8 v = (lin_h8){0, 0, 0, 0, 0, 0, 0, 0};
      if (t < 4) {
        if (q0 + r < V) v = *(const lin_h8*)((t < 2 ? Sh : Sl) + frag_off(Vp / 16, q0 + r, k0 + kq));
      } else if (r < D) {
        v = *(const lin_h8*)((t < 6 ? Xh : Xl) + frag_off((D + 15) / 16, r, k0 + kq));
      }
      nx[t] = v;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = threadIdx.x + 256 * (t & 1), r = i >> 2, kq = 8 * (i & 3);
      *(lin_h8*)&tile[buf][t >> 2][(t >> 1) & 1][r][kq] = nx[t];
    }
  };
  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < Vp; k0 += 32) {
    const bool more = k0 + 32 < Vp;
    if (more) fetch(k0 + 32);
    lin_h8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 64 * wm + 16 * i + i16;
      ah[i] = *(const lin_h8*)&tile[buf][0][0][r][8 * g];
      al[i] = *(const lin_h8*)&tile[buf][0][1][r][8 * g];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 64 * wn + 16 * j + i16;
      bh[j] = *(const lin_h8*)&tile[buf][1][0][r][8 * g];
      bl[j] = *(const lin_h8*)&tile[buf][1][1][r][8 * g];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
      }
    if (more) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int dd = 64 * wn + 16 * j + i16;
    if (dd >= D) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + 64 * wm + 16 * i + 4 * g + r;
        if (q < V) att[((n * V + q) * H + h) * (int64_t)D + dd] = acc[i][j][r] * (1.0f / AH_SSCALE);
      }
  }
}

// Mixing AND the folded projection in one launch (r06): per (row n, 64 J queries), for every head
//   xm_h^T[d][q] = sum_m x^T[d][m] S_h[q][m]          (A = x^T rows, B = score rows: the transposed formulation of the fused kernels)
//   y^T[c][q]   += sum_d Wc_h[c][d] xm_h[q][d]        (A = Wc_h rows; B = xm_h straight from the accumulators: lane (q, g) holds
//                                                     features 16 t + 4 g + r of feature tile t, i.e. after the fp16 split the B operand
//                                                     of k-step ks = tiles 2 ks, 2 ks + 1 in the element order the pack gave Wc)
// so the [M, 768] mixing result never exists in memory (attend_h3p_kernel + the 768 -> 128 GEMM wrote and read 300 MB per call at
// 192 x 256).  A wave owns 16 J queries x all 128 features (J = 2: 128 queries per workgroup, the one instantiated; J = 4: 256 - every
// x^T / Wc fragment read from the LDS then feeds 12 MFMAs instead of 6, at one wave per SIMD).  Every operand tile arrives by LDS-DMA (global_load_lds, 1 KiB
// per instruction, fragment-order tiles: frag_off) into a ring of NSLOT stage buffers [S hi | S lo | x^T hi | x^T lo]; NSLOT - 1 stages
// are in flight while one computes; a counted vmcnt + barrier per step.
// Registers: hipcc's own choice for J = 2 was 184 VGPRs + 96 AGPRs = one wave per SIMD, i.e. the two workgroups a CU's LDS holds ran
// one after the other (256 atoms x 256 rows: 148 us per call); held to two waves per SIMD it takes 184 VGPRs, no AGPRs, no scratch.
template <int J, int NSLOT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(J == 2 ? 2 : 1, J == 2 ? 2 : 1))) attend_fold_h3_kernel(const _Float16* __restrict__ s_hi, const _Float16* __restrict__ s_lo,
                                                              const _Float16* __restrict__ xt_hi, const _Float16* __restrict__ xt_lo,
                                                              const _Float16* __restrict__ wc_hi, const _Float16* __restrict__ wc_lo,
                                                              float* __restrict__ y, int64_t n_cond, int H, int V, int Vp,
                                                              float* __restrict__ hres, const float* __restrict__ lnw,
                                                              const float* __restrict__ lnb, float eps, int head_parts,
                                                              int64_t part_stride) {
  // head_parts > 1 (small launches of big molecules: 691 atoms x 16 rows are 96 workgroups): the heads over `head_parts` workgroups
  // per query tile, each writing its partial y to y + part * part_stride (hres == nullptr); parts_ln_kernel finishes the layer
  extern __shared__ __attribute__((aligned(16))) char lh_lds[];
  constexpr int D = 128, QB = 64 * J, NCH_S = QB / 16;
  constexpr int S_ARR = NCH_S * 1024, X_ARR = 8 * 1024, X_OFF = 2 * S_ARR, SLOT = 2 * S_ARR + 2 * X_ARR;
  constexpr int MIX_PER_WAVE = (2 * NCH_S + 16) / 4, FOLD_PER_WAVE = 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int tiles_m = (V + QB - 1) / QB;
  const int hpart = (int)(blockIdx.x % head_parts);
  const int tm = (int)((blockIdx.x / head_parts) % tiles_m);
  const int64_t n = blockIdx.x / head_parts / tiles_m;
  const int64_t c = n % n_cond;
  const int h_count = H / head_parts, h_begin = hpart * h_count;
  y += hpart * part_stride;
  const int q0 = tm * QB;
  const int n1 = Vp / 32;
  const int per_head = n1 + 4;
  const int total = h_count * per_head;
  const _Float16* Xh = xt_hi + n * D * (int64_t)Vp;
  const _Float16* Xl = xt_lo + n * D * (int64_t)Vp;
  // the tiles of global step t = (head, st) into `slot`; st < n1: key step st of the mixing; st >= n1: k-step st - n1 of the folded GEMM
  // (Wc hi / lo in the x^T areas).  Piece i = wave + 4 k of the stage's list is this wave's k-th: uniform per wave.
  auto issue = [&](int t, int slot) {
    const int hr = t / per_head, st = t - hr * per_head;
    const int h = h_begin + hr;
    char* base = lh_lds + slot * SLOT;
    if (st < n1) {
#pragma unroll
      for (int k = 0; k < MIX_PER_WAVE; ++k) {
        const int i = wave + 4 * k;
        const _Float16* src;
        char* dst;
        if (i < 2 * NCH_S) {
          const int ch = i < NCH_S ? i : i - NCH_S;
          int rt = q0 / 16 + ch;
          rt = rt < Vp / 16 ? rt : Vp / 16 - 1;   // query tiles past the padded molecule: any tile (their results are dropped)
          src = (i < NCH_S ? s_hi : s_lo) + (c * H + h) * (int64_t)Vp * Vp + ((int64_t)st * (Vp / 16) + rt) * 512 + lane * 8;
          dst = base + i * 1024;
        } else {
          const int ch = (i - 2 * NCH_S) & 7;
          src = (i - 2 * NCH_S < 8 ? Xh : Xl) + ((int64_t)st * (D / 16) + ch) * 512 + lane * 8;
          dst = base + X_OFF + (i - 2 * NCH_S) * 1024;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst,
                                         16, 0, 0);
      }
    } else {
      const int ks = st - n1;
#pragma unroll
      for (int k = 0; k < FOLD_PER_WAVE; ++k) {
        const int i = wave + 4 * k;   // 0 .. 7: Wc hi tile i; 8 .. 15: Wc lo tile i - 8
        const _Float16* src = (i < 8 ? wc_hi : wc_lo) + (int64_t)h * D * D + ((int64_t)ks * (D / 16) + (i & 7)) * 512 + lane * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + X_OFF + i * 1024), 16, 0, 0);
      }
    }
  };
  // this wave's LDS-DMA instructions of global step t (0 past the end)
  auto pieces = [&](int t) -> int {
    if (t >= total) return 0;
    const int h = t / per_head;
    return t - h * per_head < n1 ? MIX_PER_WAVE : FOLD_PER_WAVE;
  };
  // stage `t` has landed for every wave: at most `younger` of this wave's LDS-DMAs - the stages behind it - may still be in flight
  // (hipcc does not count LDS-DMA; s_waitcnt takes an immediate)
  auto landed = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
      case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __syncthreads();
  };
  auto younger_than = [&](int t) -> int {   // pieces of the stages t + 1 .. t + NSLOT - 2 (issued, not needed yet)
    int sum = 0;
#pragma unroll
    for (int k = 1; k <= NSLOT - 2; ++k) sum += pieces(t + k);
    return sum;
  };
  auto frag = [&](int slot, int off, int tile) -> lin_h8 { return *(const lin_h8*)(lh_lds + slot * SLOT + off + tile * 1024 + lane * 16); };
  lin_f4 acc_y[8][J];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int j = 0; j < J; ++j) acc_y[t][j] = (lin_f4){0.f, 0.f, 0.f, 0.f};
  // a wave whose queries all lie past the molecule (the last query tile of 200 atoms: 224 .. 255) moves its share of the stages and
  // meets the barriers, but leaves the matrix pipe to the other workgroup of its CU
  const bool active = q0 + 16 * J * wave < V;
#pragma unroll
  for (int k = 0; k < NSLOT - 1; ++k)
    if (k < total) issue(k, k);
  landed(younger_than(0));
  int step = 0, slot = 0;
  auto next_slot = [&](int s_) { return s_ + 1 == NSLOT ? 0 : s_ + 1; };
  int fill = NSLOT - 1;   // the slot the next issue goes to
  for (int h = 0; h < h_count; ++h) {
    lin_f4 xm[8][J];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int j = 0; j < J; ++j) xm[t][j] = (lin_f4){0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < n1; ++st, ++step) {
      if (step + NSLOT - 1 < total) issue(step + NSLOT - 1, fill);
      fill = next_slot(fill);
      lin_h8 sh[J], sl[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        sh[j] = frag(slot, 0, J * wave + j);
        sl[j] = frag(slot, S_ARR, J * wave + j);
      }
      lin_h8 xh = frag(slot, X_OFF, 0), xl = frag(slot, X_OFF + X_ARR, 0);
      if (active)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        lin_h8 nh = xh, nl = xl;
        if (t < 7) {   // the next feature tile's fragments are on their way while this one's MFMAs issue
          nh = frag(slot, X_OFF, t + 1);
          nl = frag(slot, X_OFF + X_ARR, t + 1);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
          xm[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, sh[j], xm[t][j], 0, 0, 0);
          xm[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, sl[j], xm[t][j], 0, 0, 0);
          xm[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, sh[j], xm[t][j], 0, 0, 0);
        }
        xh = nh;
        xl = nl;
      }
      landed(younger_than(step + 1));
      slot = next_slot(slot);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks, ++step) {
      if (step + NSLOT - 1 < total) issue(step + NSLOT - 1, fill);
      fill = next_slot(fill);
      // xm (x 2^10 from the scores' scale) -> the split B operand of k-step ks, in the accumulators' own element order
      lin_h8 bh[J], bl[J];
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = xm[2 * ks + e / 4][j][e % 4] * (1.0f / AH_SSCALE);
          const _Float16 hh = (_Float16)v;
          bh[j][e] = hh;
          bl[j][e] = (_Float16)(v - (float)hh);
        }
      lin_h8 wh = frag(slot, X_OFF, 0), wl = frag(slot, X_OFF + X_ARR, 0);
      if (active)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        lin_h8 nh = wh, nl = wl;
        if (t < 7) {
          nh = frag(slot, X_OFF, t + 1);
          nl = frag(slot, X_OFF + X_ARR, t + 1);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
          acc_y[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[j], acc_y[t][j], 0, 0, 0);
          acc_y[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl[j], acc_y[t][j], 0, 0, 0);
          acc_y[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh[j], acc_y[t][j], 0, 0, 0);
        }
        wh = nh;
        wl = nl;
      }
      landed(younger_than(step + 1));
      slot = next_slot(slot);
    }
  }
  // y^T tiles: lane (q = i16 of query tile j, g) holds output columns 16 t + 4 g + r
  if (hres) {
    // residual + LayerNorm 1 here as well (custom_transformer_block.py:58-62): a wave holds all 128 features of its queries -
    // per query 32 values in this lane, the rest in the three lanes i16 + 16 g'.  h is read and written in place: the mixing of
    // the other workgroups of this row reads the x^T copy, not h.
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int q = q0 + 16 * J * wave + 16 * j + i16;
      const bool ok = q < V;
      float* row = hres + (n * V + (ok ? q : 0)) * (int64_t)D;
      lin_f4 v[8];
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const lin_f4 r = ok ? *(const lin_f4*)(row + 16 * t + 4 * g) : (lin_f4){0.f, 0.f, 0.f, 0.f};
        v[t] = r + acc_y[t][j] * (1.0f / LH_WSCALE);
        sum += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.f / 128.f);
      float var = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        v[t] = v[t] - mean;
        var += (v[t][0] * v[t][0] + v[t][1] * v[t][1]) + (v[t][2] * v[t][2] + v[t][3] * v[t][3]);
      }
      var += __shfl_xor(var, 16);
      var += __shfl_xor(var, 32);
      const float rstd = 1.0f / sqrtf(var * (1.f / 128.f) + eps);
      if (ok) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          // (the raw parameter vector is 4-byte aligned: element loads)
          const float* pw = lnw + 16 * t + 4 * g;
          const float* pb = lnb + 16 * t + 4 * g;
          const lin_f4 w = (lin_f4){pw[0], pw[1], pw[2], pw[3]}, b = (lin_f4){pb[0], pb[1], pb[2], pb[3]};
          *(lin_f4*)(row + 16 * t + 4 * g) = v[t] * rstd * w + b;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int q = q0 + 16 * J * wave + 16 * j + i16;
    if (q >= V) continue;
    float* row = y + (n * V + q) * (int64_t)D;
#pragma unroll
    for (int t = 0; t < 8; ++t) *(lin_f4*)(row + 16 * t + 4 * g) = acc_y[t][j] * (1.0f / LH_WSCALE);
  }
}

// The same product for large molecules (r05: the V x V tile of attend_kernel stops at ~200 atoms): per (row n, head h) the GEMM
//   att[q, d] = sum_m scores[q, m] * vals[m, d]      M = K = V, N = D
// tiled like linear_kernel - 128 x 64 output tile per workgroup, k in steps of 16 through the LDS, v_mfma_f32_16x16x4_f32 (exact
// fp32 products, fp32 accumulate).  1-D grid over (n, h, query tile, feature tile).
__global__ void __launch_bounds__(256) attend_mfma_kernel(const float* __restrict__ scores, const float* __restrict__ vals,
                                                           float* __restrict__ att, int64_t n_cond, int H, int V, int D) {
  __shared__ float ss[16][LIN_BM + 4];   // scores tile, k-major: ss[k][q]
  __shared__ float vs[16][LIN_BN + 4];   // values tile:          vs[k][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int tiles_n = (D + LIN_BN - 1) / LIN_BN, tiles_m = (V + LIN_BM - 1) / LIN_BM;
  int64_t blk = blockIdx.x;
  const int tn = (int)(blk % tiles_n); blk /= tiles_n;
  const int tm = (int)(blk % tiles_m); blk /= tiles_m;
  const int h = (int)(blk % H);
  const int64_t n = blk / H;
  const int64_t c = n % n_cond;
  const int q0 = tm * LIN_BM, d0 = tn * LIN_BN;
  const float* S = scores + ((c * H + h) * V) * (int64_t)V;                 // [V, V] row-major
  const float* Vv = vals + (n * V * (int64_t)H + h) * D;                    // row m at + m * H * D
  const int64_t vstride = (int64_t)H * D;
  lin_f4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (lin_f4){0.f, 0.f, 0.f, 0.f};
  // (the next K = 16 slice is fetched into registers before the current slice's MFMAs, as in linear_kernel)
  float ns[8], nv[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {   // scores: 16 consecutive k of a query row are contiguous
      const int i = threadIdx.x + 256 * t, r = i / 16, k = i % 16;
      ns[t] = (q0 + r < V && k0 + k < V) ? S[(int64_t)(q0 + r) * V + k0 + k] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {   // values: the feature index is contiguous
      const int i = threadIdx.x + 256 * t, k = i / LIN_BN, dd = i % LIN_BN;
      nv[t] = (k0 + k < V && d0 + dd < D) ? Vv[(int64_t)(k0 + k) * vstride + d0 + dd] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < V; k0 += 16) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = threadIdx.x + 256 * t;
      ss[i % 16][i / 16] = ns[t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = threadIdx.x + 256 * t;
      vs[i / LIN_BN][i % LIN_BN] = nv[t];
    }
    __syncthreads();
    if (k0 + 16 < V) fetch(k0 + 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = ss[4 * ks + g][32 * wave + 16 * i + i16];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = vs[4 * ks + g][16 * j + i16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D layout: lane (i16, g) holds rows 4 g + r (r = 0..3) of column i16
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int dd = d0 + 16 * j + i16;
    if (dd >= D) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + 32 * wave + 16 * i + 4 * g + r;
        if (q < V) att[((n * V + q) * H + h) * (int64_t)D + dd] = acc[i][j][r];
      }
  }
}

// dense softmax attention for one (row, head): qkv [n, V, 3*d]; out [n, V, d]
// (torch.nn.MultiheadAttention with key padding mask)
__global__ void sdpa_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ masked, int64_t n_cond,
                            float* __restrict__ out, int V, int d, int n_head) {
  extern __shared__ float sm[];
  const int dh = d / n_head;
  float* q = sm;               // [V*dh]
  float* k = q + V * dh;       // [V*dh]
  float* v = k + V * dh;       // [V*dh]
  float* p = v + V * dh;       // [V*V]
  const int64_t n = blockIdx.x;
  const int h = blockIdx.y;
  const int64_t c = n % n_cond;
  for (int i = threadIdx.x; i < V * dh; i += blockDim.x) {
    const int a = i / dh, j = i % dh;
    const float* row = qkv + (n * V + a) * 3 * (int64_t)d;
    q[i] = row[h * dh + j] / sqrtf((float)dh);
    k[i] = row[d + h * dh + j];
    v[i] = row[2 * d + h * dh + j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < V * V; i += blockDim.x) {
    const int a = i / V, m = i % V;
    float acc = 0.f;
    for (int j = 0; j < dh; ++j) acc = fmaf(q[a * dh + j], k[m * dh + j], acc);
    p[i] = masked[c * V + m] ? -INFINITY : acc;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < V; a += blockDim.x) {
    float mx = -INFINITY;
    for (int m = 0; m < V; ++m) mx = fmaxf(mx, p[a * V + m]);
    float sum = 0.f;
    for (int m = 0; m < V; ++m) { float e = expf(p[a * V + m] - mx); p[a * V + m] = e; sum += e; }
    for (int m = 0; m < V; ++m) p[a * V + m] /= sum;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < V * dh; i += blockDim.x) {
    const int a = i / dh, j = i % dh;
    float acc = 0.f;
    for (int m = 0; m < V; ++m) acc = fmaf(p[a * V + m], v[m * dh + j], acc);
    out[(n * V + a) * (int64_t)d + h * dh + j] = acc;
  }
}

// The same attention without a V x V tile in the LDS (any V; r05): one thread per query row, keys and values read from the
// qkv buffer (every thread of a workgroup reads the same key: broadcast loads), scores recomputed in each of the three passes
// (max, sum, normalised mixing) instead of cached.  Same operations in the same order as sdpa_kernel: bit-identical.
template <int DH>   // head width rounded up (registers, not scratch: every loop over the head is unrolled)
__global__ void sdpa_rows_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ masked, int64_t n_cond,
                                 float* __restrict__ out, int V, int d, int n_head) {
  const int dh = d / n_head;
  const int64_t n = blockIdx.x;
  const int h = blockIdx.y;
  const int a = blockIdx.z * blockDim.x + threadIdx.x;
  if (a >= V) return;
  const int64_t c = n % n_cond;
  const float* base = qkv + n * V * 3 * (int64_t)d;
  const uint8_t* mk = masked + c * V;
  float q[DH];
#pragma unroll
  for (int j = 0; j < DH; ++j) q[j] = j < dh ? base[(int64_t)a * 3 * d + h * dh + j] / sqrtf((float)dh) : 0.f;
  auto score = [&](int m) {
    const float* k = base + (int64_t)m * 3 * d + d + h * dh;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < DH; ++j)
      if (j < dh) acc = fmaf(q[j], k[j], acc);
    return mk[m] ? -INFINITY : acc;
  };
  float mx = -INFINITY;
  for (int m = 0; m < V; ++m) mx = fmaxf(mx, score(m));
  float sum = 0.f;
  for (int m = 0; m < V; ++m) sum += expf(score(m) - mx);
  float acc[DH];
#pragma unroll
  for (int j = 0; j < DH; ++j) acc[j] = 0.f;
  for (int m = 0; m < V; ++m) {
    const float pm = expf(score(m) - mx) / sum;
    const float* v = base + (int64_t)m * 3 * d + 2 * d + h * dh;
#pragma unroll
    for (int j = 0; j < DH; ++j)
      if (j < dh) acc[j] = fmaf(pm, v[j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < DH; ++j)
    if (j < dh) out[(n * V + a) * (int64_t)d + h * dh + j] = acc[j];
}

// The same attention on the fp32 matrix pipe (r06; head width 16 - the configured dense model: 128 / 8), flash style: a workgroup
// stages K and V of one (row, head) in the LDS once, each of its waves walks the keys in blocks of 16 for a tile of 16 queries,
//   S^T[key][q]  = sum_j K[key][j] Q[q][j]            4 x v_mfma_f32_16x16x4_f32 (exact fp32 products)
//   online softmax over the keys per query (running maximum and sum; the query is the lane's column: l % 16)
//   O^T[d][q]   += sum_key V[key][d] P[q][key]        4 x MFMA, B operand = the lane's four probabilities as they stand:
// lane (q, g) holds keys 4 g + r of the block after the first product, so k-step r of the second takes keys {4 g + r} - the sums
// over head features and over keys do not care about the order, which spares every transpose.  sdpa_rows_kernel did this with one
// thread per query and scalar FMAs: 855 us per call at 256 atoms x 256 rows, 76 % of a dense per-op pass.
// LDS: K and V in fragment order (lane l of block b reads 16 bytes at (64 b + l) 16), the key mask as 0 / -inf.
typedef float sd_f4 __attribute__((ext_vector_type(4)));
// max / sum over the four lanes that share (lane & 15) without the LDS: v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16- and
// 32-lane blocks between two registers (the fused kernels' h3_quad_max / h3_quad_sum, csrc/tw_netblock_h3.hip)
__device__ __forceinline__ void sd_swap16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void sd_swap32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float sd_quad_max(float v) {
  float a = v, b = v;
  sd_swap16(a, b);
  a = b = fmaxf(a, b);
  sd_swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float sd_quad_sum(float v) {
  float a = v, b = v;
  sd_swap16(a, b);
  a = b = a + b;
  sd_swap32(a, b);
  return a + b;
}
__global__ void __launch_bounds__(256) sdpa_mfma_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ masked, int64_t n_cond,
                                                         float* __restrict__ out, int V, int d, int n_head, int q_tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float sd_lds[];
  const int V16 = (V + 15) / 16;
  sd_f4* kl = (sd_f4*)sd_lds;                 // [V16][64]: lane (key % 16, g): K[key][4 g .. 4 g + 3]
  sd_f4* vl = kl + V16 * 64;                  // [V16][64]: lane (d, g): V[16 b + 4 g + r][d], r = 0 .. 3
  float* mk = (float*)(vl + V16 * 64);        // [16 V16]
  const int64_t n = blockIdx.x;
  const int h = blockIdx.y;
  const int64_t c = n % n_cond;
  const float* base = qkv + n * V * 3 * (int64_t)d + h * 16;
  for (int i = threadIdx.x; i < V16 * 64; i += 256) {
    const int key = i >> 2, j4 = i & 3;   // 16 bytes of one key row
    sd_f4 kv = (sd_f4){0.f, 0.f, 0.f, 0.f}, vv = kv;
    if (key < V) {
      kv = *(const sd_f4*)(base + (int64_t)key * 3 * d + d + 4 * j4);
      vv = *(const sd_f4*)(base + (int64_t)key * 3 * d + 2 * d + 4 * j4);
    }
    const int b = key >> 4, kk = key & 15;
    kl[b * 64 + j4 * 16 + kk] = kv;
    float* vt = (float*)(vl + b * 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) vt[((kk >> 2) * 16 + 4 * j4 + e) * 4 + (kk & 3)] = vv[e];
  }
  for (int i = threadIdx.x; i < V16 * 16; i += 256) mk[i] = (i >= V || masked[c * V + i]) ? -INFINITY : 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q16 = lane & 15, g = lane >> 4;
  const int QT = V16;
  for (int it = 0; it < q_tiles_per_wave; ++it) {
    const int qt = (blockIdx.z * q_tiles_per_wave + it) * 4 + wave;
    if (qt >= QT) break;
    const int q = 16 * qt + q16;
    const sd_f4 qb = *(const sd_f4*)(base + (int64_t)(q < V ? q : V - 1) * 3 * d + 4 * g) * 0.25f;   // 1 / sqrt(16)
    float m_run = -INFINITY, l_run = 0.f;
    sd_f4 o = (sd_f4){0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < V16; ++b) {
      const sd_f4 kf = kl[b * 64 + lane];
      const sd_f4 vf = vl[b * 64 + lane];
      const sd_f4 mb = *(const sd_f4*)(mk + 16 * b + 4 * g);
      sd_f4 st = mb;
#pragma unroll
      for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qb[s], st, 0, 0, 0);
      const float mx = sd_quad_max(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
      const float m_new = fmaxf(m_run, mx);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;   // (a block of masked keys before the first real one)
      // e^(s - m) = 2^((s - m) log2 e): the difference first (exact near the maximum, where the weight is), one rounding in the
      // product, v_exp_f32 (1 ulp) - relative error <= |s - m| 2^-24, i.e. largest where the probability is smallest
      const float L2E = 1.4426950408889634f;
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * L2E);
      sd_f4 pr;
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f((st[r] - m_safe) * L2E);
      const float ps = sd_quad_sum((pr[0] + pr[1]) + (pr[2] + pr[3]));
      l_run = l_run * alpha + ps;
      o = o * alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], pr[r], o, 0, 0, 0);
      m_run = m_new;
    }
    if (q < V) *(sd_f4*)(out + (n * V + q) * (int64_t)d + h * 16 + 4 * g) = o * (1.0f / l_run);
  }
}

// h = LayerNorm(h + sum_p parts[p]) for D = 128: behind a launch that left partial sums (attend_fold_h3_kernel with head_parts > 1)
__global__ void __launch_bounds__(256) parts_ln_kernel(float* __restrict__ h, const float* __restrict__ parts, int n_parts, int64_t part_stride,
                                                        const float* __restrict__ w, const float* __restrict__ b, float eps, int64_t tokens) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= tokens) return;
  const int lane = threadIdx.x & 63;
  float* row = h + t * 128;
  float v0 = row[lane], v1 = row[64 + lane];
  for (int q = 0; q < n_parts; ++q) {
    const float* pr = parts + q * part_stride + t * 128;
    v0 += pr[lane];
    v1 += pr[64 + lane];
  }
  const float mean = wave_sum(v0 + v1) * (1.f / 128.f);
  const float d0 = v0 - mean, d1 = v1 - mean;
  const float rstd = 1.0f / sqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / 128.f) + eps);
  row[lane] = d0 * rstd * w[lane] + b[lane];
  row[64 + lane] = d1 * rstd * w[64 + lane] + b[64 + lane];
}

// h = LayerNorm(h + delta) (custom_attention_encoder.py:109-114); one wave per token
__global__ void add_ln_kernel(float* __restrict__ h, const float* __restrict__ delta, const float* __restrict__ w,
                              const float* __restrict__ b, float eps, int D, int64_t tokens) {
  const int64_t tok = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  if (tok >= tokens) return;
  const int lane = threadIdx.x % 64;
  float* row = h + tok * D;
  const float* drow = delta + tok * D;
  float sum = 0.f;
  for (int i = lane; i < D; i += 64) { float v = row[i] + drow[i]; row[i] = v; sum += v; }
  sum = wave_sum(sum);
  const float mean = sum / D;
  float var = 0.f;
  for (int i = lane; i < D; i += 64) { float dv = row[i] - mean; var += dv * dv; }
  var = wave_sum(var) / D;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int i = lane; i < D; i += 64) row[i] = (row[i] - mean) * rstd * w[i] + b[i];
}

struct SimpleWs {
  float *u, *h0, *h, *vals, *att, *ff, *tmp, *s_out, *t_out, *scores;
  _Float16 *s_hi, *s_lo, *xt_hi, *xt_lo;   // TW_PATH_SIMPLE_H3, folded attention: pre-split scores / transposed layer input
  int64_t bytes;
};

static SimpleWs simple_ws(const tw_flow_desc& d, int64_t n_rows, int V, void* base) {
  SimpleWs w;
  const int64_t M = n_rows * V;
  const int d_in = d.d_emb + 9 + (d.variant == 1 ? d.d_rff : 0);
  const int64_t wide = d.variant == 0 ? (int64_t)d.n_heads * d.d_model : 3LL * d.d_model;
  char* p = (char*)base;
  auto take = [&](int64_t floats) {
    float* r = (float*)p;
    p += ((floats * 4 + 255) / 256) * 256;
    return r;
  };
  w.u = take(M * d_in);
  w.h0 = take(M * d.d_hidden);
  w.h = take(M * d.d_model);
  w.vals = take(M * wide);
  w.att = take(M * wide);
  w.ff = take(M * d.d_ff);
  w.tmp = take(M * d.d_model);
  w.s_out = take(M * 3);
  w.t_out = take(M * 3);
  w.scores = take(n_rows * (int64_t)d.n_heads * V * V);
  w.s_hi = w.s_lo = w.xt_hi = w.xt_lo = nullptr;
  if (d.variant == 0 && d.d_model == 128) {   // (sized whatever path the call takes: one workspace serves them all)
    const int64_t Vp = (V + 31) / 32 * 32;
    w.s_hi = (_Float16*)take((n_rows * (int64_t)d.n_heads * Vp * Vp + 1) / 2);
    w.s_lo = (_Float16*)take((n_rows * (int64_t)d.n_heads * Vp * Vp + 1) / 2);
    w.xt_hi = (_Float16*)take((n_rows * (int64_t)d.d_model * Vp + 1) / 2);
    w.xt_lo = (_Float16*)take((n_rows * (int64_t)d.d_model * Vp + 1) / 2);
  }
  w.bytes = p - (char*)base;
  return w;
}

// The two nets of a coupling layer are independent until the coupling step: the second runs beside the first on a side stream, with
// activation buffers of its own behind the first net's workspace (scores / split scores are shared, read-only).  Small launches do not
// fill the chip with one net (691 atoms x 16 rows: 9.2 -> 5.2 ms per pass together with the split launches); large ones still gain
// the tails and the small kernels of one net under the big ones of the other (691 x 32: 12.4 -> 9.6 ms, 192 x 256: 10.9 -> 10.2,
// 256 x 256: 14.0 -> 13.7).  Not beyond 32 GiB per net (the workspace doubles).
static bool simple_two_streams(int64_t one_net_bytes) { return one_net_bytes <= (int64_t)32 << 30; }
// the second net's activation buffers: a second copy of everything up to t_out
static SimpleWs simple_ws_second(const tw_flow_desc& d, int64_t n_rows, int V, const SimpleWs& first, void* base) {
  SimpleWs w = simple_ws(d, n_rows, V, base);
  if (!(d.variant == 0 && d.cheb_order > 0)) w.scores = first.scores;   // (chebyshev_kernel: every net and layer computes its own scores)
  w.s_hi = first.s_hi;
  w.s_lo = first.s_lo;
  w.s_out = first.s_out;
  w.t_out = first.t_out;
  return w;
}

int64_t simple_workspace_bytes(const tw_flow_desc& d, int64_t n_rows, int n_atoms) {
  const int64_t one = simple_ws(d, n_rows, n_atoms, nullptr).bytes;
  return simple_two_streams(one) ? 2 * one : one;
}

// one net-block on the simple path; out [M,3].  dump (optional): activations after each stage.
static int netblock_simple(const FlowArgs& a, const RawLayout& L, const SimpleWs& w, int c, int net,
                           const float* z_other, float* out, float* dump) {
  const tw_flow_desc& d = *a.desc;
  const int V = a.n_atoms;
  const int64_t M = a.n_rows * V;
  const bool sp = a.simple_h3 != 0;   // TW_PATH_SIMPLE_H3: the linears on split-fp16 MFMAs
  const float* nb = a.raw + net_base(L, c, net);
  hipStream_t s = a.stream;
  const float* rff = d.variant == 1 ? a.raw + L.chain + (int64_t)c * L.coupling_size + L.rff : nullptr;
  hipLaunchKernelGGL(build_input_kernel, dim3((unsigned)M), dim3(64), 0, s, a.raw + L.emb, a.atom_types, a.x_coords,
                     a.x_velocs, z_other, rff, d.d_rff, a.n_cond, V, d.d_emb, L.d_in, w.u, M);
  TW_LAUNCH_CHECK();
  int rc;
  // TW_PATH_SIMPLE_H3 with the split-fp16 stream at hand: each of the two MLPs as ONE launch of the fused kernels' generated
  // statement on the flat token list, its hidden layer on the chip (bit 28: as two GEMMs each - A/B, tests)
  const bool io_tokens = sp && a.packed && h3_io_tokens_supported(d) && L.d_in <= 64 && !(g_debug_flags & (16777216 | 268435456));
  if (io_tokens) {
    if ((rc = h3_io_tokens(d, a.packed, c, net, false, w.u, w.h, L.d_in, M, s))) return rc;
  } else {
  if ((rc = launch_linear(w.u, nb + L.net.in0_w, nb + L.net.in0_b, w.h0, M, d.d_hidden, L.d_in, ACT_SILU, s, sp))) return rc;
  if ((rc = launch_linear(w.h0, nb + L.net.in2_w, nb + L.net.in2_b, w.h, M, d.d_model, d.d_hidden, ACT_NONE, s, sp))) return rc;
  }
  const int64_t act_sz = M * d.d_model;
  if (dump) TW_HIP_CHECK(hipMemcpyAsync(dump, w.h, act_sz * 4, hipMemcpyDeviceToDevice, s));
  for (int l = 0; l < d.n_layers; ++l) {
    const float* lb = nb + L.net.layers + (int64_t)l * L.layer.size;
    if (d.variant == 0) {
      const int HD = d.n_heads * d.d_model;
      if (d.cheb_order > 0) {
        // chebyshev_kernel: every attention layer owns its coefficients and the reference's score cache is keyed
        // by the (per-layer) basis function, so the scores are recomputed for each layer (kernel_attention.py:329-333)
        if ((rc = launch_scores(a.x_coords, a.masked, a.raw + L.lengthscales + (a.reverse ? d.n_heads : 0), d.n_heads,
                                a.n_cond, a.n_atoms, d.normalise, a.n_atoms > 25, w.scores, s, lb + L.layer.cheb,
                                d.cheb_order, d.cheb_force_zero)))
          return rc;
      }
      // TW_PATH_SIMPLE_H3 with its pack at hand (tw_flow_pack_simple_h3: the split-fp16 stream, then Wc of every (coupling, net,
      // layer)): the mixing runs on the layer input itself and ONE 768 -> 128 GEMM follows it - no value projection, no [M, 768]
      // round trip for it
      const float* wc = (sp && a.packed && V > 64 && h3_ffn_tokens_supported(d) && !(g_debug_flags & 16777216))
          ? (const float*)((const char*)a.packed + (h3_packed_bytes(d, false) + 255) / 256 * 256) +
                (((int64_t)c * 2 + net) * d.n_layers + l) * (int64_t)d.d_model * HD
          : nullptr;
      if (!wc && (rc = launch_linear(w.h, lb + L.layer.wv, nullptr, w.vals, M, HD, d.d_model, ACT_NONE, s, sp))) return rc;
      if (wc && w.s_hi && d.d_model == LH_BN && d.cheb_order == 0) {
        // folded form, operands prepared once: x^T of this layer's input here, the scores' split in simple_scores()
        const int Vp = (V + 31) / 32 * 32;
        hipLaunchKernelGGL(xt_split_kernel, dim3((unsigned)a.n_rows, (unsigned)(Vp / 32), (unsigned)((d.d_model + 31) / 32)), dim3(256), 0,
                           s, w.h, w.xt_hi, w.xt_lo, V, Vp, d.d_model);
        TW_LAUNCH_CHECK();
        // (one workgroup of the fused form walks all heads of its 128 queries: from 400 of them on.  Below that the heads go over
        // several workgroups per query tile - six up to 128 tiles (691 atoms x 16 rows are 96), two up to 400 (691 x 32: 9.0 -> 8.3 ms
        // per pass) - partial sums through w.att, parts_ln_kernel behind them; bit 26: one workgroup per tile whatever the size;
        // bit 25: the per-head launches + a GEMM + add_ln instead, 132 us per layer at 691 x 16)
        const int64_t fold_wgs = a.n_rows * ((V + 127) / 128);
        int head_parts = 1;
        if (fold_wgs < 400 && !(g_debug_flags & 67108864))
          for (int hp : {fold_wgs < 128 ? 6 : 2, 3, 2})
            if (d.n_heads % hp == 0 && head_parts == 1) head_parts = hp;
        if (!(g_debug_flags & 33554432) && (fold_wgs >= 400 || head_parts > 1 || (g_debug_flags & 67108864))) {
          // ... and the folded 768 -> 128 GEMM inside the mixing launch (bit 25: as its own GEMM behind attend_h3p_kernel; bit 26:
          // inside it whatever the launch size; A/B, tests)
          const int64_t wcf = (int64_t)d.n_coupling * 2 * d.n_layers * d.d_model * HD;   // floats of the fp32 copy in front of the fp16 ones
          const float* fold0 = (const float*)((const char*)a.packed + (h3_packed_bytes(d, false) + 255) / 256 * 256);
          const _Float16* wch = (const _Float16*)(fold0 + wcf) + (((int64_t)c * 2 + net) * d.n_layers + l) * (int64_t)d.d_model * HD;
          // (+ the residual and LayerNorm 1 in its epilogue; bit 27: as the add_ln launch behind it - A/B, tests)
          const bool ln_in = !(g_debug_flags & 134217728) && head_parts == 1;
          // 128 queries per workgroup on two stage buffers, two workgroups per CU.  Measured against it (profiles/r06_attend_fold_occupancy.txt):
          // 256 queries per workgroup (every x^T / Wc fragment read feeds 12 MFMAs instead of 6, but 489 registers = one wave per
          // SIMD) on two or three stage buffers, and 128 queries on four - all slower.
          {
            constexpr int ldsf = 2 * 32 * 1024;
            const int64_t blocks = fold_wgs * head_parts;
            TW_REQUIRE(blocks < (int64_t)1 << 31, "attend: %lld workgroups", (long long)blocks);
            static LdsLimit limf;
            if ((rc = limf.ensure((const void*)attend_fold_h3_kernel<2, 2>, ldsf))) return rc;
            hipLaunchKernelGGL((attend_fold_h3_kernel<2, 2>), dim3((unsigned)blocks), dim3(256), ldsf, s, w.s_hi, w.s_lo, w.xt_hi, w.xt_lo,
                               wch, wch + wcf, head_parts > 1 ? w.att : w.tmp, a.n_cond, d.n_heads, V, Vp, ln_in ? w.h : nullptr,
                               lb + L.layer.n1w, lb + L.layer.n1b, d.ln_eps, head_parts, M * d.d_model);
          }
          TW_LAUNCH_CHECK();
          if (head_parts > 1) {
            hipLaunchKernelGGL(parts_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, w.h, w.att, head_parts, M * d.d_model,
                               lb + L.layer.n1w, lb + L.layer.n1b, d.ln_eps, M);
            TW_LAUNCH_CHECK();
            goto ln1_done;
          }
          if (ln_in) goto ln1_done;
          goto attention_done;
        }
        const int64_t blocks = a.n_rows * d.n_heads * ((V + LH_BM - 1) / LH_BM);
        TW_REQUIRE(blocks < (int64_t)1 << 31, "attend: %lld workgroups", (long long)blocks);
        constexpr int lds = 2 * 2 * 2 * LH_BM * LH_ROW * (int)sizeof(_Float16);
        static LdsLimit limp;
        if ((rc = limp.ensure((const void*)attend_h3p_kernel, lds))) return rc;
        hipLaunchKernelGGL(attend_h3p_kernel, dim3((unsigned)blocks), dim3(256), lds, s, w.s_hi, w.s_lo, w.xt_hi, w.xt_lo, w.att,
                           a.n_cond, d.n_heads, V, Vp, d.d_model);
      } else if (sp && V > 64) {
        // TW_PATH_SIMPLE_H3: the mixing on split-fp16 MFMAs as well (128 x 128 tiles: worth it from ~64 keys on)
        const int64_t blocks = a.n_rows * d.n_heads * ((V + LH_BM - 1) / LH_BM) * ((d.d_model + LH_BN - 1) / LH_BN);
        TW_REQUIRE(blocks < (int64_t)1 << 31, "attend: %lld workgroups", (long long)blocks);
        constexpr int lds = 2 * 2 * 2 * LH_BM * LH_ROW * (int)sizeof(_Float16);
        static LdsLimit lim;
        if ((rc = lim.ensure((const void*)attend_h3_kernel, lds))) return rc;
        hipLaunchKernelGGL(attend_h3_kernel, dim3((unsigned)blocks), dim3(256), lds, s, w.scores, wc ? w.h : w.vals, w.att, a.n_cond,
                           d.n_heads, V, d.d_model, wc ? (int64_t)d.d_model : (int64_t)HD, wc ? (int64_t)0 : (int64_t)d.d_model);
      } else if (V > 64 || (g_debug_flags & 2097152)) {
        // above 64 atoms: the tiled MFMA form (no V x V tile in the LDS: any molecule size; the scalar kernel below took 12 ms
        // per call at 100 atoms x 512 rows - 78 % of a per-op pass, profiles/r05_paired_kernel_stats.csv)
        const int64_t blocks = a.n_rows * d.n_heads * ((V + LIN_BM - 1) / LIN_BM) * ((d.d_model + LIN_BN - 1) / LIN_BN);
        TW_REQUIRE(blocks < (int64_t)1 << 31, "attend: %lld workgroups", (long long)blocks);
        hipLaunchKernelGGL(attend_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w.scores, w.vals, w.att, a.n_cond, d.n_heads,
                           V, d.d_model);
      } else {
        hipLaunchKernelGGL(attend_kernel, dim3((unsigned)a.n_rows, d.n_heads), dim3(128), (size_t)V * V * 4, s, w.scores,
                           w.vals, w.att, a.n_cond, d.n_heads, V, d.d_model);
      }
      TW_LAUNCH_CHECK();
      if ((rc = launch_linear(w.att, wc ? wc : lb + L.layer.wo, nullptr, w.tmp, M, d.d_model, HD, ACT_NONE, s, sp))) return rc;
    attention_done:;
    } else {
      if ((rc = launch_linear(w.h, lb + L.layer.in_w, lb + L.layer.in_b, w.vals, M, 3 * d.d_model, d.d_model, ACT_NONE, s, sp))) return rc;
      const int dh = d.d_model / d.n_heads;
      const size_t sdpa_lds = (size_t)(3 * V * dh + V * V) * 4;
      const int V16 = (V + 15) / 16;
      const size_t mfma_lds = (size_t)V16 * (2 * 1024 + 64);
      if (dh == 16 && V > 64 && mfma_lds <= (size_t)160 * 1024 && !((unsigned)g_debug_flags.load() & 0x80000000u)) {
        // fp32 matrix pipe, K / V of a (row, head) staged once per workgroup (bit 31: the scalar kernels below - A/B, tests); a
        // workgroup's waves take q_tiles_per_wave query tiles each, as many as still leave ~1024 workgroups
        int64_t per_wave = a.n_rows * d.n_heads * (int64_t)V16 / 4 / 1024;
        per_wave = per_wave < 1 ? 1 : per_wave > (V16 + 3) / 4 ? (V16 + 3) / 4 : per_wave;
        const int chunks = (int)((V16 + 4 * per_wave - 1) / (4 * per_wave));
        TW_REQUIRE(d.n_heads <= 65535 && chunks <= 65535, "dense attention: grid %d x %d", d.n_heads, chunks);
        TW_LDS_LIMIT(sdpa_mfma_kernel, mfma_lds, V);
        hipLaunchKernelGGL(sdpa_mfma_kernel, dim3((unsigned)a.n_rows, d.n_heads, (unsigned)chunks), dim3(256), mfma_lds, s, w.vals, a.masked,
                           a.n_cond, w.att, V, d.d_model, d.n_heads, (int)per_wave);
      } else
      if (sdpa_lds > (size_t)160 * 1024 || (g_debug_flags & 2097152)) {  // no room for the score tile (or bit 21): row-wise
        TW_REQUIRE(dh <= 64, "dense attention: head width %d > 64 on the row-wise per-op kernel", dh);
        const dim3 grid((unsigned)a.n_rows, d.n_heads, (unsigned)((V + 127) / 128));
        if (dh <= 16)
          hipLaunchKernelGGL(sdpa_rows_kernel<16>, grid, dim3(128), 0, s, w.vals, a.masked, a.n_cond, w.att, V, d.d_model, d.n_heads);
        else
          hipLaunchKernelGGL(sdpa_rows_kernel<64>, grid, dim3(128), 0, s, w.vals, a.masked, a.n_cond, w.att, V, d.d_model, d.n_heads);
      } else {
        TW_LDS_LIMIT(sdpa_kernel, sdpa_lds, V);
        hipLaunchKernelGGL(sdpa_kernel, dim3((unsigned)a.n_rows, d.n_heads), dim3(128), sdpa_lds, s, w.vals, a.masked, a.n_cond,
                           w.att, V, d.d_model, d.n_heads);
      }
      TW_LAUNCH_CHECK();
      if ((rc = launch_linear(w.att, lb + L.layer.out_w, lb + L.layer.out_b, w.tmp, M, d.d_model, d.d_model, ACT_NONE, s, sp))) return rc;
    }
    hipLaunchKernelGGL(add_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, w.h, w.tmp, lb + L.layer.n1w,
                       lb + L.layer.n1b, d.ln_eps, d.d_model, M);
    TW_LAUNCH_CHECK();
  ln1_done:;
    if (sp && a.packed && h3_ffn_tokens_supported(d) && !(g_debug_flags & 16777216)) {
      // TW_PATH_SIMPLE_H3 with the split-fp16 stream at hand: FFN + residual + LayerNorm 2 as ONE launch of the fused kernels' chunk
      // loop on the flat token list - the 2048-wide hidden layer stays on the chip (tw_netblock_h3.hip: h3_ffn_tokens_kernel)
      if ((rc = h3_ffn_tokens(d, a.packed, c, net, l, w.h, M, s, w.ff, M * d.d_ff, (const char*)a.packed + simple_h3_split_offset(d)))) return rc;
    } else {
    if ((rc = launch_linear(w.h, lb + L.layer.w1, lb + L.layer.b1, w.ff, M, d.d_ff, d.d_model, ACT_RELU, s, sp))) return rc;
    if ((rc = launch_linear(w.ff, lb + L.layer.w2, lb + L.layer.b2, w.tmp, M, d.d_model, d.d_ff, ACT_NONE, s, sp))) return rc;
    hipLaunchKernelGGL(add_ln_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, w.h, w.tmp, lb + L.layer.n2w,
                       lb + L.layer.n2b, d.ln_eps, d.d_model, M);
    TW_LAUNCH_CHECK();
    }
    if (dump) TW_HIP_CHECK(hipMemcpyAsync(dump + (l + 1) * act_sz, w.h, act_sz * 4, hipMemcpyDeviceToDevice, s));
  }
  if (io_tokens) {
    if ((rc = h3_io_tokens(d, a.packed, c, net, true, w.h, out, 0, M, s))) return rc;
  } else {
  if ((rc = launch_linear(w.h, nb + L.net.out0_w, nb + L.net.out0_b, w.h0, M, d.d_hidden, d.d_model, ACT_SILU, s, sp))) return rc;
  if ((rc = launch_linear(w.h0, nb + L.net.out2_w, nb + L.net.out2_b, out, M, 3, d.d_hidden, ACT_NONE, s, sp))) return rc;
  }
  if (dump) TW_HIP_CHECK(hipMemcpyAsync(dump + (d.n_layers + 1) * act_sz, out, M * 3 * 4, hipMemcpyDeviceToDevice, s));
  return TW_OK;
}

ScoreBasis score_basis(const tw_flow_desc& d, const RawLayout& L, const float* raw, int coupling) {
  ScoreBasis b{nullptr, 0, 0, 0, 0, d.n_layers, 1};
  if (d.variant == 0 && d.cheb_order > 0) {
    b.coeff0 = raw + net_base(L, coupling, 0) + L.net.layers + L.layer.cheb;
    b.net_stride = net_base(L, coupling, 1) - net_base(L, coupling, 0);
    b.layer_stride = L.layer.size;
    b.order = d.cheb_order;
    b.force_zero = d.cheb_force_zero;
    b.n_variants = 2 * d.n_layers;
  }
  return b;
}

static int simple_scores(const FlowArgs& a, const RawLayout& L, const SimpleWs& w) {
  const tw_flow_desc& d = *a.desc;
  if (d.variant != 0 || d.cheb_order > 0) return TW_OK;  // chebyshev_kernel: per layer, in netblock_simple
  // one score matrix per flow call, shared by every encoder layer (model_constructor.py:192-195)
  const bool folded = a.simple_h3 && a.packed && w.s_hi && a.n_atoms > 64 && h3_ffn_tokens_supported(d) && !(g_debug_flags & 16777216);
  const bool direct = folded && scores_split_direct(a.n_atoms);   // the row-wise kernel writes the mixing's split operand itself
  int rc = launch_scores(a.x_coords, a.masked, a.raw + L.lengthscales + (a.reverse ? d.n_heads : 0), d.n_heads, a.n_cond, a.n_atoms, d.normalise,
                         a.n_atoms > 25, w.scores, a.stream, nullptr, 0, 0, direct ? w.s_hi : nullptr, direct ? w.s_lo : nullptr);
  if (rc) return rc;
  if (folded && !direct) {
    // the folded mixing's A operand: split once per flow pass, shared by every layer and both nets
    const int V = a.n_atoms, Vp = (V + 31) / 32 * 32;
    const int64_t rows = a.n_cond * d.n_heads * Vp, total = rows * Vp;
    hipLaunchKernelGGL(split_scores_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, a.stream, w.scores, w.s_hi, w.s_lo,
                       rows, V, Vp);
    TW_LAUNCH_CHECK();
  }
  return TW_OK;
}

int flow_pass_simple(const FlowArgs& a) {
  const tw_flow_desc& d = *a.desc;
  const RawLayout L = raw_layout(d);
  const SimpleWs w = simple_ws(d, a.n_rows, a.n_atoms, a.ws);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  int rc;
  if ((rc = simple_scores(a, L, w))) return rc;
  // (bit 29 - the small-launch measures off - keeps both nets on the caller's stream)
  const bool two = simple_two_streams(w.bytes) && 2 * w.bytes <= a.ws_bytes && !(g_debug_flags & 536870912);
  static thread_local hipStream_t sides[32] = {};
  static thread_local hipEvent_t evs[32][2] = {};
  FlowArgs a2 = a;
  SimpleWs w2 = w;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  if (two) {
    int dev_id = 0;
    TW_HIP_CHECK(hipGetDevice(&dev_id));
    TW_REQUIRE(dev_id >= 0 && dev_id < 32, "device index %d out of range", dev_id);
    if (!sides[dev_id]) {  // one side stream and event pair per device and calling thread, created on first use
      TW_HIP_CHECK(hipStreamCreateWithFlags(&sides[dev_id], hipStreamNonBlocking));
      TW_HIP_CHECK(hipEventCreateWithFlags(&evs[dev_id][0], hipEventDisableTiming));
      TW_HIP_CHECK(hipEventCreateWithFlags(&evs[dev_id][1], hipEventDisableTiming));
    }
    a2.stream = sides[dev_id];
    ev_fork = evs[dev_id][0];
    ev_join = evs[dev_id][1];
    w2 = simple_ws_second(d, a.n_rows, a.n_atoms, w, (char*)a.ws + w.bytes);
  }
  for (int i = 0; i < d.n_coupling; ++i) {
    const int c = a.reverse ? d.n_coupling - 1 - i : i;
    const bool positions = (c % 2) == d.pos_mod2;
    const float* z_other = positions ? a.z_velocs : a.z_coords;
    float* z_t = positions ? a.z_coords : a.z_velocs;
    if (two) {   // everything before (scores, the previous coupling step) -> side stream; its net -> back before the coupling step
      TW_HIP_CHECK(hipEventRecord(ev_fork, a.stream));
      TW_HIP_CHECK(hipStreamWaitEvent(a2.stream, ev_fork, 0));
    }
    if ((rc = netblock_simple(a, L, w, c, 0, z_other, w.s_out, nullptr))) return rc;
    if ((rc = netblock_simple(two ? a2 : a, L, two ? w2 : w, c, 1, z_other, w.t_out, nullptr))) return rc;
    if (two) {
      TW_HIP_CHECK(hipEventRecord(ev_join, a2.stream));
      TW_HIP_CHECK(hipStreamWaitEvent(a.stream, ev_join, 0));
    }
    if ((rc = launch_coupling(w.s_out, w.t_out, a.masked, a.n_cond, z_t, a.delta_logp, a.n_rows, a.n_atoms,
                              a.reverse, a.stream, nullptr, a.desc->range_flag)))
      return rc;
  }
  return TW_OK;
}

int debug_netblock_simple(const FlowArgs& a, int c, int net, const float* z_other, float* dump) {
  const tw_flow_desc& d = *a.desc;
  const RawLayout L = raw_layout(d);
  const SimpleWs w = simple_ws(d, a.n_rows, a.n_atoms, a.ws);
  if (w.bytes > a.ws_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)a.ws_bytes);
    return TW_ERR_WORKSPACE;
  }
  int rc;
  if ((rc = simple_scores(a, L, w))) return rc;
  return netblock_simple(a, L, w, c, net, z_other, w.s_out, dump);
}

// launch helpers used by tw_api.hip -----------------------------------------------------------------
int launch_prior_logp(const float* zc, const float* zv, const uint8_t* masked, int64_t n_cond, const float* prior,
                      const float* delta, float sign, float* out, int64_t n_rows, int V, hipStream_t s) {
  if (n_rows == 0) return TW_OK;
  hipLaunchKernelGGL(prior_logp_kernel, dim3((unsigned)n_rows), dim3(64), 0, s, zc, zv, masked, n_cond, prior, delta,
                     sign, out, V);
  TW_LAUNCH_CHECK();
  return TW_OK;
}
int launch_uncentre_add(const float* xc, const float* com, const float* resid, int64_t n_cond, float* y, int V,
                        int displacement, int64_t n_rows, hipStream_t s) {
  const int64_t total = n_rows * 3 * V;
  if (total == 0) return TW_OK;
  hipLaunchKernelGGL(uncentre_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xc, com, resid, n_cond,
                     y, V, displacement, total);
  TW_LAUNCH_CHECK();
  return TW_OK;
}
int launch_sub(const float* a, const float* b, float* o, int64_t total, hipStream_t s) {
  if (total == 0) return TW_OK;
  hipLaunchKernelGGL(sub_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, b, o, total);
  TW_LAUNCH_CHECK();
  return TW_OK;
}
int launch_kinetic(const float* v, const float* masses, int random_velocs, float kbT, float* out, int64_t n, int V,
                   hipStream_t s) {
  if (n == 0) return TW_OK;
  hipLaunchKernelGGL(kinetic_kernel, dim3((unsigned)n), dim3(64), 0, s, v, masses, random_velocs, kbT, out, V);
  TW_LAUNCH_CHECK();
  return TW_OK;
}
int launch_mh_accept(const float* energy, const float* p_xy, const float* p_yx, const float* u, const float* yc,
                     const float* yv, float* xc, float* xv, float* out_exp, float* out_pacc, uint8_t* out_acc,
                     int32_t* result, int64_t S, int V, hipStream_t s, int64_t n_chains) {
  hipLaunchKernelGGL(mh_accept_kernel, dim3((unsigned)n_chains), dim3(256), 0, s, energy, p_xy, p_yx, u, yc, yv, xc, xv,
                     out_exp, out_pacc, out_acc, result, S, V, n_chains);
  TW_LAUNCH_CHECK();
  return TW_OK;
}
int launch_chirality(const float* coords, const int32_t* centres, const float* ref, int n_centres, uint8_t* changed,
                     int64_t n_rows, int V, hipStream_t s) {
  if (n_rows == 0) return TW_OK;
  hipLaunchKernelGGL(chirality_kernel, dim3((unsigned)((n_rows + 127) / 128)), dim3(128), 0, s, coords, centres, ref,
                     n_centres, changed, n_rows, V);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

}  // namespace tw
