// One whole Metropolis-Hastings iteration of `sample_with_model` (reference utils/evaluation_utils.py:609-713) as
// ONE C-ABI call: proposals + log p(y|x) (flow reverse pass), potential / kinetic energies, chirality guard,
// log p(x~|y~) (flow forward pass), exponent, p_acc, accept test, first accepted index, new chain state.
//
// Why: driven op by op (tw_flow_sample_with_logp, tw_amber_energy, tw_kinetic_energy, tw_flow_log_likelihood,
// tw_mh_accept + a dozen elementwise torch ops) an iteration is ~85 launches, of which the 16 net-block launches are
// 95 % of the time and the other ~70 cost ~4-5 us each plus a boundary: 0.36 ms of a 7.1 ms iteration (r01 profile).
// Here the glue is four kernels - mh_begin, mh_finish, mh_pyx, mh_accept_full - around the two flow passes and one
// energy launch over S + 1 conformations (the current state rides along as row S; it runs on a side stream beside the
// forward pass); nothing is copied: the latent buffers the caller hands in become the proposals in place, and the accept
// kernel writes the new state into buffers of its own.
#include "tw_common.h"

extern "C" int tw_flow_pass(const tw_flow_desc* desc, const float* raw, const float* packed, const int32_t* atom_types,
                            const float* x_coords, const float* x_velocs, const uint8_t* masked, int64_t n_cond,
                            float* z_coords, float* z_velocs, float* delta_logp, int64_t n_rows, int32_t n_atoms,
                            int32_t reverse, int32_t path, void* workspace, int64_t workspace_bytes, void* stream);
extern "C" int64_t tw_flow_workspace_bytes(const tw_flow_desc* desc, int64_t n_rows, int32_t n_atoms);

namespace tw {
int amber_energy(const tw_forcefield* ff, const float* coords, double* out, double* terms, int64_t n, hipStream_t s);

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Normal(0, e^ls).log_prob summed over the unmasked atoms of one row (flow.py:191-203 / 322-334); one wave, same
// arithmetic as prior_logp_kernel
__device__ __forceinline__ float row_prior(const float* zc, const float* zv, const uint8_t* mk, const float* prior, int V) {
  const float sc = expf(prior[0]), sv = expf(prior[1]);
  const float var_c = sc * sc, var_v = sv * sv;
  const float lsc = logf(sc), lsv = logf(sv);
  const float half_log_2pi = 0.91893853320467274178f;
  float ac = 0.f, av = 0.f;
  for (int i = threadIdx.x; i < 3 * V; i += 64) {
    const float keep = mk[i / 3] ? 0.f : 1.f;
    const float a = zc[i], b = zv[i];
    ac += keep * (-(a * a) / (2.f * var_c) - lsc - half_log_2pi);
    av += keep * (-(b * b) / (2.f * var_v) - lsv - half_log_2pi);
  }
  return wsum(ac) + wsum(av);
}

// compute_kinetic_energy (evaluation_utils.py:416-436) of one row; one wave, same arithmetic as kinetic_kernel
__device__ __forceinline__ float row_kinetic(const float* v, const float* masses, int random_velocs, float kbT, int V) {
  float acc = 0.f;
  for (int a = threadIdx.x; a < V; a += 64) {
    const float* p = v + a * 3;
    const float s = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    acc += random_velocs ? s : masses[a] * s;
  }
  acc = wsum(acc);
  return random_velocs ? 0.5f * acc : 0.5f * acc / kbT;
}

// masked arithmetic mean of one row's coordinates (molecule_utils.py:15-29); same arithmetic as centre_kernel
__device__ __forceinline__ void row_com(const float* x, const uint8_t* mk, int V, float (&com)[3]) {
  float sx = 0, sy = 0, sz = 0, cnt = 0;
  for (int a = threadIdx.x; a < V; a += 64) {
    const float keep = mk[a] ? 0.f : 1.f;
    sx += keep * x[a * 3 + 0];
    sy += keep * x[a * 3 + 1];
    sz += keep * x[a * 3 + 2];
    cnt += keep;
  }
  sx = wsum(sx); sy = wsum(sy); sz = wsum(sz); cnt = wsum(cnt);
  com[0] = sx / cnt; com[1] = sy / cnt; com[2] = sz / cnt;
}

// Block n < S: prior log-density of the (still untouched) latents of proposal n, delta_logp = 0.
// Block S: the conditioning state - centred coordinates, centre of mass, kinetic energy (evaluated once instead of on
// S identical copies, evaluation_utils.py:620-629) - and its coordinates as row S of the energy batch.
__global__ void mh_begin_kernel(const float* __restrict__ x_coords, const float* __restrict__ x_velocs,
                                const uint8_t* __restrict__ masked, const float* __restrict__ masses,
                                const float* __restrict__ prior, float* __restrict__ zc, const float* __restrict__ zv,
                                float* __restrict__ xc, float* __restrict__ com_out, float* __restrict__ prior0,
                                float* __restrict__ delta, float* __restrict__ ekin_x, const int32_t* __restrict__ types,
                                int32_t* __restrict__ types_rep, uint8_t* __restrict__ masked_rep, int random_velocs,
                                float kbT, int64_t S, int V) {
  const int64_t n = blockIdx.x;
  if (n < S) {
    // the forward pass conditions every row on its own proposal (n_cond = S): per-row copies of the atom types / mask
    for (int a = threadIdx.x; a < V; a += 64) {
      types_rep[n * V + a] = types[a];
      masked_rep[n * V + a] = masked[a];
    }
    const float p = row_prior(zc + n * 3 * V, zv + n * 3 * V, masked, prior, V);
    if (threadIdx.x == 0) {
      prior0[n] = p;
      delta[n] = 0.f;
    }
    return;
  }
  float com[3];
  row_com(x_coords, masked, V, com);
  for (int i = threadIdx.x; i < 3 * V; i += 64) {
    xc[i] = x_coords[i] - com[i % 3];
    zc[S * 3 * V + i] = x_coords[i];
  }
  if (threadIdx.x < 3) com_out[threadIdx.x] = com[threadIdx.x];
  const float ek = row_kinetic(x_velocs, masses, random_velocs, kbT, V);
  if (threadIdx.x == 0) ekin_x[0] = ek;
}

// After the reverse pass (per proposal n): log p(y|x) = prior(z) + delta_logp, y = (x_centred + com) + residual in
// place (flow.py:303-334), E_kin(y), chirality guard (chirality.py:40-80), and everything the forward pass of the
// reverse move needs (evaluation_utils.py:648-657; flow.py:148-157): residual target x - y, target velocities sgn x_v,
// conditioning velocities sgn y_v, centred conditioning positions y - com(y), delta_logp = 0.
__global__ void mh_finish_kernel(float* __restrict__ zc /* residual in, y out */, const float* __restrict__ yv,
                                 const float* __restrict__ x_coords, const float* __restrict__ x_velocs,
                                 const float* __restrict__ xc, const float* __restrict__ com,
                                 const uint8_t* __restrict__ masked, const float* __restrict__ masses,
                                 const float* __restrict__ prior0, float* __restrict__ delta, float* __restrict__ p_xy,
                                 float* __restrict__ ekin_y, uint8_t* __restrict__ chir, const int32_t* __restrict__ centres,
                                 const float* __restrict__ ref_signs, int n_centres, float* __restrict__ t_c,
                                 float* __restrict__ t_v, float* __restrict__ c_v, float* __restrict__ c_c,
                                 int displacement, int random_velocs, float kbT, int V) {
  extern __shared__ float ys[];  // [3V] the proposal's coordinates
  const int64_t n = blockIdx.x;
  const float sgn = random_velocs ? 1.f : -1.f;
  float* zrow = zc + n * 3 * V;
  for (int i = threadIdx.x; i < 3 * V; i += 64) {
    const float base = xc[i] + com[i % 3];
    const float y = displacement ? base + zrow[i] : zrow[i];
    zrow[i] = y;
    ys[i] = y;
    t_c[n * 3 * V + i] = displacement ? x_coords[i] - y : x_coords[i];
    t_v[n * 3 * V + i] = sgn * x_velocs[i];
    c_v[n * 3 * V + i] = sgn * yv[n * 3 * V + i];
  }
  __syncthreads();
  float cy[3];
  row_com(ys, masked, V, cy);
  for (int i = threadIdx.x; i < 3 * V; i += 64) c_c[n * 3 * V + i] = ys[i] - cy[i % 3];
  const float ek = row_kinetic(yv + n * 3 * V, masses, random_velocs, kbT, V);
  if (threadIdx.x == 0) {
    p_xy[n] = prior0[n] + delta[n];
    delta[n] = 0.f;
    ekin_y[n] = ek;
    bool ch = false;
    for (int c = 0; c < n_centres; ++c) {
      const int i0 = centres[4 * c], i1 = centres[4 * c + 1], i2 = centres[4 * c + 2], i3 = centres[4 * c + 3];
      float a[3], b[3], d[3];
      for (int k = 0; k < 3; ++k) {
        a[k] = ys[3 * i1 + k] - ys[3 * i0 + k];
        b[k] = ys[3 * i2 + k] - ys[3 * i0 + k];
        d[k] = ys[3 * i3 + k] - ys[3 * i0 + k];
      }
      const float cx = b[1] * d[2] - b[2] * d[1];
      const float cyy = b[2] * d[0] - b[0] * d[2];
      const float cz = b[0] * d[1] - b[1] * d[0];
      const float dot = a[0] * cx + a[1] * cyy + a[2] * cz;
      const float s = (dot > 0.f) ? 1.f : ((dot < 0.f) ? -1.f : 0.f);
      if (s != ref_signs[c]) ch = true;
    }
    chir[n] = ch ? 1 : 0;
  }
}

// log p(x~|y~) = prior(z) - delta_logp of the forward pass, one wave per proposal
__global__ void mh_pyx_kernel(const float* __restrict__ zc, const float* __restrict__ zv, const uint8_t* __restrict__ masked,
                              const float* __restrict__ prior, const float* __restrict__ delta, float* __restrict__ p_yx, int V) {
  const int64_t n = blockIdx.x;
  const float p = row_prior(zc + n * 3 * V, zv + n * 3 * V, masked, prior, V);
  if (threadIdx.x == 0) p_yx[n] = p - delta[n];
}

// evaluation_utils.py:628-677 for one chain, single workgroup: energies in kT units (the float32 cast and the
// multiplication by the reciprocal of kbT are what `energy_fn(x) / kbT` does on the op-by-op path), chirality penalty,
// exponent, p_acc = min(1, e^-exponent) with NaN propagating as torch.min does, u < p_acc, first accepted index,
// new state = y[k] or the (resampled) current state.  stats [8, S]: p_acc, p_xy, p_yx, exponent, e_pot_y, e_kin_y,
// e_pot delta, e_kin delta.
__global__ void mh_accept_full_kernel(const double* __restrict__ e_pot, const float* __restrict__ ekin_y,
                                      const float* __restrict__ ekin_x, const uint8_t* __restrict__ chir,
                                      const float* __restrict__ p_xy, const float* __restrict__ p_yx,
                                      const float* __restrict__ u, const float* __restrict__ yc, const float* __restrict__ yv,
                                      const float* __restrict__ x_coords, const float* __restrict__ x_velocs,
                                      float* __restrict__ new_c, float* __restrict__ new_v, float* __restrict__ stats,
                                      uint8_t* __restrict__ out_acc, int32_t* __restrict__ result, float inv_kbT,
                                      int64_t S, int V) {
  __shared__ int first;
  if (threadIdx.x == 0) first = 0x7fffffff;
  __syncthreads();
  const float epx = (float)e_pot[S] * inv_kbT;
  const float ekx = ekin_x[0];
  int local = 0x7fffffff;
  for (int64_t s = threadIdx.x; s < S; s += blockDim.x) {
    float epy = (float)e_pot[s] * inv_kbT;
    if (chir[s]) epy = epy + 2000.f;
    const float eky = ekin_y[s];
    const float dkin = eky - ekx;
    const float dpot = epy - epx;
    const float energy = dpot + dkin;
    const float e = energy + p_xy[s] - p_yx[s];
    const float ee = expf(-e);
    const float p = (ee != ee) ? ee : fminf(1.f, ee);
    const bool acc = u[s] < p;
    stats[0 * S + s] = p;
    stats[1 * S + s] = p_xy[s];
    stats[2 * S + s] = p_yx[s];
    stats[3 * S + s] = e;
    stats[4 * S + s] = epy;
    stats[5 * S + s] = eky;
    stats[6 * S + s] = dpot;
    stats[7 * S + s] = dkin;
    out_acc[s] = acc ? 1 : 0;
    if (acc && (int)s < local) local = (int)s;
  }
  atomicMin(&first, local);
  __syncthreads();
  const int k = first;
  const bool any = k != 0x7fffffff;
  for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) {
    new_c[i] = any ? yc[(int64_t)k * 3 * V + i] : x_coords[i];
    new_v[i] = any ? yv[(int64_t)k * 3 * V + i] : x_velocs[i];
  }
  if (threadIdx.x == 0) {
    result[0] = any ? k : (int)(S - 1);
    result[1] = any ? 1 : 0;
    result[2] = 0;
    result[3] = 0;
  }
}

struct StepWs {
  float *xc, *com, *prior0, *delta, *ekin_x, *ekin_y, *p_xy, *p_yx, *t_c, *t_v, *c_v, *c_c;
  double* e_pot;
  uint8_t *chir, *masked_rep;
  int32_t* types_rep;
  char* flow;
  int64_t flow_bytes, bytes;
};

StepWs step_ws(const tw_flow_desc* d, int64_t S, int V, void* base) {
  StepWs w;
  char* p = (char*)base;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  const int64_t el = S * V * 3 * 4;
  w.xc = (float*)take(V * 3 * 4);
  w.com = (float*)take(16);
  w.prior0 = (float*)take(S * 4);
  w.delta = (float*)take(S * 4);
  w.ekin_x = (float*)take(16);
  w.ekin_y = (float*)take(S * 4);
  w.p_xy = (float*)take(S * 4);
  w.p_yx = (float*)take(S * 4);
  w.e_pot = (double*)take((S + 1) * 8);
  w.chir = (uint8_t*)take(S);
  w.masked_rep = (uint8_t*)take(S * V);
  w.types_rep = (int32_t*)take(S * V * 4);
  w.t_c = (float*)take(el);
  w.t_v = (float*)take(el);
  w.c_v = (float*)take(el);
  w.c_c = (float*)take(el);
  w.flow = p;
  w.flow_bytes = tw_flow_workspace_bytes(d, S, V);
  p += (w.flow_bytes + 255) / 256 * 256;
  w.bytes = p - (char*)base;
  return w;
}


// ---------------------------------------------------------------------------------------------------------------------
// C chains in lock-step (include/timewarp_hip.h: tw_mh_iteration_chains).  The same four glue kernels with a
// chain index: row n of every per-proposal array is proposal n / C of chain n % C (the reference's [S, B] reshape,
// flow.py:284-296), the conditioning state of chain c is row c of x_coords / x_velocs, and the C current states ride
// along as rows S*C .. S*C + C - 1 of the energy launch.
//
// Draws.  With a tw_mh_draws the latents, the resampled velocities and the accept uniforms come from a counter-based
// generator inside mh_begin - no generator launches, nothing to concatenate: Philox4x32-10 (Salmon et al., SC'11),
//   key     = (seed low word, seed high word)
//   counter = (element >> 2, kind | (iteration >> 32) << 4, iteration low word, global chain id)
// one 128-bit block serves four consecutive elements of a (chain, iteration, kind) stream; element e of a latent stream
// is (proposal s, component i) -> s * 3V + i, so a chain's draws do not depend on how many chains run beside it.
// kind 0: coordinate latents, 1: velocity latents, 2: resampled current velocities, 3: accept uniforms.
// Normals: Box-Muller on word pairs (0,1) and (2,3), u1 = w * 2^-32 + 2^-33 in (0, 1], u2 = (w >> 8) * 2^-24 in [0, 1).

struct Draws {
  uint32_t k0, k1, it_lo, it_hi;
  int chain0;
  int on, resample;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void draw_block(const Draws& d, int chain, int kind, uint32_t block, uint32_t (&w)[4]) {
  philox4x32_10(block, (uint32_t)kind | (d.it_hi << 4), d.it_lo, (uint32_t)(d.chain0 + chain), d.k0, d.k1, w);
}

__device__ __forceinline__ float draw_normal(const Draws& d, int chain, int kind, uint32_t e) {
  uint32_t w[4];
  draw_block(d, chain, kind, e >> 2, w);
  const int pair = (e >> 1) & 1;
  const float u1 = fmaf((float)w[2 * pair], 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  const float u2 = (float)(w[2 * pair + 1] >> 8) * 5.9604644775390625e-08f;
  const float r = sqrtf(-2.f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  return (e & 1) ? r * sn : r * cs;
}

__device__ __forceinline__ float draw_uniform(const Draws& d, int chain, uint32_t e) {
  uint32_t w[4];
  draw_block(d, chain, 3, e >> 2, w);
  return (float)(w[e & 3] >> 8) * 5.9604644775390625e-08f;  // [0, 1), 24 bits: what torch.rand's float32 holds
}

// Blocks n < S*C: (draws) latents of proposal n, then as mh_begin_kernel.  Blocks S*C + c: chain c's conditioning state;
// cur_v [C, V, 3] = the velocities this iteration runs with (resampled when the draws say so, else a copy of x_velocs).
__global__ void mhc_begin_kernel(const float* __restrict__ x_coords, const float* __restrict__ x_velocs,
                                 const uint8_t* __restrict__ masked, const float* __restrict__ masses,
                                 const float* __restrict__ prior, float* __restrict__ zc, float* __restrict__ zv,
                                 float* __restrict__ u, float* __restrict__ cur_v, float* __restrict__ xc,
                                 float* __restrict__ com_out, float* __restrict__ prior0, float* __restrict__ delta,
                                 float* __restrict__ ekin_x, const int32_t* __restrict__ types,
                                 int32_t* __restrict__ types_rep, uint8_t* __restrict__ masked_rep, int random_velocs,
                                 float kbT, int64_t S, int C, int V, Draws dr) {
  const int64_t n = blockIdx.x, rows = S * C;
  if (n < rows) {
    const int c = (int)(n % C);
    const int64_t sidx = n / C;
    for (int a = threadIdx.x; a < V; a += 64) {
      types_rep[n * V + a] = types[c * V + a];
      masked_rep[n * V + a] = masked[c * V + a];
    }
    if (dr.on) {
      const float sc = expf(prior[0]), sv = expf(prior[1]);
      for (int i = threadIdx.x; i < 3 * V; i += 64) {   // (row_prior below reads back what this same lane wrote)
        const uint32_t e = (uint32_t)(sidx * 3 * V + i);
        zc[n * 3 * V + i] = draw_normal(dr, c, 0, e) * sc;
        zv[n * 3 * V + i] = draw_normal(dr, c, 1, e) * sv;
      }
      if (threadIdx.x == 0) u[n] = draw_uniform(dr, c, (uint32_t)sidx);
    }
    const float p = row_prior(zc + n * 3 * V, zv + n * 3 * V, masked + c * V, prior, V);
    if (threadIdx.x == 0) {
      prior0[n] = p;
      delta[n] = 0.f;
    }
    return;
  }
  const int c = (int)(n - rows);
  const float* x = x_coords + (int64_t)c * 3 * V;
  float* v = cur_v + (int64_t)c * 3 * V;
  for (int i = threadIdx.x; i < 3 * V; i += 64)
    v[i] = (dr.on && dr.resample) ? draw_normal(dr, c, 2, (uint32_t)i) : x_velocs[(int64_t)c * 3 * V + i];
  __syncthreads();
  float com[3];
  row_com(x, masked + c * V, V, com);
  for (int i = threadIdx.x; i < 3 * V; i += 64) {
    xc[(int64_t)c * 3 * V + i] = x[i] - com[i % 3];
    zc[(rows + c) * 3 * V + i] = x[i];
  }
  if (threadIdx.x < 3) com_out[3 * c + threadIdx.x] = com[threadIdx.x];
  const float ek = row_kinetic(v, masses, random_velocs, kbT, V);
  if (threadIdx.x == 0) ekin_x[c] = ek;
}

// mh_finish_kernel with the chain of the row: conditioning state, centre of mass and mask of chain n % C
__global__ void mhc_finish_kernel(float* __restrict__ zc, const float* __restrict__ yv, const float* __restrict__ x_coords,
                                  const float* __restrict__ x_velocs, const float* __restrict__ xc,
                                  const float* __restrict__ com, const uint8_t* __restrict__ masked,
                                  const float* __restrict__ masses, const float* __restrict__ prior0,
                                  float* __restrict__ delta, float* __restrict__ p_xy, float* __restrict__ ekin_y,
                                  uint8_t* __restrict__ chir, const int32_t* __restrict__ centres,
                                  const float* __restrict__ ref_signs, int n_centres, float* __restrict__ t_c,
                                  float* __restrict__ t_v, float* __restrict__ c_v, float* __restrict__ c_c,
                                  int displacement, int random_velocs, float kbT, int C, int V) {
  extern __shared__ float ys[];
  const int64_t n = blockIdx.x;
  const int c = (int)(n % C);
  const float sgn = random_velocs ? 1.f : -1.f;
  const float *xq = x_coords + (int64_t)c * 3 * V, *xw = x_velocs + (int64_t)c * 3 * V, *xcc = xc + (int64_t)c * 3 * V;
  const uint8_t* mk = masked + c * V;
  float* zrow = zc + n * 3 * V;
  for (int i = threadIdx.x; i < 3 * V; i += 64) {
    const float base = xcc[i] + com[3 * c + i % 3];
    const float y = displacement ? base + zrow[i] : zrow[i];
    zrow[i] = y;
    ys[i] = y;
    t_c[n * 3 * V + i] = displacement ? xq[i] - y : xq[i];
    t_v[n * 3 * V + i] = sgn * xw[i];
    c_v[n * 3 * V + i] = sgn * yv[n * 3 * V + i];
  }
  __syncthreads();
  float cy[3];
  row_com(ys, mk, V, cy);
  for (int i = threadIdx.x; i < 3 * V; i += 64) c_c[n * 3 * V + i] = ys[i] - cy[i % 3];
  const float ek = row_kinetic(yv + n * 3 * V, masses, random_velocs, kbT, V);
  if (threadIdx.x == 0) {
    p_xy[n] = prior0[n] + delta[n];
    delta[n] = 0.f;
    ekin_y[n] = ek;
    bool ch = false;
    for (int q = 0; q < n_centres; ++q) {
      const int i0 = centres[4 * q], i1 = centres[4 * q + 1], i2 = centres[4 * q + 2], i3 = centres[4 * q + 3];
      float a[3], b[3], d[3];
      for (int k = 0; k < 3; ++k) {
        a[k] = ys[3 * i1 + k] - ys[3 * i0 + k];
        b[k] = ys[3 * i2 + k] - ys[3 * i0 + k];
        d[k] = ys[3 * i3 + k] - ys[3 * i0 + k];
      }
      const float cx = b[1] * d[2] - b[2] * d[1];
      const float cyy = b[2] * d[0] - b[0] * d[2];
      const float cz = b[0] * d[1] - b[1] * d[0];
      const float dot = a[0] * cx + a[1] * cyy + a[2] * cz;
      const float sg = (dot > 0.f) ? 1.f : ((dot < 0.f) ? -1.f : 0.f);
      if (sg != ref_signs[q]) ch = true;
    }
    chir[n] = ch ? 1 : 0;
  }
}

__global__ void mhc_pyx_kernel(const float* __restrict__ zc, const float* __restrict__ zv, const uint8_t* __restrict__ masked,
                               const float* __restrict__ prior, const float* __restrict__ delta, float* __restrict__ p_yx,
                               int C, int V) {
  const int64_t n = blockIdx.x;
  const float p = row_prior(zc + n * 3 * V, zv + n * 3 * V, masked + (n % C) * V, prior, V);
  if (threadIdx.x == 0) p_yx[n] = p - delta[n];
}

// mh_accept_full_kernel, one workgroup per chain: proposal s of chain c is row s*C + c of every array; stats [8, S, C]
__global__ void mhc_accept_kernel(const double* __restrict__ e_pot, const float* __restrict__ ekin_y,
                                  const float* __restrict__ ekin_x, const uint8_t* __restrict__ chir,
                                  const float* __restrict__ p_xy, const float* __restrict__ p_yx,
                                  const float* __restrict__ u, const float* __restrict__ yc, const float* __restrict__ yv,
                                  const float* __restrict__ x_coords, const float* __restrict__ cur_v,
                                  float* __restrict__ new_c, float* __restrict__ new_v, float* __restrict__ stats,
                                  uint8_t* __restrict__ out_acc, int32_t* __restrict__ result, float inv_kbT, int64_t S, int C,
                                  int V) {
  __shared__ int first;
  const int c = blockIdx.x;
  const int64_t rows = S * C;
  if (threadIdx.x == 0) first = 0x7fffffff;
  __syncthreads();
  const float epx = (float)e_pot[rows + c] * inv_kbT;
  const float ekx = ekin_x[c];
  int local = 0x7fffffff;
  for (int64_t s = threadIdx.x; s < S; s += blockDim.x) {
    const int64_t n = s * C + c;
    float epy = (float)e_pot[n] * inv_kbT;
    if (chir[n]) epy = epy + 2000.f;
    const float eky = ekin_y[n];
    const float dkin = eky - ekx;
    const float dpot = epy - epx;
    const float energy = dpot + dkin;
    const float e = energy + p_xy[n] - p_yx[n];
    const float ee = expf(-e);
    const float p = (ee != ee) ? ee : fminf(1.f, ee);
    const bool acc = u[n] < p;
    stats[0 * rows + n] = p;
    stats[1 * rows + n] = p_xy[n];
    stats[2 * rows + n] = p_yx[n];
    stats[3 * rows + n] = e;
    stats[4 * rows + n] = epy;
    stats[5 * rows + n] = eky;
    stats[6 * rows + n] = dpot;
    stats[7 * rows + n] = dkin;
    out_acc[n] = acc ? 1 : 0;
    if (acc && (int)s < local) local = (int)s;
  }
  atomicMin(&first, local);
  __syncthreads();
  const int k = first;
  const bool any = k != 0x7fffffff;
  for (int i = threadIdx.x; i < 3 * V; i += blockDim.x) {
    const int64_t src = ((int64_t)k * C + c) * 3 * V + i;
    new_c[(int64_t)c * 3 * V + i] = any ? yc[src] : x_coords[(int64_t)c * 3 * V + i];
    new_v[(int64_t)c * 3 * V + i] = any ? yv[src] : cur_v[(int64_t)c * 3 * V + i];
  }
  if (threadIdx.x == 0) {
    result[4 * c + 0] = any ? k : (int)(S - 1);
    result[4 * c + 1] = any ? 1 : 0;
    result[4 * c + 2] = 0;
    result[4 * c + 3] = 0;
  }
}

// the draws alone (tw_mh_draw_chains): what mhc_begin_kernel generates, written where a caller can look at it
__global__ void mhc_draws_kernel(const float* __restrict__ prior, float* __restrict__ zc, float* __restrict__ zv,
                                 float* __restrict__ u, float* __restrict__ cur_v, int64_t S, int C, int V, Draws dr) {
  const int64_t n = blockIdx.x, rows = S * C;
  if (n < rows) {
    const int c = (int)(n % C);
    const int64_t sidx = n / C;
    const float sc = expf(prior[0]), sv = expf(prior[1]);
    for (int i = threadIdx.x; i < 3 * V; i += 64) {
      const uint32_t e = (uint32_t)(sidx * 3 * V + i);
      if (zc) zc[n * 3 * V + i] = draw_normal(dr, c, 0, e) * sc;
      if (zv) zv[n * 3 * V + i] = draw_normal(dr, c, 1, e) * sv;
    }
    if (u && threadIdx.x == 0) u[n] = draw_uniform(dr, c, (uint32_t)sidx);
    return;
  }
  const int c = (int)(n - rows);
  if (cur_v)
    for (int i = threadIdx.x; i < 3 * V; i += 64) cur_v[(int64_t)c * 3 * V + i] = draw_normal(dr, c, 2, (uint32_t)i);
}

Draws make_draws(const tw_mh_draws* d) {
  Draws r{};
  if (!d) return r;
  r.on = 1;
  r.k0 = (uint32_t)(d->seed & 0xffffffffull);
  r.k1 = (uint32_t)(d->seed >> 32);
  r.it_lo = (uint32_t)((uint64_t)d->iteration & 0xffffffffull);
  r.it_hi = (uint32_t)(((uint64_t)d->iteration >> 32) & 0x0fffffffull);
  r.chain0 = d->first_chain;
  r.resample = d->resample_velocs;
  return r;
}

struct ChainsWs {
  float *xc, *com, *prior0, *delta, *ekin_x, *ekin_y, *p_xy, *p_yx, *t_c, *t_v, *c_v, *c_c;
  double* e_pot;
  uint8_t *chir, *masked_rep;
  int32_t* types_rep;
  char* flow;
  int64_t flow_bytes, bytes;
};

ChainsWs chains_ws(const tw_flow_desc* d, int64_t S, int64_t C, int V, void* base) {
  ChainsWs w;
  char* p = (char*)base;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  const int64_t rows = S * C, el = rows * V * 3 * 4;
  w.xc = (float*)take(C * V * 3 * 4);
  w.com = (float*)take(C * 3 * 4);
  w.prior0 = (float*)take(rows * 4);
  w.delta = (float*)take(rows * 4);
  w.ekin_x = (float*)take(C * 4);
  w.ekin_y = (float*)take(rows * 4);
  w.p_xy = (float*)take(rows * 4);
  w.p_yx = (float*)take(rows * 4);
  w.e_pot = (double*)take((rows + C) * 8);
  w.chir = (uint8_t*)take(rows);
  w.masked_rep = (uint8_t*)take(rows * V);
  w.types_rep = (int32_t*)take(rows * V * 4);
  w.t_c = (float*)take(el);
  w.t_v = (float*)take(el);
  w.c_v = (float*)take(el);
  w.c_c = (float*)take(el);
  w.flow = p;
  w.flow_bytes = tw_flow_workspace_bytes(d, rows, V);
  p += (w.flow_bytes + 255) / 256 * 256;
  w.bytes = p - (char*)base;
  return w;
}

}  // namespace
}  // namespace tw

using namespace tw;

extern "C" {

int64_t tw_mh_iteration_workspace_bytes(const tw_flow_desc* desc, int64_t n_proposals, int32_t n_atoms) {
  if (!desc || n_proposals <= 0 || n_atoms <= 0) return -1;
  if (tw_flow_workspace_bytes(desc, n_proposals, n_atoms) < 0) return -1;
  return step_ws(desc, n_proposals, n_atoms, nullptr).bytes;
}

int tw_mh_iteration(const tw_flow_desc* desc, const float* raw, const void* packed, int32_t path, const tw_forcefield* ff,
                    const tw_mh_options* opt, const int32_t* atom_types, const uint8_t* masked, int32_t n_atoms,
                    const float* x_coords, const float* x_velocs, float* zy_coords, float* zy_velocs, const float* u,
                    float* new_coords, float* new_velocs, float* out_stats, uint8_t* out_accepted, int32_t* result,
                    int64_t n_proposals, void* workspace, int64_t workspace_bytes, void* stream) {
  TW_REQUIRE(desc && raw && ff && opt && atom_types && masked && x_coords && x_velocs && zy_coords && zy_velocs && u &&
                 new_coords && new_velocs && out_stats && out_accepted && result && workspace,
             "NULL pointer argument");
  TW_REQUIRE(n_proposals > 0 && n_proposals < (1LL << 30) && n_atoms > 0, "bad sizes");
  TW_REQUIRE(ff->n_atoms == n_atoms, "force field is for %d atoms, the molecule has %d", ff->n_atoms, n_atoms);
  TW_REQUIRE(opt->kbT > 0.f, "kbT must be positive");
  TW_REQUIRE(opt->random_velocs || opt->masses, "masses are needed unless random_velocs");
  TW_REQUIRE(opt->n_centres == 0 || (opt->centres && opt->reference_signs), "chirality guard needs centres and signs");
  TW_REQUIRE(!desc->ignore_cond_velocity, "ignore_conditional_velocity models take the op-by-op route");
  const int64_t S = n_proposals;
  const int V = n_atoms;
  hipStream_t s = (hipStream_t)stream;
  const StepWs w = step_ws(desc, S, V, workspace);
  if (w.flow_bytes < 0) return TW_ERR_INVALID;
  if (w.bytes > workspace_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)workspace_bytes);
    return TW_ERR_WORKSPACE;
  }
  const RawLayout L = raw_layout(*desc);
  const float* prior = raw + L.prior;
  const float* masses = opt->masses ? opt->masses : x_velocs;  // never read when random_velocs
  int rc;
  hipLaunchKernelGGL(mh_begin_kernel, dim3((unsigned)(S + 1)), dim3(64), 0, s, x_coords, x_velocs, masked, masses, prior,
                     zy_coords, zy_velocs, w.xc, w.com, w.prior0, w.delta, w.ekin_x, atom_types, w.types_rep, w.masked_rep,
                     opt->random_velocs, opt->kbT, S, V);
  TW_LAUNCH_CHECK();
  // proposals: flow reverse pass on the latents, in place (flow.py:284-300)
  if ((rc = tw_flow_pass(desc, raw, (const float*)packed, atom_types, w.xc, x_velocs, masked, 1, zy_coords, zy_velocs, w.delta,
                         S, V, 1, path, w.flow, w.flow_bytes, stream)))
    return rc;
  hipLaunchKernelGGL(mh_finish_kernel, dim3((unsigned)S), dim3(64), (size_t)3 * V * sizeof(float), s, zy_coords, zy_velocs,
                     x_coords, x_velocs, w.xc, w.com, masked, masses, w.prior0, w.delta, w.p_xy, w.ekin_y, w.chir,
                     opt->centres, opt->reference_signs, opt->n_centres, w.t_c, w.t_v, w.c_v, w.c_c, desc->displacement,
                     opt->random_velocs, opt->kbT, V);
  TW_LAUNCH_CHECK();
  // Potential energy of the S proposals and of the current state (row S) in one launch - on a side stream: it depends
  // on the proposals only and nothing needs it before the accept test, so it runs beside the forward pass (whose
  // net-block workgroups leave a few CUs free) instead of between two launches of the main stream.
  static thread_local hipStream_t sides[32] = {};
  static thread_local hipEvent_t evs[32][2] = {};
  int dev_id = 0;
  TW_HIP_CHECK(hipGetDevice(&dev_id));
  TW_REQUIRE(dev_id >= 0 && dev_id < 32, "device index %d out of range", dev_id);
  if (!sides[dev_id]) {  // one side stream and event pair per device and calling thread, created on first use
    TW_HIP_CHECK(hipStreamCreateWithFlags(&sides[dev_id], hipStreamNonBlocking));
    TW_HIP_CHECK(hipEventCreateWithFlags(&evs[dev_id][0], hipEventDisableTiming));
    TW_HIP_CHECK(hipEventCreateWithFlags(&evs[dev_id][1], hipEventDisableTiming));
  }
  hipStream_t side = sides[dev_id];
  hipEvent_t ev_y = evs[dev_id][0], ev_e = evs[dev_id][1];
  // Side stream or not?  The overlap pays only while the flow's launches leave compute units idle.  Once they fill the chip
  // (from ~160 workgroups of 192 token slots per net pair on) the energy kernel merely time-shares with the first net-block
  // launch of the forward pass, and the two event hand-offs cost the main stream 6-8 us each: measured r05
  // (tools/ab_energy.py, one box) inline is 25 us per iteration faster on alanine dipeptide x 1000 (0.4 %), 70 us on the dense
  // model, neutral at 61-65 atoms x 512.  Bit 22 forces the main stream, bit 23 the side stream (A/B).
  const int dbg = g_debug_flags;
  const bool inline_energy = (dbg & 8388608) ? false : ((dbg & 4194304) ? true : 2 * S * (int64_t)V >= (int64_t)160 * 192);
  if (inline_energy) {
    if ((rc = amber_energy(ff, zy_coords, w.e_pot, nullptr, S + 1, s))) return rc;
  } else {
    TW_HIP_CHECK(hipEventRecord(ev_y, s));
    TW_HIP_CHECK(hipStreamWaitEvent(side, ev_y, 0));
    if ((rc = amber_energy(ff, zy_coords, w.e_pot, nullptr, S + 1, side))) return rc;
    TW_HIP_CHECK(hipEventRecord(ev_e, side));
  }
  // reverse move: flow forward pass, every row conditioned on its own proposal (evaluation_utils.py:648-657)
  if ((rc = tw_flow_pass(desc, raw, (const float*)packed, w.types_rep, w.c_c, w.c_v, w.masked_rep, S, w.t_c, w.t_v, w.delta, S,
                         V, 0, path, w.flow, w.flow_bytes, stream)))
    return rc;
  hipLaunchKernelGGL(mh_pyx_kernel, dim3((unsigned)S), dim3(64), 0, s, w.t_c, w.t_v, masked, prior, w.delta, w.p_yx, V);
  TW_LAUNCH_CHECK();
  if (!inline_energy) TW_HIP_CHECK(hipStreamWaitEvent(s, ev_e, 0));
  hipLaunchKernelGGL(mh_accept_full_kernel, dim3(1), dim3(1024), 0, s, w.e_pot, w.ekin_y, w.ekin_x, w.chir, w.p_xy, w.p_yx, u,
                     zy_coords, zy_velocs, x_coords, x_velocs, new_coords, new_velocs, out_stats, out_accepted, result,
                     1.0f / opt->kbT, S, V);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

int64_t tw_mh_iteration_chains_workspace_bytes(const tw_flow_desc* desc, int64_t n_proposals, int64_t n_chains, int32_t n_atoms) {
  if (!desc || n_proposals <= 0 || n_chains <= 0 || n_atoms <= 0) return -1;
  if (tw_flow_workspace_bytes(desc, n_proposals * n_chains, n_atoms) < 0) return -1;
  return chains_ws(desc, n_proposals, n_chains, n_atoms, nullptr).bytes;
}

int tw_mh_iteration_chains(const tw_flow_desc* desc, const float* raw, const void* packed, int32_t path, const tw_forcefield* ff,
                           const tw_mh_options* opt, const tw_mh_draws* draws, const int32_t* atom_types, const uint8_t* masked,
                           int32_t n_atoms, const float* x_coords, const float* x_velocs, float* cur_velocs, float* zy_coords,
                           float* zy_velocs, float* u, float* new_coords, float* new_velocs, float* out_stats,
                           uint8_t* out_accepted, int32_t* result, int64_t n_proposals, int64_t n_chains, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  TW_REQUIRE(desc && raw && ff && opt && atom_types && masked && x_coords && x_velocs && cur_velocs && zy_coords && zy_velocs &&
                 u && new_coords && new_velocs && out_stats && out_accepted && result && workspace,
             "NULL pointer argument");
  TW_REQUIRE(n_proposals > 0 && n_chains > 0 && n_atoms > 0 && n_proposals * n_chains < (1LL << 30), "bad sizes");
  TW_REQUIRE(n_chains < (1 << 20), "at most 2^20 chains per call");
  TW_REQUIRE(ff->n_atoms == n_atoms, "force field is for %d atoms, the molecule has %d", ff->n_atoms, n_atoms);
  TW_REQUIRE(opt->kbT > 0.f, "kbT must be positive");
  TW_REQUIRE(opt->random_velocs || opt->masses, "masses are needed unless random_velocs");
  TW_REQUIRE(opt->n_centres == 0 || (opt->centres && opt->reference_signs), "chirality guard needs centres and signs");
  TW_REQUIRE(!desc->ignore_cond_velocity, "ignore_conditional_velocity models take the op-by-op route");
  TW_REQUIRE(!draws || (draws->iteration >= 0 && draws->first_chain >= 0), "draws: negative iteration or chain id");
  TW_REQUIRE(!draws || !draws->resample_velocs || opt->random_velocs, "draws: resampled velocities need random_velocs");
  TW_REQUIRE(!draws || n_proposals * 3 * (int64_t)n_atoms < (1LL << 32), "draws: a chain's latent stream has 2^32 elements");
  const int64_t S = n_proposals, rows = n_proposals * n_chains;
  const int Cn = (int)n_chains, V = n_atoms;
  hipStream_t s = (hipStream_t)stream;
  const ChainsWs w = chains_ws(desc, S, n_chains, V, workspace);
  if (w.flow_bytes < 0) return TW_ERR_INVALID;
  if (w.bytes > workspace_bytes) {
    set_error("workspace too small: need %lld bytes, have %lld", (long long)w.bytes, (long long)workspace_bytes);
    return TW_ERR_WORKSPACE;
  }
  const RawLayout L = raw_layout(*desc);
  const float* prior = raw + L.prior;
  const float* masses = opt->masses ? opt->masses : x_velocs;  // never read when random_velocs
  const Draws dr = make_draws(draws);
  int rc;
  hipLaunchKernelGGL(mhc_begin_kernel, dim3((unsigned)(rows + Cn)), dim3(64), 0, s, x_coords, x_velocs, masked, masses, prior,
                     zy_coords, zy_velocs, u, cur_velocs, w.xc, w.com, w.prior0, w.delta, w.ekin_x, atom_types, w.types_rep,
                     w.masked_rep, opt->random_velocs, opt->kbT, S, Cn, V, dr);
  TW_LAUNCH_CHECK();
  // proposals of every chain: ONE flow reverse pass, row n conditioned on chain n % C (flow.py:284-300)
  if ((rc = tw_flow_pass(desc, raw, (const float*)packed, atom_types, w.xc, cur_velocs, masked, n_chains, zy_coords, zy_velocs,
                         w.delta, rows, V, 1, path, w.flow, w.flow_bytes, stream)))
    return rc;
  hipLaunchKernelGGL(mhc_finish_kernel, dim3((unsigned)rows), dim3(64), (size_t)3 * V * sizeof(float), s, zy_coords, zy_velocs,
                     x_coords, cur_velocs, w.xc, w.com, masked, masses, w.prior0, w.delta, w.p_xy, w.ekin_y, w.chir,
                     opt->centres, opt->reference_signs, opt->n_centres, w.t_c, w.t_v, w.c_v, w.c_c, desc->displacement,
                     opt->random_velocs, opt->kbT, Cn, V);
  TW_LAUNCH_CHECK();
  // S*C proposals + the C current states in one energy launch, in line (C chains fill the chip: tw_mh_iteration's side
  // stream pays only while the flow's launches leave compute units idle)
  if ((rc = amber_energy(ff, zy_coords, w.e_pot, nullptr, rows + Cn, s))) return rc;
  // reverse moves: ONE flow forward pass, every row conditioned on its own proposal (evaluation_utils.py:648-657)
  if ((rc = tw_flow_pass(desc, raw, (const float*)packed, w.types_rep, w.c_c, w.c_v, w.masked_rep, rows, w.t_c, w.t_v, w.delta,
                         rows, V, 0, path, w.flow, w.flow_bytes, stream)))
    return rc;
  hipLaunchKernelGGL(mhc_pyx_kernel, dim3((unsigned)rows), dim3(64), 0, s, w.t_c, w.t_v, masked, prior, w.delta, w.p_yx, Cn, V);
  TW_LAUNCH_CHECK();
  hipLaunchKernelGGL(mhc_accept_kernel, dim3((unsigned)Cn), dim3(S >= 512 ? 1024 : 256), 0, s, w.e_pot, w.ekin_y, w.ekin_x,
                     w.chir, w.p_xy, w.p_yx, u, zy_coords, zy_velocs, x_coords, cur_velocs, new_coords, new_velocs, out_stats,
                     out_accepted, result, 1.0f / opt->kbT, S, Cn, V);
  TW_LAUNCH_CHECK();
  return TW_OK;
}

int tw_mh_draw_chains(const tw_flow_desc* desc, const float* raw, const tw_mh_draws* draws, float* z_coords, float* z_velocs,
                      float* u, float* velocs, int64_t n_proposals, int64_t n_chains, int32_t n_atoms, void* stream) {
  TW_REQUIRE(desc && raw && draws, "NULL pointer argument");
  TW_REQUIRE(n_proposals >= 0 && n_chains > 0 && n_atoms > 0 && n_proposals * n_chains < (1LL << 30), "bad sizes");
  TW_REQUIRE(draws->iteration >= 0 && draws->first_chain >= 0, "draws: negative iteration or chain id");
  TW_REQUIRE(n_proposals * 3 * (int64_t)n_atoms < (1LL << 32), "draws: a chain's latent stream has 2^32 elements");
  const RawLayout L = raw_layout(*desc);
  hipLaunchKernelGGL(mhc_draws_kernel, dim3((unsigned)(n_proposals * n_chains + n_chains)), dim3(64), 0, (hipStream_t)stream,
                     raw + L.prior, z_coords, z_velocs, u, velocs, n_proposals, (int)n_chains, n_atoms, make_draws(draws));
  TW_LAUNCH_CHECK();
  return TW_OK;
}

}  // extern "C"
