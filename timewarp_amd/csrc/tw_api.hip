// extern "C" entry points of libtimewarp_hip.so (see include/timewarp_hip.h).
#include "tw_common.h"

namespace tw {
const char* last_error();
int launch_prior_logp(const float* zc, const float* zv, const uint8_t* masked, int64_t n_cond, const float* prior,
                      const float* delta, float sign, float* out, int64_t n_rows, int V, hipStream_t s);
int launch_uncentre_add(const float* xc, const float* com, const float* resid, int64_t n_cond, float* y, int V,
                        int displacement, int64_t n_rows, hipStream_t s);
int launch_sub(const float* a, const float* b, float* o, int64_t total, hipStream_t s);
int launch_kinetic(const float* v, const float* masses, int random_velocs, float kbT, float* out, int64_t n, int V,
                   hipStream_t s);
int launch_mh_accept(const float* energy, const float* p_xy, const float* p_yx, const float* u, const float* yc,
                     const float* yv, float* xc, float* xv, float* out_exp, float* out_pacc, uint8_t* out_acc,
                     int32_t* result, int64_t S, int V, hipStream_t s, int64_t n_chains = 1);
int launch_chirality(const float* coords, const int32_t* centres, const float* ref, int n_centres, uint8_t* changed,
                     int64_t n_rows, int V, hipStream_t s);
int amber_energy(const tw_forcefield* ff, const float* coords, double* out, double* terms, int64_t n, hipStream_t s);
int amber_energy_forces(const tw_forcefield* ff, const float* coords, double* out_energy, double* out_forces, int64_t n, hipStream_t s);
int langevin_steps(const tw_forcefield* ff, const float* masses, float* coords, float* velocs, int n_steps, double dt,
                   double friction, double kbT, int scheme, unsigned long long seed, long long step0, double* out_energy,
                   int64_t n, hipStream_t s);
}  // namespace tw

using namespace tw;

static int check_desc(const tw_flow_desc* d) {
  TW_REQUIRE(d != nullptr, "desc is NULL");
  TW_REQUIRE(d->variant == 0 || d->variant == 1, "unknown variant %d", d->variant);
  TW_REQUIRE(d->n_coupling > 0 && d->n_layers > 0 && d->d_model > 0 && d->d_ff > 0 && d->d_hidden > 0 &&
                 d->d_emb > 0 && d->n_heads > 0 && d->n_elements > 0,
             "non-positive dimension in tw_flow_desc");
  TW_REQUIRE(d->variant == 0 || d->d_model % d->n_heads == 0, "d_model %% n_heads != 0");
  TW_REQUIRE(d->d_rff >= 0 && d->d_rff % 2 == 0, "d_rff must be even");
  TW_REQUIRE(d->cheb_order >= 0 && d->cheb_order <= 64 && (d->cheb_order == 0 || d->variant == 0), "bad cheb_order");
  return TW_OK;
}

static bool fused_supported(const tw_flow_desc& d, int n_atoms) {
  FusedGeom g;
  if (d.variant == 1) return dense_fused_supported(d, n_atoms);
  return d.variant == 0 && d.d_model == 128 && d.d_hidden % 32 == 0 && d.d_ff % 32 == 0 && d.d_emb % 4 == 0 &&
         d.d_emb + 9 <= 48 && fused_geom(n_atoms, &g);
}

static int resolve_path(const tw_flow_desc& d, int n_atoms, int path, const float* packed, int* out) {
  if (path == TW_PATH_AUTO) path = (packed && fused_supported(d, n_atoms)) ? TW_PATH_FUSED : TW_PATH_SIMPLE;
  if (path == TW_PATH_FUSED_H3) {
    TW_REQUIRE(h3_supported(d, n_atoms), "split-fp16 path unsupported for this config (variant=%d d_model=%d n_atoms=%d)",
               d.variant, d.d_model, n_atoms);
    TW_REQUIRE(packed != nullptr, "split-fp16 path needs its packed weight stream (tw_flow_pack_h3)");
  } else if (path == TW_PATH_FUSED_H1) {
    TW_REQUIRE(h1_supported(d, n_atoms), "single-MFMA path unsupported for this config (variant=%d d_model=%d n_atoms=%d)",
               d.variant, d.d_model, n_atoms);
    TW_REQUIRE(packed != nullptr, "single-MFMA path needs its packed weight stream (tw_flow_pack_h1)");
  } else if (path == TW_PATH_FUSED) {
    TW_REQUIRE(fused_supported(d, n_atoms), "fused path unsupported for this config (variant=%d d_model=%d n_atoms=%d)",
               d.variant, d.d_model, n_atoms);
    TW_REQUIRE(packed != nullptr, "fused path needs the packed weight stream (tw_flow_pack)");
  } else {
    TW_REQUIRE(path == TW_PATH_SIMPLE || path == TW_PATH_SIMPLE_H3, "unknown path %d", path);
  }
  *out = path;
  return TW_OK;
}

extern "C" {

const char* tw_last_error(void) { return tw::last_error(); }
int tw_abi_version(void) { return TW_ABI_VERSION; }

int tw_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int64_t tw_flow_raw_floats(const tw_flow_desc* desc) {
  if (check_desc(desc)) return -1;
  return raw_layout(*desc).total;
}

int64_t tw_flow_packed_floats(const tw_flow_desc* desc) {
  if (check_desc(desc)) return -1;
  if (!fused_supported(*desc, 22)) return 0;
  return packed_layout(*desc).total;
}

int tw_flow_pack(const tw_flow_desc* desc, const float* raw, float* packed, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  TW_REQUIRE(fused_supported(*desc, 22), "fused path unsupported for this config");
  TW_REQUIRE(raw && packed, "NULL buffer");
  return pack_weights(*desc, raw, packed, (hipStream_t)stream);
}

int64_t tw_flow_packed_h3_bytes(const tw_flow_desc* desc) {
  if (check_desc(desc)) return -1;
  if (!h3_supported(*desc, 22)) return 0;
  return h3_packed_bytes(*desc);
}

int tw_flow_path_supported(const tw_flow_desc* desc, int32_t n_atoms, int32_t path) {
  if (check_desc(desc) || n_atoms <= 0) return 0;
  switch (path) {
    case TW_PATH_AUTO:
    case TW_PATH_SIMPLE:
    case TW_PATH_SIMPLE_H3: return 1;
    case TW_PATH_FUSED: return fused_supported(*desc, n_atoms) ? 1 : 0;
    case TW_PATH_FUSED_H3: return h3_supported(*desc, n_atoms) ? 1 : 0;
    case TW_PATH_FUSED_H1: return h1_supported(*desc, n_atoms) ? 1 : 0;
    default: return 0;
  }
}

int tw_flow_pack_h3(const tw_flow_desc* desc, const float* raw, void* packed_h3, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  TW_REQUIRE(h3_supported(*desc, 22), "split-fp16 path unsupported for this config");
  TW_REQUIRE(raw && packed_h3, "NULL buffer");
  // the last 256 bytes of the over-fetch slack double as scratch for the per-matrix scale search
  float* scratch = (float*)((char*)packed_h3 + h3_packed_bytes(*desc) - 256);
  return h3_pack_weights(*desc, raw, (char*)packed_h3, scratch, (hipStream_t)stream);
}

int64_t tw_flow_packed_simple_h3_bytes(const tw_flow_desc* desc) {
  if (check_desc(desc)) return -1;
  if (!h3_supported(*desc, 22)) return 0;
  return simple_h3_split_offset(*desc) + h3_ffn_split_bytes(*desc);
}

int tw_flow_pack_simple_h3(const tw_flow_desc* desc, const float* raw, void* packed, void* stream) {
  int rc = tw_flow_pack_h3(desc, raw, packed, stream);
  if (rc) return rc;
  if ((rc = simple_h3_fold(*desc, raw, (float*)((char*)packed + (h3_packed_bytes(*desc) + 255) / 256 * 256), (hipStream_t)stream))) return rc;
  return h3_ffn_split_pack(*desc, packed, (char*)packed + simple_h3_split_offset(*desc), (hipStream_t)stream);
}

int64_t tw_flow_packed_h1_bytes(const tw_flow_desc* desc) {
  if (check_desc(desc)) return -1;
  if (!h1_supported(*desc, 22)) return 0;
  return h3_packed_bytes(*desc, true);
}

int tw_flow_pack_h1(const tw_flow_desc* desc, const float* raw, void* packed_h1, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  TW_REQUIRE(h1_supported(*desc, 22), "single-MFMA path unsupported for this config");
  TW_REQUIRE(raw && packed_h1, "NULL buffer");
  float* scratch = (float*)((char*)packed_h1 + h3_packed_bytes(*desc, true) - 256);  // as tw_flow_pack_h3
  return h3_pack_weights(*desc, raw, (char*)packed_h1, scratch, (hipStream_t)stream, true);
}

int64_t tw_flow_workspace_bytes(const tw_flow_desc* desc, int64_t n_rows, int32_t n_atoms) {
  if (check_desc(desc) || n_rows < 0 || n_atoms <= 0) return -1;
  const int64_t rows = n_rows > 0 ? n_rows : 1;
  int64_t a = simple_workspace_bytes(*desc, rows, n_atoms);
  int64_t b = fused_supported(*desc, n_atoms) ? fused_workspace_bytes(*desc, rows, n_atoms) : 0;
  if (h3_supported(*desc, n_atoms)) {
    const int64_t c = h3_workspace_bytes(*desc, rows, n_atoms);
    if (c > b) b = c;
  }
  // the likelihood / sampling entry points carve their own temporaries in front of the flow scratch
  const int64_t extra = 6 * ((rows * n_atoms * 3 * 4 + 255) / 256 * 256) + 4 * ((rows * 4 + 255) / 256 * 256);
  return (a > b ? a : b) + extra;
}

int tw_flow_pass(const tw_flow_desc* desc, const float* raw, const float* packed, const int32_t* atom_types,
                 const float* x_coords, const float* x_velocs, const uint8_t* masked, int64_t n_cond, float* z_coords,
                 float* z_velocs, float* delta_logp, int64_t n_rows, int32_t n_atoms, int32_t reverse, int32_t path,
                 void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  TW_REQUIRE(n_rows >= 0 && n_atoms > 0 && n_cond > 0, "bad sizes");
  TW_REQUIRE(n_rows % n_cond == 0, "n_rows (%lld) must be a multiple of n_cond (%lld)", (long long)n_rows,
             (long long)n_cond);
  if (n_rows == 0) return TW_OK;
  TW_REQUIRE(raw && atom_types && x_coords && x_velocs && masked && z_coords && z_velocs && delta_logp && workspace,
             "NULL pointer argument");
  int p;
  if ((rc = resolve_path(*desc, n_atoms, path, packed, &p))) return rc;
  FlowArgs a{desc, raw, packed, atom_types, x_coords, x_velocs, masked, n_cond, z_coords, z_velocs,
             delta_logp, n_rows, n_atoms, reverse, workspace, workspace_bytes, (hipStream_t)stream};
  if (p == TW_PATH_FUSED_H3 || p == TW_PATH_FUSED_H1) {
    a.h1 = p == TW_PATH_FUSED_H1 ? 1 : 0;
    return flow_pass_h3(a);
  }
  a.simple_h3 = p == TW_PATH_SIMPLE_H3 ? 1 : 0;
  return p == TW_PATH_FUSED ? flow_pass_fused(a) : flow_pass_simple(a);
}

static char* carve(char*& p, int64_t bytes) {
  char* r = p;
  p += (bytes + 255) / 256 * 256;
  return r;
}

int tw_flow_log_likelihood(const tw_flow_desc* desc, const float* raw, const float* packed, const int32_t* atom_types,
                           const float* x_coords, const float* x_velocs, const float* y_coords, const float* y_velocs,
                           const uint8_t* masked, float* out_logp, int64_t n_rows, int32_t n_atoms, int32_t path,
                           void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  TW_REQUIRE(n_rows >= 0 && n_atoms > 0, "bad sizes");
  if (n_rows == 0) return TW_OK;
  TW_REQUIRE(raw && atom_types && x_coords && x_velocs && y_coords && y_velocs && masked && out_logp && workspace,
             "NULL pointer argument");
  hipStream_t s = (hipStream_t)stream;
  const int64_t el = n_rows * n_atoms * 3;
  char* p = (char*)workspace;
  float* xc = (float*)carve(p, el * 4);
  float* zc = (float*)carve(p, el * 4);
  float* zv = (float*)carve(p, el * 4);
  float* xv0 = (float*)carve(p, el * 4);
  float* delta = (float*)carve(p, n_rows * 4);
  const int64_t used = p - (char*)workspace;
  TW_REQUIRE(used < workspace_bytes, "workspace too small");
  // flow.py:148-157: residual target, centred conditioning positions
  if (desc->displacement) {
    if ((rc = launch_sub(y_coords, x_coords, zc, el, s))) return rc;
  } else {
    TW_HIP_CHECK(hipMemcpyAsync(zc, y_coords, el * 4, hipMemcpyDeviceToDevice, s));
  }
  TW_HIP_CHECK(hipMemcpyAsync(zv, y_velocs, el * 4, hipMemcpyDeviceToDevice, s));
  if ((rc = launch_centre(x_coords, masked, xc, nullptr, n_rows, n_atoms, s))) return rc;
  const float* xv = x_velocs;
  if (desc->ignore_cond_velocity) {
    TW_HIP_CHECK(hipMemsetAsync(xv0, 0, el * 4, s));
    xv = xv0;
  }
  TW_HIP_CHECK(hipMemsetAsync(delta, 0, n_rows * 4, s));
  if ((rc = tw_flow_pass(desc, raw, packed, atom_types, xc, xv, masked, n_rows, zc, zv, delta, n_rows, n_atoms, 0, path,
                         p, workspace_bytes - used, stream)))
    return rc;
  const RawLayout L = raw_layout(*desc);
  // flow.py:191-203: log p(y) = log N(z) - delta_logp
  return launch_prior_logp(zc, zv, masked, n_rows, raw + L.prior, delta, -1.f, out_logp, n_rows, n_atoms, s);
}

static int sample_with_logp_impl(const tw_flow_desc* desc, const float* raw, const float* packed, const int32_t* atom_types,
                                 const float* x_coords, const float* x_velocs, const uint8_t* masked, const float* z_coords,
                                 const float* z_velocs, float* y_coords, float* y_velocs, float* out_logp, int64_t n_samples,
                                 int64_t n_cond, int32_t n_atoms, int32_t path, void* workspace, int64_t workspace_bytes,
                                 void* stream, bool multi) {
  int rc = check_desc(desc);
  if (rc) return rc;
  TW_REQUIRE(n_samples >= 0 && n_cond > 0 && n_atoms > 0, "bad sizes");
  TW_REQUIRE(multi || n_cond == 1 || n_samples == 1,
             "the reference's mask broadcast (flow.py:326) needs n_cond == 1 or n_samples == 1");
  const int64_t n_rows = n_samples * n_cond;
  if (n_rows == 0) return TW_OK;
  TW_REQUIRE(raw && atom_types && x_coords && x_velocs && masked && z_coords && z_velocs && y_coords && y_velocs &&
                 out_logp && workspace,
             "NULL pointer argument");
  hipStream_t s = (hipStream_t)stream;
  const int64_t el = n_rows * n_atoms * 3, elc = n_cond * n_atoms * 3;
  char* p = (char*)workspace;
  float* xc = (float*)carve(p, elc * 4);
  float* com = (float*)carve(p, n_cond * 3 * 4);
  float* rc_ = (float*)carve(p, el * 4);
  float* xv0 = (float*)carve(p, elc * 4);
  float* delta = (float*)carve(p, n_rows * 4);
  const int64_t used = p - (char*)workspace;
  TW_REQUIRE(used < workspace_bytes, "workspace too small");
  if ((rc = launch_centre(x_coords, masked, xc, com, n_cond, n_atoms, s))) return rc;
  const float* xv = x_velocs;
  if (desc->ignore_cond_velocity) {
    TW_HIP_CHECK(hipMemsetAsync(xv0, 0, elc * 4, s));
    xv = xv0;
  }
  // the flow updates in place: residual coords go through a scratch copy, velocities straight into y_velocs
  TW_HIP_CHECK(hipMemcpyAsync(rc_, z_coords, el * 4, hipMemcpyDeviceToDevice, s));
  TW_HIP_CHECK(hipMemcpyAsync(y_velocs, z_velocs, el * 4, hipMemcpyDeviceToDevice, s));
  TW_HIP_CHECK(hipMemsetAsync(delta, 0, n_rows * 4, s));
  if ((rc = tw_flow_pass(desc, raw, packed, atom_types, xc, xv, masked, n_cond, rc_, y_velocs, delta, n_rows, n_atoms, 1,
                         path, p, workspace_bytes - used, stream)))
    return rc;
  const RawLayout L = raw_layout(*desc);
  // flow.py:322-334: log p(y|x) = log N(z) + delta_logp, with z the INPUT latents
  if ((rc = launch_prior_logp(z_coords, z_velocs, masked, n_cond, raw + L.prior, delta, +1.f, out_logp, n_rows, n_atoms, s)))
    return rc;
  // flow.py:303-310: y = (x_centred + com) + residual
  return launch_uncentre_add(xc, com, rc_, n_cond, y_coords, n_atoms, desc->displacement, n_rows, s);
}

extern "C" {
int tw_flow_sample_with_logp(const tw_flow_desc* desc, const float* raw, const float* packed, const int32_t* atom_types,
                             const float* x_coords, const float* x_velocs, const uint8_t* masked, const float* z_coords,
                             const float* z_velocs, float* y_coords, float* y_velocs, float* out_logp, int64_t n_samples,
                             int64_t n_cond, int32_t n_atoms, int32_t path, void* workspace, int64_t workspace_bytes,
                             void* stream) {
  return sample_with_logp_impl(desc, raw, packed, atom_types, x_coords, x_velocs, masked, z_coords, z_velocs, y_coords,
                               y_velocs, out_logp, n_samples, n_cond, n_atoms, path, workspace, workspace_bytes, stream, false);
}

int tw_flow_sample_with_logp_multi(const tw_flow_desc* desc, const float* raw, const float* packed,
                                   const int32_t* atom_types, const float* x_coords, const float* x_velocs,
                                   const uint8_t* masked, const float* z_coords, const float* z_velocs, float* y_coords,
                                   float* y_velocs, float* out_logp, int64_t n_samples, int64_t n_cond, int32_t n_atoms,
                                   int32_t path, void* workspace, int64_t workspace_bytes, void* stream) {
  return sample_with_logp_impl(desc, raw, packed, atom_types, x_coords, x_velocs, masked, z_coords, z_velocs, y_coords,
                               y_velocs, out_logp, n_samples, n_cond, n_atoms, path, workspace, workspace_bytes, stream, true);
}
}  // extern "C"

int tw_kernel_scores(const float* x_coords, const uint8_t* masked, const float* lengthscales, int32_t n_heads,
                     int64_t n_cond, int32_t n_atoms, int32_t normalise, int32_t use_mm, float* out, void* stream) {
  TW_REQUIRE(x_coords && masked && lengthscales && out, "NULL pointer argument");
  TW_REQUIRE(n_heads > 0 && n_cond >= 0 && n_atoms > 0, "bad sizes");
  return launch_scores(x_coords, masked, lengthscales, n_heads, n_cond, n_atoms, normalise, use_mm, out,
                       (hipStream_t)stream);  // (molecules whose V x V tile exceeds the CU's LDS take the row-wise kernel)
}

int tw_kernel_scores_cheb(const float* x_coords, const uint8_t* masked, const float* lengthscales, const float* cheb_coeffs,
                          int32_t cheb_order, int32_t force_zero, int32_t n_heads, int64_t n_cond, int32_t n_atoms,
                          int32_t normalise, int32_t use_mm, float* out, void* stream) {
  TW_REQUIRE(x_coords && masked && lengthscales && cheb_coeffs && out, "NULL pointer argument");
  TW_REQUIRE(n_heads > 0 && n_cond >= 0 && n_atoms > 0 && cheb_order >= 1, "bad sizes");
  return launch_scores(x_coords, masked, lengthscales, n_heads, n_cond, n_atoms, normalise, use_mm, out,
                       (hipStream_t)stream, cheb_coeffs, cheb_order, force_zero);
}

int tw_centre(const float* x_coords, const uint8_t* masked, float* out_centred, float* out_com, int64_t n_rows,
              int32_t n_atoms, void* stream) {
  TW_REQUIRE(x_coords && masked, "NULL pointer argument");
  return launch_centre(x_coords, masked, out_centred, out_com, n_rows, n_atoms, (hipStream_t)stream);
}

int tw_kinetic_energy(const float* velocs, const float* masses, int32_t random_velocs, float kbT, float* out,
                      int64_t n_rows, int32_t n_atoms, void* stream) {
  TW_REQUIRE(velocs && out && (random_velocs || masses), "NULL pointer argument");
  TW_REQUIRE(random_velocs || kbT > 0.f, "kbT required unless random_velocs");
  return launch_kinetic(velocs, masses, random_velocs, kbT, out, n_rows, n_atoms, (hipStream_t)stream);
}

int tw_amber_energy(const tw_forcefield* ff, const float* coords, double* out_energy, double* out_terms, int64_t n_rows,
                    void* stream) {
  TW_REQUIRE(ff && coords && out_energy, "NULL pointer argument");
  TW_REQUIRE(ff->n_atoms > 0, "n_atoms must be positive");  // upper bound: the kernels' LDS check (~1000 atoms)
  return amber_energy(ff, coords, out_energy, out_terms, n_rows, (hipStream_t)stream);
}

int tw_amber_energy_forces(const tw_forcefield* ff, const float* coords, double* out_energy, double* out_forces,
                           int64_t n_rows, void* stream) {
  TW_REQUIRE(ff && coords && out_forces, "NULL pointer argument");
  TW_REQUIRE(ff->n_atoms > 0, "n_atoms must be positive");  // upper bound: the kernels' LDS check (~1000 atoms)
  return amber_energy_forces(ff, coords, out_energy, out_forces, n_rows, (hipStream_t)stream);
}

int tw_langevin_steps(const tw_forcefield* ff, const float* masses, float* coords, float* velocs, int32_t n_steps,
                      double timestep_ps, double friction_per_ps, double kbT, int32_t scheme, uint64_t seed, int64_t first_step,
                      double* out_energy, int64_t n_rows, void* stream) {
  TW_REQUIRE(ff && masses && coords && velocs, "NULL pointer argument");
  TW_REQUIRE(ff->n_atoms > 0, "n_atoms must be positive");  // upper bound: the kernels' LDS check (~1000 atoms)
  TW_REQUIRE(n_steps >= 0 && timestep_ps > 0.0 && friction_per_ps >= 0.0 && kbT >= 0.0 && (scheme == 0 || scheme == 1),
             "bad integrator parameters");
  return langevin_steps(ff, masses, coords, velocs, n_steps, timestep_ps, friction_per_ps, kbT, scheme, seed, first_step,
                        out_energy, n_rows, (hipStream_t)stream);
}

int tw_mh_accept(const float* energy, const float* p_xy, const float* p_yx, const float* u, const float* y_coords,
                 const float* y_velocs, float* x_coords, float* x_velocs, float* out_exponent, float* out_p_acc,
                 uint8_t* out_accepted, int32_t* result, int64_t n_proposals, int32_t n_atoms, void* stream) {
  TW_REQUIRE(energy && p_xy && p_yx && u && out_exponent && out_p_acc && out_accepted && result, "NULL pointer argument");
  TW_REQUIRE((x_coords == nullptr) == (x_velocs == nullptr), "x_coords and x_velocs must both be given or both be NULL");
  TW_REQUIRE(!x_coords || (y_coords && y_velocs), "state update needs y_coords / y_velocs");
  TW_REQUIRE(n_proposals > 0 && n_proposals < (1LL << 30), "bad n_proposals");
  return launch_mh_accept(energy, p_xy, p_yx, u, y_coords, y_velocs, x_coords, x_velocs, out_exponent, out_p_acc,
                          out_accepted, result, n_proposals, n_atoms, (hipStream_t)stream);
}

int tw_mh_accept_chains(const float* energy, const float* p_xy, const float* p_yx, const float* u, const float* y_coords,
                        const float* y_velocs, float* x_coords, float* x_velocs, float* out_exponent, float* out_p_acc,
                        uint8_t* out_accepted, int32_t* result, int64_t n_proposals, int64_t n_chains, int32_t n_atoms,
                        void* stream) {
  TW_REQUIRE(energy && p_xy && p_yx && u && out_exponent && out_p_acc && out_accepted && result, "NULL pointer argument");
  TW_REQUIRE((x_coords == nullptr) == (x_velocs == nullptr), "x_coords and x_velocs must both be given or both be NULL");
  TW_REQUIRE(!x_coords || (y_coords && y_velocs), "state update needs y_coords / y_velocs");
  TW_REQUIRE(n_proposals > 0 && n_proposals < (1LL << 30) && n_chains > 0 && n_chains < (1LL << 20), "bad sizes");
  return launch_mh_accept(energy, p_xy, p_yx, u, y_coords, y_velocs, x_coords, x_velocs, out_exponent, out_p_acc,
                          out_accepted, result, n_proposals, n_atoms, (hipStream_t)stream, n_chains);
}

int tw_chirality_changed(const float* coords, const int32_t* centres, const float* reference_signs, int32_t n_centres,
                         uint8_t* out_changed, int64_t n_rows, int32_t n_atoms, void* stream) {
  TW_REQUIRE(coords && out_changed && (n_centres == 0 || (centres && reference_signs)), "NULL pointer argument");
  return launch_chirality(coords, centres, reference_signs, n_centres, out_changed, n_rows, n_atoms,
                          (hipStream_t)stream);
}

const char* tw_last_netblock_kernel(void) { return last_netblock_kernel(); }

const char* tw_flow_selected_kernel(const tw_flow_desc* desc, int32_t n_atoms, int64_t n_rows, int32_t path) {
  if (check_desc(desc) || n_atoms <= 0 || n_rows <= 0) return "";
  const bool h1 = path == TW_PATH_FUSED_H1;
  if (!(path == TW_PATH_FUSED_H3 || h1)) return "";
  if (!(h1 ? h1_supported(*desc, n_atoms) : h3_supported(*desc, n_atoms))) return "";
  const char* before = last_netblock_kernel();
  note_netblock_kernel("");
  const int rc = h3_selected_kernel(*desc, n_atoms, n_rows, h1);
  const char* name = rc == TW_OK ? last_netblock_kernel() : "";
  note_netblock_kernel(before);
  return name;
}

int tw_flow_nonfinite(int32_t reset, int32_t* out_flag) {
  TW_REQUIRE(out_flag != nullptr, "NULL pointer argument");
  int v = 0;
  int rc = nonfinite_flag(reset, &v);
  *out_flag = v;
  return rc;
}

int tw_debug_set_flags(int flags) {
#ifndef TW_EXPERIMENTS
  TW_REQUIRE((flags & TW_WRONG_RESULT_BITS) == 0,
             "tw_debug_set_flags: bits %d are timing experiments that make results wrong; they exist only in a "
             "-DTW_EXPERIMENTS build of the library", flags & TW_WRONG_RESULT_BITS);
#endif
  tw::g_debug_flags.store(flags, std::memory_order_relaxed);
  return TW_OK;
}

// ---- tw_probe_mfma_clock: the bare MFMA stream of tools/probe/mfma_stream_probe.hip as a measurement hook
namespace {
__global__ void __launch_bounds__(256) mfma_stream_kernel(long long* out, int iters, unsigned seed) {
  // operands: A = v[0:7] (two tiles), B = v[8:31] (six tiles): fp16 pairs of magnitude ~1 with random mantissas
  unsigned r[32];
  unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    s = s * 1664525u + 1013904223u;
    r[i] = (0x3800u + ((s >> 8) & 0x7ffu)) | ((0xb800u + ((s >> 20) & 0x7ffu)) << 16);
  }
  unsigned long long t0 = 0, t1 = 0;
  asm volatile(
      "v_mov_b32 v0, %[r0]\n\tv_mov_b32 v1, %[r1]\n\tv_mov_b32 v2, %[r2]\n\tv_mov_b32 v3, %[r3]\n\tv_mov_b32 v4, %[r4]\n\tv_mov_b32 v5, %[r5]\n\tv_mov_b32 v6, %[r6]\n\tv_mov_b32 v7, %[r7]\n\t"
      "v_mov_b32 v8, %[r8]\n\tv_mov_b32 v9, %[r9]\n\tv_mov_b32 v10, %[r10]\n\tv_mov_b32 v11, %[r11]\n\tv_mov_b32 v12, %[r12]\n\tv_mov_b32 v13, %[r13]\n\tv_mov_b32 v14, %[r14]\n\tv_mov_b32 v15, %[r15]\n\t"
      "v_mov_b32 v16, %[r16]\n\tv_mov_b32 v17, %[r17]\n\tv_mov_b32 v18, %[r18]\n\tv_mov_b32 v19, %[r19]\n\tv_mov_b32 v20, %[r20]\n\tv_mov_b32 v21, %[r21]\n\tv_mov_b32 v22, %[r22]\n\tv_mov_b32 v23, %[r23]\n\t"
      "v_mov_b32 v24, %[r24]\n\tv_mov_b32 v25, %[r25]\n\tv_mov_b32 v26, %[r26]\n\tv_mov_b32 v27, %[r27]\n\tv_mov_b32 v28, %[r28]\n\tv_mov_b32 v29, %[r29]\n\tv_mov_b32 v30, %[r30]\n\tv_mov_b32 v31, %[r31]\n\t"
      ".set i, 0\n\t.rept 12\n\tv_accvgpr_write_b32 a[i], 0\n\t.set i, i + 1\n\t.endr\n\t"
      "s_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
      ".Ltw_probe_loop_%=:\n\t"
      ".rept 4\n\t"
      "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[8:11], a[0:3]\n\t"
      "v_mfma_f32_16x16x32_f16 a[4:7], v[0:3], v[16:19], a[4:7]\n\t"
      "v_mfma_f32_16x16x32_f16 a[8:11], v[0:3], v[24:27], a[8:11]\n\t"
      "v_mfma_f32_16x16x32_f16 a[0:3], v[0:3], v[12:15], a[0:3]\n\t"
      "v_mfma_f32_16x16x32_f16 a[4:7], v[0:3], v[20:23], a[4:7]\n\t"
      "v_mfma_f32_16x16x32_f16 a[8:11], v[0:3], v[28:31], a[8:11]\n\t"
      "v_mfma_f32_16x16x32_f16 a[0:3], v[4:7], v[8:11], a[0:3]\n\t"
      "v_mfma_f32_16x16x32_f16 a[4:7], v[4:7], v[16:19], a[4:7]\n\t"
      "v_mfma_f32_16x16x32_f16 a[8:11], v[4:7], v[24:27], a[8:11]\n\t"
      ".endr\n\t"
      "s_sub_u32 %[it], %[it], 1\n\t"
      "s_cmp_lg_u32 %[it], 0\n\t"
      "s_cbranch_scc1 .Ltw_probe_loop_%=\n\t"
      "s_nop 15\n\t"
      "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
      : [t0] "=&s"(t0), [t1] "=&s"(t1), [it] "+s"(iters)
      : [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [r3] "v"(r[3]), [r4] "v"(r[4]), [r5] "v"(r[5]), [r6] "v"(r[6]), [r7] "v"(r[7]), [r8] "v"(r[8]), [r9] "v"(r[9]), [r10] "v"(r[10]), [r11] "v"(r[11]), [r12] "v"(r[12]), [r13] "v"(r[13]), [r14] "v"(r[14]), [r15] "v"(r[15]), [r16] "v"(r[16]), [r17] "v"(r[17]), [r18] "v"(r[18]), [r19] "v"(r[19]), [r20] "v"(r[20]), [r21] "v"(r[21]), [r22] "v"(r[22]), [r23] "v"(r[23]), [r24] "v"(r[24]), [r25] "v"(r[25]), [r26] "v"(r[26]), [r27] "v"(r[27]), [r28] "v"(r[28]), [r29] "v"(r[29]), [r30] "v"(r[30]), [r31] "v"(r[31])
      : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",
        "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "a0", "a1", "a2", "a3", "a4",
        "a5", "a6", "a7", "a8", "a9", "a10", "a11", "scc");
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (long long)(t1 - t0);
}
}  // namespace

int tw_probe_mfma_clock(int32_t workgroups, int32_t iters, int64_t* cycles, double* ms, void* stream) {
  TW_REQUIRE(workgroups > 0 && workgroups <= 65536 && iters > 0 && cycles && ms, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  long long* dev = nullptr;
  TW_HIP_CHECK(hipMalloc(&dev, sizeof(long long)));
  hipEvent_t e0, e1;
  TW_HIP_CHECK(hipEventCreate(&e0));
  TW_HIP_CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_stream_kernel, dim3((unsigned)workgroups), dim3(256), 0, st, dev, iters, 12345u);  // clock ramp
  TW_HIP_CHECK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(mfma_stream_kernel, dim3((unsigned)workgroups), dim3(256), 0, st, dev, iters, 54321u);
  TW_HIP_CHECK(hipEventRecord(e1, st));
  TW_HIP_CHECK(hipEventSynchronize(e1));
  float t = 0.f;
  TW_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
  long long c = 0;
  TW_HIP_CHECK(hipMemcpy(&c, dev, sizeof(c), hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(dev);
  *cycles = (int64_t)c;
  *ms = (double)t;
  return TW_OK;
}

int tw_profile_begin(void) { return profile_begin(); }
int tw_profile_end(double* total_ms, int64_t* launches) { return profile_end(total_ms, launches); }

int tw_debug_netblock(const tw_flow_desc* desc, const float* raw, const float* packed, int32_t coupling, int32_t net,
                      const int32_t* atom_types, const float* x_coords, const float* x_velocs, const uint8_t* masked,
                      int64_t n_cond, const float* z_other, int64_t n_rows, int32_t n_atoms, int32_t path, float* dump,
                      void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  TW_REQUIRE(raw && atom_types && x_coords && x_velocs && masked && z_other && dump && workspace, "NULL pointer argument");
  TW_REQUIRE(coupling >= 0 && coupling < desc->n_coupling && (net == 0 || net == 1), "bad coupling/net index");
  TW_REQUIRE(n_rows > 0 && n_cond > 0 && n_rows % n_cond == 0, "bad sizes");
  int p;
  if ((rc = resolve_path(*desc, n_atoms, path, packed, &p))) return rc;
  FlowArgs a{desc, raw, packed, atom_types, x_coords, x_velocs, masked, n_cond, nullptr, nullptr,
             nullptr, n_rows, n_atoms, 0, workspace, workspace_bytes, (hipStream_t)stream};
  if (p == TW_PATH_FUSED_H3 || p == TW_PATH_FUSED_H1) {
    a.h1 = p == TW_PATH_FUSED_H1 ? 1 : 0;
    return debug_netblock_h3(a, coupling, net, z_other, dump);
  }
  a.simple_h3 = p == TW_PATH_SIMPLE_H3 ? 1 : 0;
  return p == TW_PATH_FUSED ? debug_netblock_fused(a, coupling, net, z_other, dump)
                            : debug_netblock_simple(a, coupling, net, z_other, dump);
}

}  // extern "C"
