/*
 * timewarp_hip.h -- C ABI of libtimewarp_hip.so: the MI355X (gfx950) implementation of
 * Timewarp's conditional-flow sampling hot path.
 *
 * The reference (microsoft/timewarp) is pure Python/PyTorch: it has no FFI of its own, so the
 * "binding" a maintainer adds is a ctypes stub (INTEGRATION.md shows it).  Each entry point below
 * names the reference function (file:line under the reference root) whose torch-op chain it
 * replaces.  Conventions:
 *
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - all float tensors are contiguous fp32, index tensors int32, masks uint8 (1 = masked);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing synchronises the host;
 *   - return value: 0 on success, a negative tw_status otherwise; tw_last_error() gives the
 *     message of the last failure on the calling thread.  No entry point allocates or frees
 *     caller memory; workspaces are caller-provided (tw_flow_workspace_bytes);
 *   - thread-safety: the compute entry points are re-entrant across streams.  Process-wide state: the per-thread error
 *     string; the debug / measurement word of tw_debug_set_flags (one atomic int, read once per launch: set it while no
 *     other thread launches); the profile hooks (tw_profile_begin / tw_profile_end, not thread-safe); the sticky
 *     per-device non-finite flag of tw_flow_nonfinite (used by descriptors whose range_flag is NULL; since ABI 7 a
 *     descriptor can carry its caller's own word instead); tw_mh_iteration keeps one helper stream and two events per device
 *     and calling thread, created on first use.
 */
#ifndef TIMEWARP_HIP_H
#define TIMEWARP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TW_ABI_VERSION 8

typedef enum {
  TW_OK = 0,
  TW_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  TW_ERR_HIP = -2,       /* a HIP runtime call failed */
  TW_ERR_WORKSPACE = -3, /* workspace too small */
  TW_ERR_NO_DEVICE = -4  /* no gfx950 device visible */
} tw_status;

/* Hyper-parameters of one flow (model_configs.py:51-76, custom_attention_encoder.py:126-137,
 * transformer_block.py:11-15).  kernel variant: n_heads = len(lengthscales), value_dim = d_model. */
typedef struct {
  int32_t variant;      /* 0 = kernel (custom_attention_transformer_nvp), 1 = dense (transformer_nvp) */
  int32_t n_coupling;   /* 8 */
  int32_t n_layers;     /* 3 encoder layers per net */
  int32_t d_model;      /* 128 */
  int32_t d_ff;         /* 2048 */
  int32_t d_hidden;     /* 256: the single hidden layer of in_mlp / out_mlp */
  int32_t d_emb;        /* 32 */
  int32_t n_heads;      /* 6 (kernel) / 8 (dense) */
  int32_t d_rff;        /* dense only: RFF encoding dims (0 in transformer_nvp.yaml) */
  int32_t n_elements;   /* 5 = len(ELEMENT_VOCAB) */
  int32_t pos_mod2;     /* position_layer_index_mod_2 */
  int32_t displacement; /* use_displacement_as_target */
  int32_t ignore_cond_velocity;
  int32_t normalise;    /* normalise_kernel_values */
  float ln_eps;         /* 1e-5 */
  int32_t cheb_order;   /* kernel variant: 0 = Gaussian basis exp(-s^2) (attention_type "kernel"/"learnable_kernel");
                           > 0 = attention_type "chebyshev_kernel": rational Chebyshev expansion of that order of s^2
                           with per-layer coefficients (kernel_attention.py:12-66, 255-339); all execution paths (the
                           fused kernels take one score-fragment set per (net, layer) of the coupling layer in flight) */
  int32_t cheb_force_zero; /* force_asymptotic_zero: subtract each head's mean coefficient */
  int32_t* range_flag;  /* ABI 7 - split-fp16 range guard, per model / per call: DEVICE pointer to a caller-owned int32
                           (zeroed by the caller) that the flow kernels OR to 1 when a coupling net returns a non-finite
                           scale or shift; every entry point that takes this descriptor reports there.  NULL: the
                           per-device word of tw_flow_nonfinite (two models on one device then share one flag). */
} tw_flow_desc;

const char* tw_last_error(void);
int tw_abi_version(void);
/* Number of gfx950 devices visible (0 on a CPU-only box; never fails). */
int tw_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Weights.  The host packs the reference state_dict (SURVEY.md section 8b names) into ONE flat fp32
 * device buffer in the canonical order documented in DESIGN.md ("raw layout"):
 *   embedding[n_elements,d_emb], lengthscales[2,n_heads] (kernel; row 0 is used by forward passes, row 1 by
 *   reverse passes - equal unless the lengthscales are learnable, see timewarp_amd/weights.py), prior log-scales[2],
 *   then for c in coupling layers, net in (scale, shift):  [dense: rff vectors[3,d_rff/2] once per c]
 *     in_mlp.0.{w[d_hidden,d_in],b}, in_mlp.2.{w[d_model,d_hidden],b},
 *     per layer: kernel: values_proj.w[H*d_model,d_model], out_projection.w[d_model,H*d_model],
 *                        [cheb_coeffs[H,cheb_order] when cheb_order > 0]
 *                dense : in_proj.{w[3d,d],b[3d]}, out_proj.{w[d,d],b[d]}
 *                linear1.{w,b}, linear2.{w,b}, norm1.{w,b}, norm2.{w,b}
 *     out_mlp.0.{w[d_hidden,d_model],b}, out_mlp.2.{w[3,d_hidden],b}
 * tw_flow_raw_floats returns the total so the host can check its packing.
 * ------------------------------------------------------------------------------------------- */
int64_t tw_flow_raw_floats(const tw_flow_desc* desc);
/* Size (floats) of the MFMA-fragment-ordered weight stream used by the fused kernel path
 * (kernel variant only; 0 for dense). */
int64_t tw_flow_packed_floats(const tw_flow_desc* desc);
/* Builds the fused-path weight stream from the raw buffer on the device (folds
 * W_o[:,h] @ W_v[h] per head in fp64, re-tiles every matrix into 16x16 MFMA A-fragments). */
int tw_flow_pack(const tw_flow_desc* desc, const float* raw, float* packed, void* stream);

/* The split-fp16 weight stream of TW_PATH_FUSED_H3: size in BYTES (0 if unsupported) and builder
 * (per-matrix power-of-two scaling, fp16 hi/lo tile pairs in LDS-DMA stage order). */
int64_t tw_flow_packed_h3_bytes(const tw_flow_desc* desc);
int tw_flow_pack_h3(const tw_flow_desc* desc, const float* raw, void* packed_h3, void* stream);
/* ABI 8 - the pack of TW_PATH_SIMPLE_H3 (0 bytes where the model has no split-fp16 stream: that path then takes packed = NULL):
 * the tw_flow_pack_h3 stream (its FFN stages feed the path's fused FFN launches), then - kernel attention - the per-head folded
 * value / output projections Wc[coupling][net][layer][d_model][n_heads d_model] (fp32, folded in fp64) and their split-fp16 copies
 * per head in MFMA-operand order, which let the mixing run on the layer input with the ONE remaining GEMM inside the same launch; then
 * the FFN stages of every (coupling, net, layer) once more as four self-contained streams of a quarter of the hidden layer each (small
 * launches spread an FFN over four workgroups per token tile). */
int64_t tw_flow_packed_simple_h3_bytes(const tw_flow_desc* desc);
int tw_flow_pack_simple_h3(const tw_flow_desc* desc, const float* raw, void* packed, void* stream);
/* The same for TW_PATH_FUSED_H1 (fp16 hi tiles only: 8 tiles per 9 KiB stage, half as many stages). */
int64_t tw_flow_packed_h1_bytes(const tw_flow_desc* desc);
int tw_flow_pack_h1(const tw_flow_desc* desc, const float* raw, void* packed_h1, void* stream);

/* Bytes of scratch the flow entry points need for n_rows conformations of n_atoms atoms.  (The per-op paths run the two nets of a
 * coupling layer side by side - the second on a library-owned side stream that waits for and is waited for by `stream` through
 * events, so the caller sees ordinary stream order - and size the scratch for two sets of activations.) */
int64_t tw_flow_workspace_bytes(const tw_flow_desc* desc, int64_t n_rows, int32_t n_atoms);

/* Execution path selector for the flow entry points. */
#define TW_PATH_AUTO 0   /* fused MFMA path where supported, else simple */
#define TW_PATH_FUSED 1  /* fused f32-MFMA net-block kernel (kernel variant, n_atoms <= 64) */
#define TW_PATH_SIMPLE 2 /* one plain HIP kernel per reference op (all variants) */
#define TW_PATH_FUSED_H3 3 /* fused split-fp16 kernel: every fp32 product as 3 half-precision MFMAs with fp32
                              accumulation (2^-22 operand representation); needs |activations| < 65504.  Kernel
                              attention: n_atoms <= 48 in 48-token waves holding floor(48 / n_atoms) molecules, and
                              25 .. 192 atoms in the "wide" layout - floor(192 / n_atoms) molecules packed over a workgroup's
                              four waves (81 .. 95 atoms: two, at a slot stride of 96) - chosen per launch where both exist (fewer rounds of
                              the chip); dense softmax attention: n_atoms <= 48.  tw_flow_path_supported answers per size.
                              `packed` must then point at the tw_flow_pack_h3 stream.  Never chosen by
                              TW_PATH_AUTO. */
#define TW_PATH_FUSED_H1 4 /* "fast" mode, NOT a parity path: the same fused kernel with ONE half-precision MFMA per
                              product (fp16 operands, 11 significand bits, fp32 accumulation) and half the weight stream.
                              Kernel attention, every molecule size TW_PATH_FUSED_H3 takes (48-token waves and the wide
                              layout); the dense softmax model up to 48 atoms too (its in / FFN / out
                              sections - 88 % of the work - run single-MFMA, the softmax attention block stays in
                              split form; with position features the in-MLP as well).  Results deviate from the reference's fp32 arithmetic by
                              ~1e-4 relative (measured per case in tests/test_flow_h1_gpu.py); proposal and reverse-move
                              densities of an MH iteration are evaluated by the same arithmetic.  `packed` must point at
                              the tw_flow_pack_h1 stream.  Only ever chosen by name. */
#define TW_PATH_SIMPLE_H3 5 /* ABI 8: TW_PATH_SIMPLE with its linear layers (in / out MLPs, value and output projections, q / k / v,
                              FFN) as split-fp16 MFMA GEMMs - the arithmetic of TW_PATH_FUSED_H3 (3 half-precision MFMAs per fp32
                              product, fp32 accumulation), one kernel per reference op, ANY molecule size and both model variants:
                              what serves the sizes no fused layout takes (kernel attention above 192 atoms, dense above 64) at 2-4x
                              the rate of the exact-f32 per-op kernels.  Scores, softmax, LayerNorm stay fp32.  Needs
                              |activations| < 65504 and |weights| < 256 (else non-finite outputs -> range flag).  `packed`: NULL, or
                              the tw_flow_pack_simple_h3 buffer where that exists (d_model 128) - the FFN of every encoder layer
                              and the in / out MLPs then run as ONE launch each of the fused kernels' statements on the flat token
                              list (hidden layer on chip) instead of two GEMMs through HBM, and kernel attention mixes the layer
                              input itself with the one folded 768 -> 128 GEMM inside the mixing launch. */

/* 1 if `path` can run this configuration on molecules of n_atoms atoms (TW_PATH_AUTO / TW_PATH_SIMPLE / TW_PATH_SIMPLE_H3: always), else 0.
 * What a caller asks before it requests TW_PATH_FUSED / TW_PATH_FUSED_H3 by name (those fail with TW_ERR_INVALID on an
 * unsupported shape instead of falling back). */
int tw_flow_path_supported(const tw_flow_desc* desc, int32_t n_atoms, int32_t path);

/* ConditionalSequentialFlow.forward (modules/model_wrappers/flow.py:51-103) over
 * NVPCouplingLayer.forward (modules/layers/nvp.py:22-183) with
 * CustomAttentionTransformerCouplingLayer._get_scale_and_shift (modules/custom_transformer_nvp.py:44-93)
 * or TransformerCouplingLayer (modules/transformer_nvp.py:58-97).
 *   x_coords/x_velocs/atom_types/masked: [n_cond, n_atoms, ...] conditioning rows; row n of z uses
 *   conditioning row n % n_cond (the reference's .repeat(S,1,1) tiling, flow.py:284-296).
 *   z_coords/z_velocs [n_rows,n_atoms,3] and delta_logp [n_rows] are updated in place.
 *   x_coords must already be centred (flow.py:156-157 / 261-262).  `packed` may be NULL for
 *   TW_PATH_SIMPLE. */
int tw_flow_pass(const tw_flow_desc* desc, const float* raw, const float* packed,
                 const int32_t* atom_types, const float* x_coords, const float* x_velocs,
                 const uint8_t* masked, int64_t n_cond, float* z_coords, float* z_velocs,
                 float* delta_logp, int64_t n_rows, int32_t n_atoms, int32_t reverse, int32_t path,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* ConditionalFlowDensityModel.log_likelihood (modules/model_wrappers/flow.py:131-215).
 * All inputs [n_rows,n_atoms,(3)]; out_logp [n_rows]. */
int tw_flow_log_likelihood(const tw_flow_desc* desc, const float* raw, const float* packed,
                           const int32_t* atom_types, const float* x_coords, const float* x_velocs,
                           const float* y_coords, const float* y_velocs, const uint8_t* masked,
                           float* out_logp, int64_t n_rows, int32_t n_atoms, int32_t path,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* ConditionalFlowDensityModel.conditional_sample_with_logp (flow.py:242-336) with the latent
 * noise passed in: z_* [n_samples,n_cond,n_atoms,3] ALREADY multiplied by exp(prior log-scale)
 * (the reference draws Normal(0,scale).rsample((S,)), coords first, flow.py:274-275).
 * Outputs y_* [n_samples,n_cond,n_atoms,3], out_logp [n_samples,n_cond].  The reference's mask
 * broadcast (flow.py:326) only works for n_cond == 1 or n_samples == 1; same restriction here. */
int tw_flow_sample_with_logp(const tw_flow_desc* desc, const float* raw, const float* packed,
                             const int32_t* atom_types, const float* x_coords,
                             const float* x_velocs, const uint8_t* masked, const float* z_coords,
                             const float* z_velocs, float* y_coords, float* y_velocs,
                             float* out_logp, int64_t n_samples, int64_t n_cond, int32_t n_atoms,
                             int32_t path, void* workspace, int64_t workspace_bytes, void* stream);

/* Extension: the same without the reference's n_cond == 1 || n_samples == 1 restriction (flow.py:326 broadcasts a
 * [B,V,1] mask into [S*B,V,3]); rows are ordered sample-major, row = sample * n_cond + cond, as the reference's
 * reshape would produce.  Used to evaluate several chains' proposals in one launch. */
int tw_flow_sample_with_logp_multi(const tw_flow_desc* desc, const float* raw, const float* packed,
                                   const int32_t* atom_types, const float* x_coords,
                                   const float* x_velocs, const uint8_t* masked,
                                   const float* z_coords, const float* z_velocs, float* y_coords,
                                   float* y_velocs, float* out_logp, int64_t n_samples, int64_t n_cond,
                                   int32_t n_atoms, int32_t path, void* workspace,
                                   int64_t workspace_bytes, void* stream);

/* compute_kernel_attention_scores (modules/layers/kernel_attention.py:69-121):
 * out [n_cond, n_heads, n_atoms, n_atoms].  use_mm != 0 selects torch.cdist's matmul
 * formulation (the reference gets it for n_atoms > 25). */
int tw_kernel_scores(const float* x_coords, const uint8_t* masked, const float* lengthscales,
                     int32_t n_heads, int64_t n_cond, int32_t n_atoms, int32_t normalise,
                     int32_t use_mm, float* out, void* stream);

/* Same with the learnable rational-Chebyshev basis (chebyshev_basis_function / chebyshev_expansion,
 * kernel_attention.py:12-66):  score = sum_c coeff[h,c] R_c(s^2),  R_0 = 1, R_1 = (x-1)/(x+1),
 * R_{n+1} = 2 R_1 R_n - R_{n-1};  cheb_coeffs [n_heads, cheb_order];  force_zero subtracts each head's mean
 * coefficient (force_asymptotic_zero). */
int tw_kernel_scores_cheb(const float* x_coords, const uint8_t* masked, const float* lengthscales,
                          const float* cheb_coeffs, int32_t cheb_order, int32_t force_zero, int32_t n_heads,
                          int64_t n_cond, int32_t n_atoms, int32_t normalise, int32_t use_mm, float* out,
                          void* stream);

/* get_centre_of_mass (utils/molecule_utils.py:15-29): out_centred = x - masked mean,
 * out_com [n_rows,3] (either output may be NULL). */
int tw_centre(const float* x_coords, const uint8_t* masked, float* out_centred, float* out_com,
              int64_t n_rows, int32_t n_atoms, void* stream);

/* compute_kinetic_energy (utils/evaluation_utils.py:416-436):
 * random_velocs ? 0.5*sum v^2 : 0.5*sum m v^2 / kbT.  masses [n_atoms]; out [n_rows]. */
int tw_kinetic_energy(const float* velocs, const float* masses, int32_t random_velocs, float kbT,
                      float* out, int64_t n_rows, int32_t n_atoms, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Potential energy: replaces OpenmmPotentialEnergyTorch.forward
 * (utils/openmm/openmm_bridge.py:281-294 -> bgflow -> OpenMM Context.getState(getEnergy)) for a
 * System built by simulation/md.py:128-187 (HarmonicBond, HarmonicAngle, PeriodicTorsion,
 * Nonbonded CutoffNonPeriodic with reaction field, GBSAOBC).  Parameter tables are what
 * system.getForces() exposes (INTEGRATION.md has the extractor).  Units nm / kJ/mol / e.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_atoms;
  int32_t n_bonds;      /* bond_idx [n_bonds,2] int32,  bond_par [n_bonds,2] = (r0, k)          */
  int32_t n_angles;     /* angle_idx [n_angles,3],      angle_par [n_angles,2] = (theta0, k)    */
  int32_t n_torsions;   /* torsion_idx [n_torsions,4],  torsion_par [n_torsions,3] = (n, phase, k) */
  int32_t n_exceptions; /* exc_idx [n_exceptions,2],    exc_par [n_exceptions,3] = (qq, sigma, eps); 1-2/1-3 have zeros */
  int32_t has_gbsa;     /* 0: no implicit solvent; 1: GBSA-OBC II (GBSAOBCForce, amber99_obc.xml); 2: GBSA-OBC I (obc1.xml) */
  double cutoff;        /* nonbondedCutoff (nm); <= 0 means NoCutoff */
  double rf_dielectric; /* reaction-field dielectric (78.3 default; OpenMM uses 1.0 when GBSA is present) */
  double solute_dielectric, solvent_dielectric; /* GBSA: 1.0 / 78.5 */
  double surface_area_energy;                   /* GBSA ACE term coefficient, 2.25936 kJ/mol/nm^2 */
  const int32_t* bond_idx;    const double* bond_par;
  const int32_t* angle_idx;   const double* angle_par;
  const int32_t* torsion_idx; const double* torsion_par;
  const int32_t* exc_idx;     const double* exc_par;
  const double* atom_par;     /* [n_atoms,5] = (charge, sigma, epsilon, gb_radius, gb_scale) */
} tw_forcefield;

/* coords [n_rows,n_atoms,3] fp32 (nm) -> out_energy [n_rows] fp64 kJ/mol (fp64 like the bridge's
 * numpy round trip, openmm_bridge.py:206-221).  All tw_forcefield pointers are device pointers.
 * out_terms, if not NULL, receives [n_rows,5] = bond, angle, torsion, nonbonded, gbsa
 * (the decomposition of simulation/md.py:288-413).  One conformation per workgroup (four waves up to 64 atoms, sixteen above; the
 * force / Langevin kernels one / sixteen), coordinates and the pair-exclusion bit matrix in LDS: molecules up to ~1000 atoms (TW_ERR_INVALID beyond what 160 KiB hold; the force / Langevin entry points
 * below ~850 / ~800). */
int tw_amber_energy(const tw_forcefield* ff, const float* coords, double* out_energy,
                    double* out_terms, int64_t n_rows, void* stream);

/* Forces of the same potential (kJ/mol/nm, fp64): out_forces [n_rows,n_atoms,3] = -dE/dx, analytic for all five terms
 * (the Born-radius chain rule of GBSA-OBC included); out_energy [n_rows] may be NULL.  What OpenMM's
 * Context.getState(getForces=True) returns for the Systems of simulation/md.py:128-187; pinned against the forces of the
 * reference's own known-answer file (simulation/testdata/implicit-2olx-traj-cpu-arrays.npz, simulation/tests/test_md.py:35-47). */
int tw_amber_energy_forces(const tw_forcefield* ff, const float* coords, double* out_energy, double* out_forces,
                           int64_t n_rows, void* stream);

/* `n_steps` integration steps of every conformation on the device - what `openmm_step` (utils/evaluation_utils.py:439-466)
 * does through the caller's openmm.app.Simulation, for the two integrators simulation/md.py:213-231 builds:
 *   scheme 0  LangevinMiddleIntegrator (v += dt F/m; x += dt/2 v; v <- a v + sqrt(1 - a^2) sqrt(kT/m) N(0,1); x += dt/2 v)
 *   scheme 1  LangevinIntegrator       (v <- a v + (1 - a)/friction F/m + sqrt(kT (1 - a^2)/m) N(0,1); x += dt v),  a = exp(-friction dt)
 * coords (nm) / velocs (nm/ps) [n_rows,n_atoms,3] are updated in place; masses [n_atoms] in dalton; kbT in kJ/mol;
 * friction 0 = plain leapfrog.  The Gaussian noise is counter-based on (seed, conformation, first_step + step, atom): a
 * trajectory is reproducible for a seed but NOT the trajectory OpenMM would produce (its generator is its own).
 * out_energy [n_rows] (may be NULL): potential energy at the positions of the last force evaluation. */
int tw_langevin_steps(const tw_forcefield* ff, const float* masses, float* coords, float* velocs, int32_t n_steps,
                      double timestep_ps, double friction_per_ps, double kbT, int32_t scheme, uint64_t seed, int64_t first_step,
                      double* out_energy, int64_t n_rows, void* stream);

/* The accept step of sample_with_model (utils/evaluation_utils.py:659-713) for one chain:
 *   exp_ = e_pot_y/kbT(scaled by caller) ...: exponent[s] = energy[s] + p_xy[s] - p_yx[s];
 *   p_acc = min(1, e^-exponent); accepted[s] = u[s] < p_acc; k = first accepted index (or S-1);
 *   if any accepted: x <- y[k].  result (device, int32[4]) = {k_unclipped, any_accepted, 0, 0}.
 * energy = (e_pot_y - e_pot_x) + (e_kin_y - e_kin_x) is formed by the caller.
 * x_coords/x_velocs [n_atoms,3] are the chain state, updated in place; pass both NULL to only
 * score the proposals (the host then applies the reference's `k = min(k, N - i)` clipping,
 * evaluation_utils.py:680, before it moves the chain). */
int tw_mh_accept(const float* energy, const float* p_xy, const float* p_yx, const float* u,
                 const float* y_coords, const float* y_velocs, float* x_coords, float* x_velocs,
                 float* out_exponent, float* out_p_acc, uint8_t* out_accepted, int32_t* result,
                 int64_t n_proposals, int32_t n_atoms, void* stream);

/* Extension (no reference counterpart): n_chains independent chains evaluated in one flow
 * call of n_chains conditioning states x n_proposals samples.  Every per-proposal array is in the row order of that
 * call, index = proposal * n_chains + chain; x_coords / x_velocs [n_chains, n_atoms, 3]; result int32 [n_chains, 4].
 * Per chain exactly the semantics of tw_mh_accept. */
int tw_mh_accept_chains(const float* energy, const float* p_xy, const float* p_yx, const float* u,
                        const float* y_coords, const float* y_velocs, float* x_coords, float* x_velocs,
                        float* out_exponent, float* out_p_acc, uint8_t* out_accepted, int32_t* result,
                        int64_t n_proposals, int64_t n_chains, int32_t n_atoms, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One whole iteration of the loop body of sample_with_model (utils/evaluation_utils.py:609-713) in one call, for the
 * case the reference's scripts run: B = 1 conditioning state, S parallel proposals, accept = True.  Equivalent to
 *   tw_flow_sample_with_logp (proposals y, log p(y|x))  ->  tw_amber_energy (y and x)  ->  tw_kinetic_energy (y and x)
 *   ->  tw_chirality_changed  ->  tw_flow_log_likelihood (reverse move, velocities negated unless random_velocs)
 *   ->  tw_mh_accept
 * with the elementwise steps in between done on the device; the glue is four small launches instead of ~70.
 *   zy_coords / zy_velocs [n_proposals + 1, n_atoms, 3]: IN the latent noise of the S proposals, already multiplied by
 *     exp(prior log-scale) (rows 0..S-1; flow.py:274-275, coords drawn first); OUT the proposals y_coords / y_velocs in
 *     those rows.  Row S of zy_coords is scratch (the current state rides along in the energy launch).
 *   x_coords / x_velocs [n_atoms, 3]: the current state (x_velocs already resampled if the caller resamples);
 *   new_coords / new_velocs [n_atoms, 3]: y[k] if a proposal was accepted, else the current state.
 *   u [S]: the uniform draws of the accept test.
 *   out_stats [8, S]: p_acc, log p(y|x), log p(x~|y~), exponent, E_pot(y)/kT (+2000 on a chirality change), E_kin(y),
 *     E_pot(y)/kT - E_pot(x)/kT, E_kin(y) - E_kin(x)  (the ChainStats columns, evaluation_utils.py:721-728);
 *   out_accepted [S]; result int32[4] = {first accepted index or S-1, any accepted, 0, 0}  (as tw_mh_accept).
 * The caller applies the reference's clip k = min(k, N - i) (:680) when it emits the chain states. */
typedef struct {
  int32_t random_velocs;       /* compute_kinetic_energy form and the sign of the reverse move's velocities */
  int32_t n_centres;           /* chirality guard (utils/chirality.py): 0 = off */
  const int32_t* centres;      /* [n_centres, 4] */
  const float* reference_signs;/* [n_centres] */
  const float* masses;         /* [n_atoms]; may be NULL when random_velocs */
  float kbT;                   /* kJ/mol */
} tw_mh_options;

int64_t tw_mh_iteration_workspace_bytes(const tw_flow_desc* desc, int64_t n_proposals, int32_t n_atoms);
int tw_mh_iteration(const tw_flow_desc* desc, const float* raw, const void* packed, int32_t path,
                    const tw_forcefield* ff, const tw_mh_options* opt, const int32_t* atom_types,
                    const uint8_t* masked, int32_t n_atoms, const float* x_coords, const float* x_velocs,
                    float* zy_coords, float* zy_velocs, const float* u, float* new_coords, float* new_velocs,
                    float* out_stats, uint8_t* out_accepted, int32_t* result, int64_t n_proposals,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ABI 8 - extension (the reference runs ONE chain per process, utils/evaluation_utils.py:517): the same
 * iteration for n_chains independent chains of one molecule in lock-step - ONE flow reverse pass over n_proposals x n_chains
 * rows, ONE energy launch (proposals + the n_chains current states), ONE forward pass, one accept workgroup per chain.  Row
 * order of every per-proposal array: index = proposal * n_chains + chain (the reference's [S, B] reshape, flow.py:284-296).
 * Per chain the arithmetic is that of tw_mh_iteration on that chain alone.
 *   atom_types / masked [n_chains, n_atoms]; x_coords / x_velocs [n_chains, n_atoms, 3]: the current states;
 *   cur_velocs [n_chains, n_atoms, 3] OUT: the velocities the iteration ran with (x_velocs, or the resampled ones);
 *   zy_coords [(n_proposals + 1) * n_chains, n_atoms, 3], zy_velocs [n_proposals * n_chains, n_atoms, 3]: latents IN (scaled,
 *     as for tw_mh_iteration) / proposals OUT; the last n_chains rows of zy_coords are scratch;
 *   u [n_proposals, n_chains]: accept uniforms; new_coords / new_velocs [n_chains, n_atoms, 3];
 *   out_stats [8, n_proposals, n_chains], out_accepted [n_proposals, n_chains], result int32 [n_chains, 4].
 * draws == NULL: the caller drew the latents, the (resampled) x_velocs and u - the route the oracle / trace tests take.
 * draws != NULL: they are drawn inside the first glue kernel by a counter-based generator - Philox4x32-10 with
 *   key = (seed & 0xffffffff, seed >> 32), counter = (element >> 2, kind | (iteration >> 32) << 4, iteration & 0xffffffff,
 *   first_chain + chain); kind 0 coordinate latents, 1 velocity latents (element = proposal * 3 n_atoms + component), 2
 *   resampled current velocities (element = component), 3 uniforms (element = proposal); word element & 3 of the block; normals
 *   by Box-Muller on the word pairs (0,1), (2,3): u1 = w * 2^-32 + 2^-33, u2 = (w >> 8) * 2^-24, cos for even elements, sin for
 *   odd; latents multiplied by exp(prior log-scale); uniforms (w >> 8) * 2^-24.  zy_* and u are then pure outputs, and a
 *   chain's draws depend on (seed, first_chain + chain, iteration) only - not on n_chains, not on what ran before.
 *   tw_mh_draw_chains writes exactly those draws (any of the four output pointers may be NULL) for tests and replays. */
typedef struct {
  uint64_t seed;
  int64_t iteration;        /* >= 0; the caller counts iterations */
  int32_t first_chain;      /* global id of chain 0 of this call (rank * chains_per_rank) */
  int32_t resample_velocs;  /* draw fresh N(0,1) current velocities (the reference's --resample-velocities; needs random_velocs) */
} tw_mh_draws;

int64_t tw_mh_iteration_chains_workspace_bytes(const tw_flow_desc* desc, int64_t n_proposals, int64_t n_chains, int32_t n_atoms);
int tw_mh_iteration_chains(const tw_flow_desc* desc, const float* raw, const void* packed, int32_t path,
                           const tw_forcefield* ff, const tw_mh_options* opt, const tw_mh_draws* draws,
                           const int32_t* atom_types, const uint8_t* masked, int32_t n_atoms, const float* x_coords,
                           const float* x_velocs, float* cur_velocs, float* zy_coords, float* zy_velocs, float* u,
                           float* new_coords, float* new_velocs, float* out_stats, uint8_t* out_accepted, int32_t* result,
                           int64_t n_proposals, int64_t n_chains, void* workspace, int64_t workspace_bytes, void* stream);
int tw_mh_draw_chains(const tw_flow_desc* desc, const float* raw, const tw_mh_draws* draws, float* z_coords, float* z_velocs,
                      float* u, float* velocs, int64_t n_proposals, int64_t n_chains, int32_t n_atoms, void* stream);

/* check_symmetry_change (utils/chirality.py:40-80): sign of the triple product at each
 * chirality centre vs reference_signs; out_changed [n_rows] uint8.  centres [n_centres,4] int32. */
int tw_chirality_changed(const float* coords, const int32_t* centres, const float* reference_signs,
                         int32_t n_centres, uint8_t* out_changed, int64_t n_rows, int32_t n_atoms,
                         void* stream);

/* Measurement hooks (bench.py roofline leg): between tw_profile_begin() and tw_profile_end() every
 * launch of the dominant kernel (the fused net-block kernel) is bracketed by hipEvents recorded on
 * the launch stream.  tw_profile_end synchronises those events and returns (host pointers) the
 * summed kernel time in milliseconds and the number of launches (of the bracketed ones: with TW_PROFILE_STRIDE=n in the
 * environment at tw_profile_begin only every n-th launch is bracketed - two event records cost a few microseconds of
 * stream time per launch).  Not thread-safe. */
int tw_profile_begin(void);
int tw_profile_end(double* total_ms, int64_t* launches);

/* Measurement hook (bench.py roofline.power_bound): a bare stream of v_mfma_f32_16x16x32_f16 - 36 * iters instructions per
 * wave, one wave per SIMD, `workgroups` workgroups of four waves (256 fills the chip), random fp16 operands of magnitude ~1 -
 * i.e. what the matrix pipes of THIS device sustain when nothing else happens (tools/probe/mfma_stream_probe.hip as an entry
 * point).  Returns (host pointers) the shader clocks one wave spent in the stream and the launch's duration from HIP events:
 * clock = cycles / ms, sustained rate = workgroups * 4 * 36 * iters * 16384 FLOP / ms.  Synchronous; launches on `stream`. */
int tw_probe_mfma_clock(int32_t workgroups, int32_t iters, int64_t* cycles, double* ms, void* stream);

/* Measurement hook (bench.py roofline.kernel, ABI 7): the template instantiation of the fused net-block kernel the calling
 * thread launched last, spelled as rocprofv3 prints it ("tw::netblock_h3_kernel<3, true, false, false, false, true, false,
 * false>": NT, ASM, DENSE, WIDE, RFF, ENC, H1, NG6) - which layout and build a flow call took is decided per launch (molecule
 * size, row count, tw_debug_set_flags).  A static string ("" before the first launch); never NULL. */
const char* tw_last_netblock_kernel(void);

/* ABI 8: the instantiation a flow pass over n_rows x n_atoms on `path` (TW_PATH_FUSED_H3 / TW_PATH_FUSED_H1) WOULD launch under
 * the debug flags in force - the launch code's own branch run dry, nothing is launched and no GPU is needed.  "" when the path
 * does not serve the shape.  tests/test_host_logic.py enumerates 1 .. 192 atoms through it and holds every kernel it names to
 * ScratchSize 0 (one stated exception). */
const char* tw_flow_selected_kernel(const tw_flow_desc* desc, int32_t n_atoms, int64_t n_rows, int32_t path);

/* Safety net for TW_PATH_FUSED_H3 (no counterpart in the reference): *out_flag = 1 if, since the last reset, any
 * coupling net on the current device returned a non-finite scale or shift.  The split-fp16 kernel holds its operands in
 * fp16 (|value| < 65504); a checkpoint whose activations leave that range produces inf/NaN there, which the exact-f32
 * paths would not.  Synchronous (a device-to-host copy of one int): call it where the caller synchronises anyway.
 * `reset` != 0 clears the flag. */
int tw_flow_nonfinite(int32_t reset, int32_t* out_flag);

/* Debug / measurement switches of the split-fp16 kernel.  0 restores normal operation.  Process-wide (see the
 * thread-safety note at the top).  Bits 0, 1, 6, 7, 11 are timing experiments that make results WRONG: the product
 * library refuses them (TW_ERR_INVALID) and compiles their branches out; they exist in a -DTW_EXPERIMENTS build only
 * (TW_EXPERIMENTS=1 python -m timewarp_amd.build).
 *   bit 0 (1)  no weight LDS-DMA after the prologue   } timing experiments on the compiled-C++ sections only:
 *   bit 1 (2)  no workgroup barriers                  } results become WRONG
 *   bit 2 (4)  tw_debug_netblock dumps the attention output (before the first LayerNorm) instead of the layer output
 *   bit 3 (8)  run the compiled-C++ variant of the kernel (attention / FFN / in / out sections as C++ instead of the
 *              generated asm blocks); same results, slower - the A/B reference for the asm
 *   bit 4 (16) tw_debug_netblock: wave 0 of workgroup 0 writes s_memtime stamps of the section boundaries into the
 *              dump buffer instead of activations (tools/profile_h3_sections.py)
 *   bit 5 (32) split-fp16 flow pass: every affine coupling update as its own launch instead of in the next net-block
 *              launch's prologue (A/B switch; same results up to the summation order of the log-determinant)
 *   bit 6 (64), bit 7 (128) fused dense kernel, timing experiments (results WRONG): no softmax section / also no LDS round
 *              trip of q, k, v - what the attention block costs beyond its MFMAs (0.8 of 11.4 ms per 1000-proposal pass)
 *   bit 10 (1024) split-fp16 kernel: do not zero the padding tokens of a wave between sections (A/B switch; results equal)
 *   bit 11 (2048) split-fp16 dense kernel, compiled-C++ attention: no scores / softmax / P.V (timing experiment, results WRONG)
 *   bit 12 (4096) split-fp16 kernel-attention kernel, <= 48 atoms: run the per-section build (attention / FFN asm blocks with
 *              compiled glue between them) instead of the encoder-stack statement (tools/gen_h3_enc_asm.py); same results
 *              up to the last bits.  Activation dumps (tw_debug_netblock) and bits 2 / 4 take that build anyway.
 *   bit 13 (8192) ... the encoder-stack build even with a dump buffer (profiling: only stamps outside the stack)
 *   bit 14 (16384) molecules of 25 .. 48 atoms: never the wide layout; bit 15 (32768): the wide layout wherever it exists
 *              (the launch code otherwise picks the layout that needs fewer rounds of the chip; same results up to the last
 *              bits; A/B switch and tests)
 *   bit 16 (65536) molecules of 49 .. 64 atoms: always 64-token waves (one molecule per wave); bit 17 (131072): never -
 *              the wide layout instead (the launch code otherwise picks by rounds of the chip x cost per workgroup)
 *   bit 18 (262144) wide layout, 65 .. 96 atoms: five-group key windows (molecules back to back where that fits) instead of
 *              the 96-slot stride with three-group windows; same results up to the last bits (A/B switch and tests)
 *   bit 19 (524288) wide layout: the transposed tile written with two-byte stores (r03's form) instead of through the matrix
 *              pipe; bit-identical results (A/B switch and tests)
 *   bit 20 (1048576) molecules of 97 .. 128 atoms: never the paired layout (one molecule per pair of 64-token waves, two per
 *              workgroup; r05) - the wide layout's 48-token waves instead, one molecule per workgroup; same results up to the
 *              last bits (A/B switch and tests).  The paired layout exists as the encoder-stack build only: bits 2 / 4 / 12
 *              (without 13) and tw_debug_netblock take the 48-token wide layout as well
 *   bit 21 (2097152) per-op path: the row-wise scores kernel and the tiled MFMA mixing kernel (what molecules above 160 / 64
 *              atoms take) at every size; the same scores up to the order of a row sum's double additions, the mixing in another
 *              summation order (A/B switch and tests)
 *   bit 22 (4194304) / bit 23 (8388608)  tw_mh_iteration: the energy kernel on the caller's stream / on the side stream,
 *              whatever the launch size (default: side stream only while the flow's launches leave compute units idle)
 *   bit 24 (16777216) TW_PATH_SIMPLE_H3: the FFN as two linear launches + add_ln, the attention unfolded, even when the path's pack
 *              is at hand (default then: one launch of the fused kernels' chunk loop on the flat token list; folded attention);
 *              A/B switch and tests
 *   bit 25 (33554432) TW_PATH_SIMPLE_H3, folded attention: the 768 -> 128 GEMM as its own launch behind the mixing kernel instead
 *              of inside it (attend_fold_h3_kernel); bit 26 (67108864): inside it with one workgroup per query tile whatever the
 *              launch size (default below 400 workgroups: the heads over 6 / 2 workgroups per tile + a finishing launch); bit 27 (134217728): residual + LayerNorm 1 as the add_ln launch behind that kernel instead of
 *              in its epilogue; A/B switches and tests
 *   bit 28 (268435456) TW_PATH_SIMPLE_H3: the in-MLP and the out-MLP as two linear launches each instead of one launch of the fused
 *              kernels' statements on the flat token list (h3_io_tokens_kernel)
 *   bit 29 (536870912) / bit 30 (1073741824)  TW_PATH_SIMPLE_H3: the FFN / MLP token launches on 48-token / on 64-token waves whatever
 *              the launch size (default: whichever needs fewer rounds' worth of the chip); same arithmetic per token.  Bit 29 also
 *              turns off what the per-op paths do for launches that do not fill the chip: the FFN's hidden layer over four workgroups
 *              per token tile, and the second net of a coupling layer on a side stream (both nets then run on the caller's stream)
 *   bit 31 (pass INT_MIN) per-op paths, dense softmax variant: the scalar attention kernels above 64 atoms instead of
 *              sdpa_mfma_kernel (fp32 matrix pipe); A/B switches and tests */
int tw_debug_set_flags(int flags);

/* Debug/inspection: run ONE net-block of the fused path and dump the activation after every
 * stage (in_mlp, each encoder layer, out_mlp) as [n_rows,n_atoms,d] row-major floats.
 * dump must hold (n_layers+1)*n_rows*n_atoms*d_model + n_rows*n_atoms*3 floats.  Tests only. */
int tw_debug_netblock(const tw_flow_desc* desc, const float* raw, const float* packed,
                      int32_t coupling, int32_t net, const int32_t* atom_types,
                      const float* x_coords, const float* x_velocs, const uint8_t* masked,
                      int64_t n_cond, const float* z_other, int64_t n_rows, int32_t n_atoms,
                      int32_t path, float* dump, void* workspace, int64_t workspace_bytes,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TIMEWARP_HIP_H */
