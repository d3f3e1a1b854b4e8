"""Shared test helpers: golden loading and the name-seeded full-size weights."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

from oracle import flow_oracle as fo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FULL_KERNEL_SPEC = fo.FlowSpec(variant="kernel")
FULL_DENSE_SPEC = fo.FlowSpec(variant="dense", n_head=8)
TINY_KERNEL_SPEC = fo.FlowSpec(variant="kernel", num_coupling_layers=2, num_transformer_layers=2)
TINY_LEARNABLE_SPEC = fo.FlowSpec(variant="kernel", num_coupling_layers=2, num_transformer_layers=2,
                                  attention_type="learnable_kernel")
TINY_CHEB_SPEC = fo.FlowSpec(variant="kernel", num_coupling_layers=2, num_transformer_layers=2,
                             attention_type="chebyshev_kernel", force_asymptotic_zero=False)
TINY_CHEB_ZERO_SPEC = fo.FlowSpec(variant="kernel", num_coupling_layers=2, num_transformer_layers=2,
                                  attention_type="chebyshev_kernel", force_asymptotic_zero=True)
TINY_DENSE_SPEC = fo.FlowSpec(variant="dense", num_coupling_layers=2, num_transformer_layers=2, n_head=2)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    data = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("sd::") and z[k].dtype.kind != "U"}
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return data, sd


_cache = {}


def full_kernel_sd(calibrated=False):
    key = ("k", calibrated)
    if key not in _cache:
        _cache[key] = fo.synth_state_dict(fo.make_template(FULL_KERNEL_SPEC), 0, calibrated)
    return _cache[key]


FULL_CHEB_SPEC = fo.FlowSpec(variant="kernel", attention_type="chebyshev_kernel", force_asymptotic_zero=True)


def full_cheb_sd():
    """Full-size chebyshev_kernel model, order 6, name-seeded (cheb_coeffs too: different in every attention layer)."""
    if "c" not in _cache:
        _cache["c"] = fo.synth_state_dict(fo.make_template(FULL_CHEB_SPEC, cheb_order=6), 0)
    return _cache["c"]


def full_dense_sd():
    if "d" not in _cache:
        _cache["d"] = fo.synth_state_dict(fo.make_template(FULL_DENSE_SPEC), 0)
    return _cache["d"]


def full_dense_posenc_sd():
    """transformer_nvp_posenc.yaml at full size: name-seeded weights, the position encoders' Gaussian vectors (buffers
    the reference draws at construction) from the golden file."""
    if "dp" not in _cache:
        _, gv = load("dense_posenc_full_ad")
        t = fo.make_template(FULL_DENSE_SPEC, rff_dim=128)
        for k, v in gv.items():
            assert k in t and t[k].shape == v.shape, k
            t[k] = v
        _cache["dp"] = fo.synth_state_dict(t, 0)
    return _cache["dp"]


def rel_err(a, b):
    """max |a - b| / max |b|: error relative to the scale of the tensor (coordinates, whose elements pass through 0)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_rel_err(a, b):
    """max over elements of |a - b| / |b|: the element-wise relative error, for quantities bounded away from 0 such as
    log-densities (log p ~ -184 for alanine dipeptide); the north-star bar read literally."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float(((a - b).abs() / b.abs().clamp_min(1e-30)).max())


# ---------------------------------------------------------------------------------------------
# product-side helpers (GPU tests)
# ---------------------------------------------------------------------------------------------
def tw_kernel_model(sd, emb=32, d_model=128, ff=2048, hidden=256, n_coupling=8, n_layers=3,
                    lengthscales=(0.1, 0.2, 0.5, 0.7, 1.0, 1.2), path=0, device="cuda", attention_type="kernel",
                    cheb_order=None, force_asymptotic_zero=None, normalise=True, pos_mod2=0):
    import timewarp_amd as tw

    enc = tw.CustomAttentionEncoderLayerConfig(d_model=d_model, dim_feedforward=ff, dropout=0.0,
                                               num_heads=len(lengthscales), attention_type=attention_type,
                                               lengthscales=list(lengthscales), normalise_kernel_values=normalise,
                                               cheb_order=cheb_order, force_asymptotic_zero=force_asymptotic_zero)
    cfg = tw.ModelConfig("custom_attention_transformer_nvp",
                         custom_transformer_nvp_config=tw.CustomAttentionTransformerNVPConfig(
                             emb, [hidden], n_coupling, n_layers, enc, position_layer_index_mod_2=pos_mod2))
    m = tw.model_constructor(cfg)
    m.load_state_dict(sd)
    if path is not None:  # None: keep the constructor's default (TW_EXECUTION_PATH)
        m.execution_path = path
    return m.to(device).eval()


def tw_dense_model(sd, emb=32, d_model=128, ff=2048, hidden=256, n_coupling=8, n_layers=3, n_head=8, rff_dim=0,
                   path=0, device="cuda"):
    import timewarp_amd as tw

    rff = tw.RFFPositionEncoderConfig(rff_dim, 1.0, 1.0) if rff_dim else None
    cfg = tw.ModelConfig("transformer_nvp", transformer_nvp_config=tw.TransformerNVPConfig(
        emb, d_model, [hidden], n_coupling, n_layers, tw.TransformerConfig(n_head, ff, 0.0), rff))
    m = tw.model_constructor(cfg)
    m.load_state_dict(sd)
    if path is not None:  # None: keep the constructor's default (TW_EXECUTION_PATH)
        m.execution_path = path
    return m.to(device).eval()


def run_model_case(model, d, prefix="", device="cuda"):
    """Run the three model calls of an MH iteration through the product API; returns a dict of
    CPU tensors named like the golden keys."""
    g = lambda k: d[prefix + k].to(device)
    out = {}
    out["loglik"] = model.log_likelihood(
        atom_types=g("atom_types"), x_coords=g("x_coords"), x_velocs=g("x_velocs"), y_coords=g("y_coords"),
        y_velocs=g("y_velocs"), adj_list=None, edge_batch_idx=None, masked_elements=g("masked")).cpu()
    if prefix + "z_coords" in d:
        S = d[prefix + "z_coords"].shape[0]
        yc, yv, lp = model.conditional_sample_with_logp(
            atom_types=g("atom_types"), x_coords=g("x_coords"), x_velocs=g("x_velocs"), adj_list=None,
            edge_batch_idx=None, masked_elements=g("masked"), num_samples=S, z_coords=g("z_coords"),
            z_velocs=g("z_velocs"))
        out.update(s_y_coords=yc.cpu(), s_y_velocs=yv.cpu(), s_logp=lp.cpu())
        # reverse-move density on the GOLDEN proposals (isolates this call from the sampling error)
        gy, gv = g("s_y_coords").squeeze(1), g("s_y_velocs").squeeze(1)
        out["logp_yx"] = model.log_likelihood(
            atom_types=g("atom_types").repeat(S, 1), x_coords=gy, x_velocs=-gv,
            y_coords=g("x_coords").repeat(S, 1, 1), y_velocs=-g("x_velocs").repeat(S, 1, 1), adj_list=None,
            edge_batch_idx=None, masked_elements=g("masked").repeat(S, 1)).cpu()
    assert_not_demoted(model)
    return out


def assert_not_demoted(model):
    """A parity test of the split-fp16 kernels must not pass because the range guard moved the model to the f32 kernels."""
    assert not getattr(model, "demoted", False), "the split-fp16 range guard fired: this test ran on the f32 kernels"


def assert_case_close(out, d, prefix="", tol=1e-5):
    keep = ~d[prefix + "masked"][0]
    assert rel_err(out["loglik"], d[prefix + "loglik"]) < tol, ("loglik", rel_err(out["loglik"], d[prefix + "loglik"]))
    if "s_logp" in out:
        for k in ("s_y_coords", "s_y_velocs"):
            e = rel_err(out[k][:, :, keep], d[prefix + k][:, :, keep])
            assert e < tol, (k, e)
        for k in ("s_logp", "logp_yx"):
            e = rel_err(out[k], d[prefix + k])
            assert e < tol, (k, e)
            if float(d[prefix + k].abs().min()) > 1.0:  # element-wise too where no log-density is near 0
                e = elem_rel_err(out[k], d[prefix + k])
                assert e < tol, (k + " element-wise", e)


# ---------------------------------------------------------------------------------------------
# C energy oracle (oracle/energy_oracle.c)
# ---------------------------------------------------------------------------------------------
class _OracleFF(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_atoms", "n_bonds", "n_angles", "n_torsions", "n_exceptions", "has_gbsa")] + \
               [(n, C.c_double) for n in ("cutoff", "rf_dielectric", "solute_dielectric", "solvent_dielectric", "surface_area_energy")] + \
               [(n, C.c_void_p) for n in ("bond_idx", "bond_par", "angle_idx", "angle_par", "torsion_idx", "torsion_par", "exc_idx", "exc_par", "atom_par")]


def oracle_energy(tables, coords, dtype=np.float32):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_build", "libenergy_oracle.so")
    if not os.path.exists(so):  # test infrastructure: gcc only
        subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(so)
    arrs = [np.ascontiguousarray(a) for a in (
        tables.bond_idx.astype(np.int32), tables.bond_par, tables.angle_idx.astype(np.int32), tables.angle_par,
        tables.torsion_idx.astype(np.int32), tables.torsion_par, tables.exc_idx.astype(np.int32), tables.exc_par, tables.atom_par)]
    ff = _OracleFF(tables.n_atoms, len(arrs[0]), len(arrs[2]), len(arrs[4]), len(arrs[6]), int(tables.has_gbsa),
                   tables.cutoff, tables.rf_dielectric, tables.solute_dielectric, tables.solvent_dielectric,
                   tables.surface_area_energy, *[a.ctypes.data for a in arrs])
    x = np.ascontiguousarray(coords, dtype=dtype)
    n = x.shape[0]
    out, terms = np.zeros(n), np.zeros((n, 5))
    fn = lib.oracle_amber_energy if dtype == np.float32 else lib.oracle_amber_energy_f64
    fn(C.byref(ff), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                            terms.ctypes.data_as(C.c_void_p), C.c_int64(n))
    return out, terms


# ---------------------------------------------------------------------------------------------
# MH loop at the bench configuration: shared host noise, C energy oracle, weight variants
# ---------------------------------------------------------------------------------------------
class HostNoise:
    """The loop's random draws from a seeded CPU generator, in the reference's order (the DeviceNoise protocol of
    timewarp_amd/utils/evaluation_utils.py; also what oracle/mh_oracle.py takes).  Two instances with one seed feed
    the product (device="cuda") and the oracle (device="cpu") the same numbers."""

    def __init__(self, seed, device="cpu"):
        self.g = torch.Generator().manual_seed(seed)
        self.device = device

    def randn_like(self, t):
        return torch.randn(t.shape, generator=self.g).to(self.device)

    def latents(self, S, B, V, scale_c, scale_v):
        zc = torch.randn((S, B, V, 3), generator=self.g) * scale_c.cpu()
        zv = torch.randn((S, B, V, 3), generator=self.g) * scale_v.cpu()
        return zc.to(self.device), zv.to(self.device)

    def uniform(self, S):
        return torch.rand(S, generator=self.g).to(self.device)


class OracleAmberEnergy:
    """oracle/energy_oracle.c behind the energy-callable contract ([n,1] float32 kJ/mol, .kbT)."""

    def __init__(self, tables, temperature=310.0):
        self.tables, self.kbT = tables, 8.314462618e-3 * temperature

    def __call__(self, coords):
        out, _ = oracle_energy(self.tables, coords.reshape(-1, self.tables.n_atoms, 3).numpy())
        return torch.from_numpy(out).to(torch.float32)[:, None]


def mh_state_dict(kind, random_velocs, out_scale=1e-4, coords_log_scale=-7.0):
    """Full-size kernel_transformer_nvp weights for the MH-iteration parity tests.
    "bench": exactly bench.py's calibration (identity flow: last out_mlp layer zeroed; coordinate prior e^-7).
    "scaled": the last out_mlp layers scaled by 1e-4 instead of zeroed - every coupling net's output now moves the
    proposals, the log-determinants and both log-densities (shifts ~1e-4 nm keep the stiff bonded terms acceptable)."""
    if kind == "bench":
        return dict(fo.synth_state_dict(fo.make_template(FULL_KERNEL_SPEC), 0, calibrated=True, coords_log_scale=-7.0,
                                        velocs_log_scale=0.0))
    sd = dict(fo.synth_state_dict(fo.make_template(FULL_KERNEL_SPEC), 0))
    for k in sd:
        if ".out_mlp._layers.2." in k:
            sd[k] = sd[k] * out_scale
    sd["coords_prior_log_scale"] = torch.tensor(float(coords_log_scale))
    sd["velocs_prior_log_scale"] = torch.tensor(0.0 if random_velocs else -3.0)
    return sd


# ---- the counter-based generator of tw_mh_iteration_chains (include/timewarp_hip.h, tw_mh_draws), restated in numpy -----------

def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on arrays of
    counters [..., 4] and keys [..., 2] (uint32); returns [..., 4] uint32.  Checked against the paper's known-answer vectors
    in tests/test_host_logic.py."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & m32, p1 >> np.uint64(32), p1 & m32
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, axis=-1).astype(np.uint32)


def chain_draw_words(seed, iteration, chain, kind, n_elements):
    """The 32-bit words behind elements 0 .. n_elements - 1 of one (chain, iteration, kind) stream: [n_elements, 4] (the whole
    block of each element) - counter = (element >> 2, kind | (iteration >> 32) << 4, iteration & 0xffffffff, chain)."""
    e = np.arange(n_elements, dtype=np.uint64)
    ctr = np.zeros((n_elements, 4), dtype=np.uint64)
    ctr[:, 0] = e >> np.uint64(2)
    ctr[:, 1] = np.uint64(kind) | (np.uint64((iteration >> 32) & 0x0FFFFFFF) << np.uint64(4))
    ctr[:, 2] = np.uint64(iteration & 0xFFFFFFFF)
    ctr[:, 3] = np.uint64(chain)
    key = np.zeros((n_elements, 2), dtype=np.uint64)
    key[:, 0], key[:, 1] = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    return philox4x32_10(ctr, key)


def chain_draw_uniforms(seed, iteration, chain, n):
    w = chain_draw_words(seed, iteration, chain, 3, n)
    return ((w[np.arange(n), np.arange(n) & 3] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24))


def chain_draw_normals(seed, iteration, chain, kind, n):
    """Standard normals of a stream in float64 (the kernel's are float32 Box-Muller: compare to ~1e-6)."""
    w = chain_draw_words(seed, iteration, chain, kind, n).astype(np.float64)
    e = np.arange(n)
    pair = (e >> 1) & 1
    w1 = w[e, 2 * pair]
    w2 = np.floor(w[e, 2 * pair + 1] / 256.0)
    u1 = (np.float32(w1) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float64)   # the kernel's fmaf, float32 operands
    u2 = w2 * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u1))
    return np.where(e & 1, r * np.sin(2 * np.pi * u2), r * np.cos(2 * np.pi * u2))
