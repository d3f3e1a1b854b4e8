"""Shared test helpers: golden loading and the name-seeded full-size weights."""
import os

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


def full_dense_sd():
    if "d" not in _cache:
        _cache["d"] = fo.synth_state_dict(fo.make_template(FULL_DENSE_SPEC), 0)
    return _cache["d"]


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ---------------------------------------------------------------------------------------------
# product-side helpers (GPU tests)
# ---------------------------------------------------------------------------------------------
def tw_kernel_model(sd, emb=32, d_model=128, ff=2048, hidden=256, n_coupling=8, n_layers=3,
                    lengthscales=(0.1, 0.2, 0.5, 0.7, 1.0, 1.2), path=0, device="cuda", attention_type="kernel",
                    cheb_order=None, force_asymptotic_zero=None):
    import timewarp_amd as tw

    enc = tw.CustomAttentionEncoderLayerConfig(d_model=d_model, dim_feedforward=ff, dropout=0.0,
                                               num_heads=len(lengthscales), attention_type=attention_type,
                                               lengthscales=list(lengthscales), normalise_kernel_values=True,
                                               cheb_order=cheb_order, force_asymptotic_zero=force_asymptotic_zero)
    cfg = tw.ModelConfig("custom_attention_transformer_nvp",
                         custom_transformer_nvp_config=tw.CustomAttentionTransformerNVPConfig(
                             emb, [hidden], n_coupling, n_layers, enc))
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
    return out


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
