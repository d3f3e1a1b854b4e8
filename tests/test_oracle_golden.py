"""Pin the oracle (oracle/flow_oracle.py) against vectors produced by the real reference
(oracle/gen_golden.py).  CPU only.  Tolerance: 1e-5 relative is the north-star bar; the oracle
is the same fp32 op chain as the reference so it lands at ~1e-6."""
import torch
import pytest

from oracle import flow_oracle as fo
from tests import helpers as H

TOL = 2e-6


def _check_case(d, sd, spec, prefix=""):
    g = lambda k: d[prefix + k]
    ll = fo.log_likelihood(sd, spec, g("atom_types"), g("x_coords"), g("x_velocs"), g("y_coords"),
                           g("y_velocs"), g("masked"))
    assert H.rel_err(ll, g("loglik")) < TOL
    if prefix + "z_coords" in d:
        yc, yv, lp = fo.conditional_sample_with_logp(
            sd, spec, g("atom_types"), g("x_coords"), g("x_velocs"), g("masked"), g("z_coords"), g("z_velocs"))
        keep = ~g("masked")[0]
        assert H.rel_err(yc[:, :, keep], g("s_y_coords")[:, :, keep]) < TOL
        assert H.rel_err(yv[:, :, keep], g("s_y_velocs")[:, :, keep]) < TOL
        assert H.rel_err(lp, g("s_logp")) < TOL
        S = yc.shape[0]
        p_yx = fo.log_likelihood(
            sd, spec, g("atom_types").repeat(S, 1), g("s_y_coords").squeeze(1), -g("s_y_velocs").squeeze(1),
            g("x_coords").repeat(S, 1, 1), -g("x_velocs").repeat(S, 1, 1), g("masked").repeat(S, 1))
        assert H.rel_err(p_yx, g("logp_yx")) < TOL


def test_kernel_tiny_batch_and_sampling():
    d, sd = H.load("kernel_tiny")
    _check_case(d, sd, H.TINY_KERNEL_SPEC)
    _check_case(d, sd, H.TINY_KERNEL_SPEC, "b1_")


def test_kernel_normalise_flag_is_ignored():
    """A model configured with normalise_kernel_values=False: the reference still L1-normalises its scores
    (kernel_attention.py:197-206 never forwards the flag), and so does the oracle whatever its spec says."""
    import dataclasses

    d, sd = H.load("kernel_nonorm_tiny")
    for flag in (False, True):
        spec = dataclasses.replace(H.TINY_KERNEL_SPEC, normalise_kernel_values=flag)
        _check_case(d, sd, spec)
        _check_case(d, sd, spec, "b1_")


def test_kernel_learnable_lengthscales():
    """attention_type "learnable_kernel" with different log_lengthscales per layer: a flow call uses the
    lengthscales of the first attention layer it evaluates (the reference's score cache ignores them)."""
    d, sd = H.load("kernel_learnable_tiny")
    _check_case(d, sd, H.TINY_LEARNABLE_SPEC)
    _check_case(d, sd, H.TINY_LEARNABLE_SPEC, "b1_")
    # and the quirk is real: using chain[0]'s lengthscales for the reverse pass does not reproduce the reference
    wrong = dict(sd)
    k = "flow.chain.{}.scale_transformer.encoder_layers.0.self_attn.attention.log_lengthscales"
    wrong[k.format(1)] = sd[k.format(0)]
    g = lambda n: d["b1_" + n]
    yc, _, _ = fo.conditional_sample_with_logp(wrong, H.TINY_LEARNABLE_SPEC, g("atom_types"), g("x_coords"), g("x_velocs"),
                                               g("masked"), g("z_coords"), g("z_velocs"))
    keep = ~g("masked")[0]
    assert H.rel_err(yc[:, :, keep], g("s_y_coords")[:, :, keep]) > 1e-4


@pytest.mark.parametrize("name,spec", [("kernel_cheb_tiny", H.TINY_CHEB_SPEC), ("kernel_cheb_zero_tiny", H.TINY_CHEB_ZERO_SPEC)])
def test_kernel_chebyshev_attention(name, spec):
    """attention_type "chebyshev_kernel": rational-Chebyshev basis with per-layer coefficients, scores recomputed
    in every attention layer, optional coefficient centring (kernel_attention.py:12-66, 255-339)."""
    d, sd = H.load(name)
    _check_case(d, sd, spec)
    _check_case(d, sd, spec, "b1_")


def test_template_matches_reference_names():
    _, sd = H.load("kernel_tiny")
    t = fo.make_template(H.TINY_KERNEL_SPEC, atom_embedding_dim=4, d_model=8, dim_feedforward=16,
                         mlp_hidden=(8,), lengthscales=(0.1, 0.5, 1.2))
    assert set(t) == set(sd)
    assert all(t[k].shape == sd[k].shape for k in sd)
    _, sdd = H.load("dense_tiny")
    t = fo.make_template(H.TINY_DENSE_SPEC, atom_embedding_dim=4, d_model=8, dim_feedforward=16,
                         mlp_hidden=(8,), rff_dim=4)
    assert set(t) == set(sdd)
    assert all(t[k].shape == sdd[k].shape for k in sdd)


@pytest.mark.parametrize("name,calibrated", [("kernel_full_ad", False), ("kernel_full_ad_calibrated", True)])
def test_kernel_full_ad(name, calibrated):
    d, _ = H.load(name)
    _check_case(d, H.full_kernel_sd(calibrated), H.FULL_KERNEL_SPEC)


def test_kernel_chebyshev_full_ad():
    """Full-size chebyshev_kernel model with different coefficients in every attention layer (name-seeded)."""
    d, _ = H.load("kernel_cheb_full_ad")
    _check_case(d, H.full_cheb_sd(), H.FULL_CHEB_SPEC)


def test_kernel_full_trace_and_scores():
    d, _ = H.load("kernel_full_ad")
    sd, spec = H.full_kernel_sd(), H.FULL_KERNEL_SPEC
    xc = d["x_coords"] - fo.centre_of_mass(d["x_coords"], d["masked"])
    ls = torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2])
    sc = fo.kernel_scores(xc, d["masked"], ls)
    assert H.rel_err(sc, d["scores"]) < TOL
    assert torch.allclose(sc.sum(-1), torch.ones_like(sc.sum(-1)), atol=1e-3)  # tests/test_kernel_attention.py:19-46
    # first net of the reverse pass: chain[7] is a velocity layer -> input uses z_coords
    feats = torch.nn.functional.embedding(d["atom_types"], sd["flow.atom_embedder.weight"])
    S = 2
    u = torch.cat([feats.repeat(S, 1, 1), xc.repeat(S, 1, 1), d["x_velocs"].repeat(S, 1, 1),
                   d["z_coords"][:S, 0]], dim=-1)
    trace = []
    fo.kernel_netblock(sd, "flow.chain.7.scale_transformer", u, sc.repeat(S, 1, 1, 1), spec, trace)
    for name, val in trace:
        assert H.rel_err(val, d["tr_" + name]) < TOL, name


def test_kernel_full_v60_mm_branch():
    d, _ = H.load("kernel_full_v60")
    xc = d["x_coords"] - fo.centre_of_mass(d["x_coords"], d["masked"])
    # V=60 > 25: torch.cdist uses the matmul formulation; the explicit restatement agrees
    assert H.rel_err(fo.cdist_mm(xc, xc), torch.cdist(xc, xc)) < 1e-5
    assert (fo.cdist_direct(xc, xc) - torch.cdist(xc, xc)).abs().max() > 1e-5  # and the direct form does not
    _check_case(d, H.full_kernel_sd(), H.FULL_KERNEL_SPEC)


def test_dense_tiny_and_full():
    d, sd = H.load("dense_tiny")
    _check_case(d, sd, H.TINY_DENSE_SPEC)
    _check_case(d, sd, H.TINY_DENSE_SPEC, "b1_")
    d, _ = H.load("dense_full_ad")
    _check_case(d, H.full_dense_sd(), H.FULL_DENSE_SPEC)


def test_dense_posenc_full():
    """transformer_nvp_posenc.yaml (128 random Fourier position features, rff_position_encoder.py:41-137) at full size."""
    d, _ = H.load("dense_posenc_full_ad")
    _check_case(d, H.full_dense_posenc_sd(), H.FULL_DENSE_SPEC)


def test_euler_maruyama():
    d, sd = H.load("euler_maruyama")
    cm, cs, vm, vs = fo.euler_maruyama_dist(sd, d["atom_types"], d["x_coords"], d["x_velocs"], d["x_forces"])
    for a, b in ((cm, "coord_mean"), (cs, "coord_std"), (vm, "veloc_mean"), (vs, "veloc_std")):
        assert H.rel_err(a, d[b]) < TOL


def test_forward_reverse_roundtrip():
    """Size-independent property: flow(reverse) then flow(forward) returns the latents and the
    two log-densities agree (reference self-consistency 4.6e-5 abs)."""
    d, _ = H.load("kernel_full_ad")
    sd, spec = H.full_kernel_sd(), H.FULL_KERNEL_SPEC
    yc, yv, lp = fo.conditional_sample_with_logp(sd, spec, d["atom_types"], d["x_coords"], d["x_velocs"],
                                                 d["masked"], d["z_coords"][:2], d["z_velocs"][:2])
    ll = fo.log_likelihood(sd, spec, d["atom_types"].repeat(2, 1), d["x_coords"].repeat(2, 1, 1),
                           d["x_velocs"].repeat(2, 1, 1), yc.squeeze(1), yv.squeeze(1), d["masked"].repeat(2, 1))
    assert (ll - lp.squeeze(1)).abs().max() < 1e-3
