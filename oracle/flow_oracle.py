"""CPU oracle for the Timewarp conditional-flow hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) *restatement* of the reference algorithm for
SURVEY.md section 8 rows a2-a13 / a19.  It is the checker that `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` call; nothing under
`timewarp_amd/` may import it (the product path is HIP only and fails loudly when the
extension is missing).

Pinning: `oracle/gen_golden.py` imports the real reference from /root/reference (in the
build container only) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks every function below against those vectors.  The oracle is therefore *pinned* for
the flow (both attention variants).  See `energy_oracle.c` for the energy.

Every function cites the reference file:line (relative to /root/reference) it follows.
All tensors are torch CPU tensors; weights come from a reference-named ``state_dict``
(keys as in SURVEY.md section 8b, an optional ``module.`` prefix is stripped by the caller).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]


@dataclass
class FlowSpec:
    """The hyper-parameters the reference reads from its config dataclasses
    (model_configs.py:51-76, custom_attention_encoder.py:126-137, transformer_block.py:11-15)."""

    variant: str = "kernel"  # "kernel" (custom_attention_transformer_nvp) | "dense" (transformer_nvp)
    num_coupling_layers: int = 8
    num_transformer_layers: int = 3
    position_layer_index_mod_2: int = 0
    n_head: int = 8  # dense variant only
    layer_norm_eps: float = 1e-5
    use_displacement_as_target: bool = True
    ignore_conditional_velocity: bool = False
    normalise_kernel_values: bool = True  # a config field the reference never acts on (scores are always normalised)
    attention_type: str = "kernel"  # "kernel" | "learnable_kernel" | "chebyshev_kernel" (kernel_attention.py:159-339)
    force_asymptotic_zero: bool = False  # chebyshev_kernel only


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------


def centre_of_mass(coords: Tensor, masked: Optional[Tensor]) -> Tensor:
    """utils/molecule_utils.py:15-29 -- masked arithmetic mean (not mass weighted)."""
    if masked is not None:
        inv = ~masked
        coords = inv.unsqueeze(-1) * coords
        n = inv.sum(dim=-1, keepdim=True).unsqueeze(-1)
    else:
        n = coords.shape[-2]
    return coords.sum(dim=-2, keepdim=True) / n


def mlp(sd: StateDict, prefix: str, x: Tensor) -> Tensor:
    """modules/layers/mlp.py:6-26 -- Linear / SiLU stack; layers live at `_layers.{0,2,..}`."""
    idx = 0
    while f"{prefix}._layers.{idx}.weight" in sd:
        x = F.linear(x, sd[f"{prefix}._layers.{idx}.weight"], sd[f"{prefix}._layers.{idx}.bias"])
        if f"{prefix}._layers.{idx + 2}.weight" in sd:
            x = F.silu(x)
        idx += 2
    return x


def cdist_direct(a: Tensor, b: Tensor) -> Tensor:
    """Direct-difference Euclidean distance (what torch.cdist does for <=25 points)."""
    d = a.unsqueeze(-2) - b.unsqueeze(-3)
    return d.pow(2).sum(-1).sqrt()


def cdist_mm(a: Tensor, b: Tensor) -> Tensor:
    """The |a|^2+|b|^2-2ab formulation torch.cdist switches to above 25 points
    (kernel_attention.py:76,98-102 passes 'use_mm_for_euclid_dist_if_necessary')."""
    an = a.pow(2).sum(-1, keepdim=True)
    bn = b.pow(2).sum(-1, keepdim=True)
    a_ = torch.cat([a * -2, an, torch.ones_like(an)], -1)
    b_ = torch.cat([b, torch.ones_like(bn), bn], -1)
    return a_.matmul(b_.transpose(-1, -2)).clamp_min(0).sqrt()


def chebyshev_basis(scaled: Tensor, coeffs: Tensor, force_asymptotic_zero: bool) -> Tensor:
    """kernel_attention.py:12-66: sum_c coeffs[h,c] R_c(scaled^2) with the rational Chebyshev functions
    R_0 = 1, R_1 = (x-1)/(x+1), R_{n+1} = 2 R_1 R_n - R_{n-1};  coeffs [H, order], scaled [B,H,V,V]."""
    if force_asymptotic_zero:
        coeffs = coeffs - coeffs.mean(dim=1, keepdim=True)
    x = scaled**2
    order = coeffs.shape[1]
    rf = (x - 1.0) / (x + 1.0)
    terms = [torch.ones_like(x)]
    if order >= 2:
        terms.append(rf)
    for _ in range(2, order):
        terms.append(2.0 * rf * terms[-1] - terms[-2])
    return torch.einsum("bhcqm,hc->bhqm", torch.stack(terms, dim=2), coeffs)


def kernel_scores(x_coords: Tensor, masked: Tensor, lengthscales: Tensor, normalise: bool = True,
                  cheb_coeffs: Optional[Tensor] = None, force_asymptotic_zero: bool = False) -> Tensor:
    """modules/layers/kernel_attention.py:69-121.

    A[b,h,q,m] = basis(|x_q-x_m|/l_h), basis(s) = exp(-s^2) or the Chebyshev expansion; masked keys -> 0;
    A /= sum_m |A| + 1e-5.  Returns [B,H,V,V]."""
    dist = torch.cdist(x_coords, x_coords, compute_mode="use_mm_for_euclid_dist_if_necessary")
    scaled = dist.unsqueeze(-3).expand(-1, len(lengthscales), -1, -1) / lengthscales[None, :, None, None]
    a = torch.exp(-(scaled**2)) if cheb_coeffs is None else chebyshev_basis(scaled, cheb_coeffs, force_asymptotic_zero)
    a = a.masked_fill(masked[:, None, None, :], 0.0)
    if normalise:
        a = a / (torch.abs(a).sum(dim=-1, keepdim=True) + 1e-5)
    return a


# --------------------------------------------------------------------------------------
# kernel-attention encoder (custom_attention_transformer_nvp)
# --------------------------------------------------------------------------------------


def kernel_self_attention(sd: StateDict, prefix: str, h: Tensor, scores: Tensor) -> Tensor:
    """kernel_self_attention.py:29-48 + kernel_attention.py:185-214 (+124-156).

    values = h W_v^T -> [B,V,H,Dv]; attended = scores @ values^T(1,2); heads concatenated;
    bias-free out projection."""
    wv = sd[f"{prefix}.values_proj.weight"]
    wo = sd[f"{prefix}.attention._out_projection.weight"]
    n_heads = scores.shape[1]
    values = F.linear(h, wv)
    values = values.reshape(values.shape[0], values.shape[1], n_heads, -1)
    attended = scores @ values.transpose(1, 2)  # [B,H,V,Dv]
    flat = attended.transpose(-2, -3).reshape(attended.shape[0], attended.shape[2], -1)
    return F.linear(flat, wo)


def encoder_layer_tail(sd: StateDict, prefix: str, h: Tensor, attn_out: Tensor, eps: float) -> Tensor:
    """custom_attention_encoder.py:109-114 -- post-norm residual + ReLU FFN (dropout=0)."""
    d = h.shape[-1]
    h = h + attn_out
    h = F.layer_norm(h, (d,), sd[f"{prefix}.norm1.weight"], sd[f"{prefix}.norm1.bias"], eps)
    ff = F.linear(
        F.relu(F.linear(h, sd[f"{prefix}.linear1.weight"], sd[f"{prefix}.linear1.bias"])),
        sd[f"{prefix}.linear2.weight"],
        sd[f"{prefix}.linear2.bias"],
    )
    h = h + ff
    return F.layer_norm(h, (d,), sd[f"{prefix}.norm2.weight"], sd[f"{prefix}.norm2.bias"], eps)


def kernel_netblock(
    sd: StateDict, prefix: str, u: Tensor, scores: Optional[Tensor], spec: FlowSpec, trace: Optional[list] = None,
    positions: Optional[Tensor] = None, masked: Optional[Tensor] = None,
) -> Tensor:
    """custom_transformer_block.py:46-82: in_mlp -> L encoder layers -> out_mlp.  `scores` is the per-call matrix
    (Gaussian bases); for chebyshev_kernel it is None and every layer computes its own from `positions`: there the
    reference's cache key contains the per-layer basis-function closure (kernel_attention.py:329-333), so nothing
    is shared between layers."""
    h = mlp(sd, f"{prefix}.in_mlp", u)
    if trace is not None:
        trace.append(("in_mlp", h))
    for l in range(spec.num_transformer_layers):
        p = f"{prefix}.encoder_layers.{l}"
        layer_scores = scores
        if layer_scores is None:
            att = f"{p}.self_attn.attention."
            layer_scores = kernel_scores(positions, masked, sd[att + "lengthscales"], True,
                                         sd[att + "cheb_coeffs"], spec.force_asymptotic_zero)
        a = kernel_self_attention(sd, f"{p}.self_attn", h, layer_scores)
        h = encoder_layer_tail(sd, p, h, a, spec.layer_norm_eps)
        if trace is not None:
            trace.append((f"enc{l}", h))
    out = mlp(sd, f"{prefix}.out_mlp", h)
    if trace is not None:
        trace.append(("out_mlp", out))
    return out


# --------------------------------------------------------------------------------------
# dense softmax encoder (transformer_nvp) -- nn.TransformerEncoderLayer restated
# --------------------------------------------------------------------------------------


def dense_self_attention(sd: StateDict, prefix: str, h: Tensor, masked: Tensor, n_head: int) -> Tensor:
    """transformer_block.py:36-46 -> torch.nn.MultiheadAttention (batch_first, key padding mask).

    q,k,v = split(h W_in^T + b_in); softmax(q k^T / sqrt(dh) with -inf on padded keys) v; out_proj."""
    b, v, d = h.shape
    dh = d // n_head
    qkv = F.linear(h, sd[f"{prefix}.in_proj_weight"], sd[f"{prefix}.in_proj_bias"])
    q, k, val = qkv.split(d, dim=-1)
    q = q.reshape(b, v, n_head, dh).transpose(1, 2)
    k = k.reshape(b, v, n_head, dh).transpose(1, 2)
    val = val.reshape(b, v, n_head, dh).transpose(1, 2)
    logits = (q / math.sqrt(dh)) @ k.transpose(-1, -2)
    logits = logits.masked_fill(masked[:, None, None, :], float("-inf"))
    attn = torch.softmax(logits, dim=-1)
    out = (attn @ val).transpose(1, 2).reshape(b, v, d)
    return F.linear(out, sd[f"{prefix}.out_proj.weight"], sd[f"{prefix}.out_proj.bias"])


def rff_encode(coords: Tensor, gaussian_vectors: Tensor) -> Tensor:
    """rff_position_encoder.py:41-64."""
    n = gaussian_vectors.shape[1]
    ips = coords @ gaussian_vectors
    if n == 0:
        return ips
    return math.sqrt(1.0 / n) * torch.cat((torch.cos(ips), torch.sin(ips)), dim=-1)


def dense_netblock(sd: StateDict, prefix: str, u: Tensor, masked: Tensor, spec: FlowSpec) -> Tensor:
    """transformer_block.py:58-72."""
    h = mlp(sd, f"{prefix}.in_mlp", u)
    for l in range(spec.num_transformer_layers):
        p = f"{prefix}.transformer.layers.{l}"
        a = dense_self_attention(sd, f"{p}.self_attn", h, masked, spec.n_head)
        h = encoder_layer_tail(sd, p, h, a, spec.layer_norm_eps)
    return mlp(sd, f"{prefix}.out_mlp", h)


# --------------------------------------------------------------------------------------
# coupling layers and the flow
# --------------------------------------------------------------------------------------


def scale_and_shift(
    sd: StateDict,
    spec: FlowSpec,
    c: int,
    z_coords: Tensor,
    z_velocs: Tensor,
    x_features: Tensor,
    x_coords: Tensor,
    x_velocs: Tensor,
    masked: Tensor,
    scores: Optional[Tensor],
) -> Tuple[Tensor, Tensor]:
    """custom_transformer_nvp.py:44-93 / transformer_nvp.py:58-97.
    Even (mod position_layer_index_mod_2) layers transform positions (model_constructor.py:169)."""
    positions = c % 2 == spec.position_layer_index_mod_2
    z_other = z_velocs if positions else z_coords
    parts = [x_features, x_coords, x_velocs, z_other]
    pre = f"flow.chain.{c}"
    if spec.variant == "dense":
        parts.append(rff_encode(x_coords, sd[f"{pre}.position_encoder.gaussian_vectors"]))
    u = torch.cat(parts, dim=-1)
    if spec.variant == "kernel":
        s = kernel_netblock(sd, f"{pre}.scale_transformer", u, scores, spec, positions=x_coords, masked=masked)
        t = kernel_netblock(sd, f"{pre}.shift_transformer", u, scores, spec, positions=x_coords, masked=masked)
    else:
        s = dense_netblock(sd, f"{pre}.scale_transformer", u, masked, spec)
        t = dense_netblock(sd, f"{pre}.shift_transformer", u, masked, spec)
    return torch.exp(s), t


def flow_pass(
    sd: StateDict,
    spec: FlowSpec,
    z_coords: Tensor,
    z_velocs: Tensor,
    x_features: Tensor,
    x_coords: Tensor,
    x_velocs: Tensor,
    masked: Tensor,
    delta_logp: Tensor,
    reverse: bool,
) -> Tuple[Tensor, Tensor, Tensor]:
    """flow.py:51-103 (layer order) + layers/nvp.py:22-183 (affine coupling + log-det)."""
    scores = None
    if spec.variant == "kernel" and spec.attention_type != "chebyshev_kernel":
        # one score matrix per flow call: the reference's Cache makes all 48 encoder layers
        # share it (model_constructor.py:192-195, flow.py:188,299).
        # The cache key ignores the lengthscales (keyword transform Returns(0)), so the scores are those of
        # the attention layer evaluated FIRST in this call: chain[0]'s scale net going forward, chain[n-1]'s
        # going in reverse (custom_transformer_nvp.py:44-93 evaluates the scale net before the shift net).
        first = spec.num_coupling_layers - 1 if reverse else 0
        att = f"flow.chain.{first}.scale_transformer.encoder_layers.0.self_attn.attention."
        if spec.attention_type == "learnable_kernel":
            ls = torch.exp(sd[att + "log_lengthscales"])  # kernel_attention.py:251-252
        else:
            ls = sd[att + "lengthscales"]
        # always normalised: KernelAttention.forward (kernel_attention.py:197-206) does not pass its
        # normalise_kernel_values on, and compute_kernel_attention_scores defaults to True (:75)
        scores = kernel_scores(x_coords, masked, ls, True)
    order = range(spec.num_coupling_layers)
    if reverse:
        order = reversed(order)
    keep = ~masked[:, :, None]
    for c in order:
        scale, shift = scale_and_shift(
            sd, spec, c, z_coords, z_velocs, x_features, x_coords, x_velocs, masked, scores
        )
        log_scales = torch.log(scale) * keep
        positions = c % 2 == spec.position_layer_index_mod_2
        if reverse:
            logdet = -torch.sum(log_scales, dim=(-1, -2))  # nvp.py:175-176
            if positions:
                z_coords = (z_coords - shift) / scale  # nvp.py:178-181
            else:
                z_velocs = (z_velocs - shift) / scale
        else:
            logdet = torch.sum(log_scales, dim=(-1, -2))  # nvp.py:127-128
            if positions:
                z_coords = z_coords * scale + shift  # nvp.py:130-133
            else:
                z_velocs = z_velocs * scale + shift
        delta_logp = delta_logp - logdet  # nvp.py:86
    return z_coords, z_velocs, delta_logp


def _normal_log_prob(z: Tensor, log_scale: Tensor) -> Tensor:
    """torch.distributions.Normal(0, exp(log_scale)).log_prob(z) spelled out."""
    scale = torch.exp(log_scale)
    var = scale**2
    return -(z**2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))


def log_likelihood(
    sd: StateDict,
    spec: FlowSpec,
    atom_types: Tensor,
    x_coords: Tensor,
    x_velocs: Tensor,
    y_coords: Tensor,
    y_velocs: Tensor,
    masked: Tensor,
) -> Tensor:
    """modules/model_wrappers/flow.py:131-215."""
    if spec.ignore_conditional_velocity:
        x_velocs = torch.zeros_like(x_velocs)
    resid = y_coords - x_coords if spec.use_displacement_as_target else y_coords
    x_coords = x_coords - centre_of_mass(x_coords, masked)
    feats = F.embedding(atom_types, sd["flow.atom_embedder.weight"])
    delta = torch.zeros(x_coords.shape[0])
    z_c, z_v, delta = flow_pass(sd, spec, resid, y_velocs, feats, x_coords, x_velocs, masked, delta, False)
    keep = ~masked[:, :, None]
    lp = (keep * _normal_log_prob(z_c, sd["coords_prior_log_scale"])).sum(dim=(-1, -2))
    lp = lp + (keep * _normal_log_prob(z_v, sd["velocs_prior_log_scale"])).sum(dim=(-1, -2))
    return lp - delta


def conditional_sample_with_logp(
    sd: StateDict,
    spec: FlowSpec,
    atom_types: Tensor,
    x_coords: Tensor,
    x_velocs: Tensor,
    masked: Tensor,
    z_coords: Tensor,  # [S,B,V,3] latent noise ALREADY scaled by exp(coords_prior_log_scale)
    z_velocs: Tensor,  # [S,B,V,3]
) -> Tuple[Tensor, Tensor, Tensor]:
    """modules/model_wrappers/flow.py:242-336 with the noise passed in explicitly
    (the reference draws it at :274-275 with Normal.rsample((S,)), coords first)."""
    s, b = z_coords.shape[0], x_coords.shape[0]
    if spec.ignore_conditional_velocity:
        x_velocs = torch.zeros_like(x_velocs)
    com = centre_of_mass(x_coords, masked)
    xc = x_coords - com
    zc = z_coords.reshape(-1, z_coords.shape[-2], 3)
    zv = z_velocs.reshape(-1, z_velocs.shape[-2], 3)
    feats = F.embedding(atom_types, sd["flow.atom_embedder.weight"])
    delta = torch.zeros(b).repeat(s)
    rc, rv, delta = flow_pass(
        sd,
        spec,
        zc,
        zv,
        feats.repeat(s, 1, 1),
        xc.repeat(s, 1, 1),
        x_velocs.repeat(s, 1, 1),
        masked.repeat(s, 1),
        delta,
        True,
    )
    x_back = (xc + com).repeat(s, 1, 1)  # flow.py:303-304 (x - com + com: not bit-identical to x)
    yc = x_back + rc if spec.use_displacement_as_target else rc
    yc = yc.reshape(s, b, yc.shape[-2], 3)
    yv = rv.reshape(s, b, rv.shape[-2], 3)
    # flow.py:326 multiplies a [B,V,1] mask into [S*B,V,3]; this only broadcasts for B==1 or S==1
    keep = ~masked[:, :, None]
    lp = (keep * _normal_log_prob(zc, sd["coords_prior_log_scale"])).sum(dim=(-1, -2))
    lp = lp + (keep * _normal_log_prob(zv, sd["velocs_prior_log_scale"])).sum(dim=(-1, -2))
    return yc, yv, (lp + delta).reshape(s, b)


def draw_latents(sd: StateDict, num_samples: int, shape: Tuple[int, int, int], gen: torch.Generator):
    """Noise in the reference's order (flow.py:274-275): z_coords [S,B,V,3] then z_velocs, each
    eps * exp(log_scale) with eps ~ N(0,1) from the given CPU generator (rsample = loc + eps*scale)."""
    b, v, _ = shape
    eps_c = torch.randn((num_samples, b, v, 3), generator=gen)
    eps_v = torch.randn((num_samples, b, v, 3), generator=gen)
    return eps_c * torch.exp(sd["coords_prior_log_scale"]), eps_v * torch.exp(sd["velocs_prior_log_scale"])


# --------------------------------------------------------------------------------------
# cfg-1 plumbing model
# --------------------------------------------------------------------------------------

EM_K_B = 1.380649e-23 * 1e-3 * 6.02214076e23  # baselines.py:176
EM_TEMPERATURE = 310
EM_GAMMA = 0.3


def euler_maruyama_dist(sd: StateDict, atom_types: Tensor, x_coords: Tensor, x_velocs: Tensor, x_forces: Tensor,
                        step_width_init: float = 1.0):
    """modules/baselines.py:254-296 -- means and stds of the two Normals."""
    delta_t = step_width_init * 0.5 * 1e-3
    coord_stds = torch.exp(sd["atom_coord_std_params"][atom_types])
    masses = torch.exp(sd["atom_mass_params"][atom_types])
    f = torch.exp(sd["delta_t_factor_param"])
    coord_mean = x_coords + delta_t * f * x_velocs
    force_term = (x_forces / masses[:, :, None]) * delta_t * f
    friction = -EM_GAMMA * x_velocs * delta_t * f
    veloc_mean = x_velocs + force_term + friction
    veloc_stds = torch.sqrt(2.0 * EM_GAMMA * EM_K_B * EM_TEMPERATURE * delta_t * f / masses)
    veloc_stds = veloc_stds + torch.exp(sd["atom_veloc_std_params"][atom_types])
    return coord_mean, coord_stds[:, :, None].repeat(1, 1, 3), veloc_mean, veloc_stds[:, :, None].repeat(1, 1, 3)


# --------------------------------------------------------------------------------------
# name-seeded synthetic weights (shared recipe: the golden generator fills the REFERENCE
# model with these, tests/bench regenerate them, so full-size weights never hit the repo)
# --------------------------------------------------------------------------------------


def _name_seed(name: str, base: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ base) & 0x7FFFFFFF


def synth_tensor(name: str, shape, base_seed: int = 0) -> Tensor:
    """N(0,1) * fan_in^-1/2 for matrices, small N(0,0.1^2) offsets for vectors; LayerNorm weight ~ 1."""
    g = torch.Generator().manual_seed(_name_seed(name, base_seed))
    shape = tuple(shape)
    if name.endswith("lengthscales") or name.endswith("gaussian_vectors"):
        raise ValueError("buffers are not synthesised")
    if len(shape) == 2:
        return torch.randn(shape, generator=g) / math.sqrt(shape[1])
    if len(shape) == 0:
        return torch.randn((), generator=g) * 0.1
    t = torch.randn(shape, generator=g) * 0.1
    if ".norm" in name and name.endswith("weight"):
        t = t + 1.0
    return t


def synth_state_dict(template: Dict[str, Tensor], base_seed: int = 0, calibrated: bool = False,
                     coords_log_scale: float = -5.0, velocs_log_scale: float = -5.0) -> StateDict:
    """Fill a state_dict with the name-seeded recipe.  `template` supplies names/shapes (and the
    values of persistent buffers, which are kept).  calibrated=True applies SURVEY section 8d's
    throughput calibration: prior log-scales = -5 and the last out_mlp layer zeroed (s=1, t=0)."""
    out: StateDict = {}
    for k, v in template.items():
        if k.endswith("lengthscales") or k.endswith("gaussian_vectors"):
            out[k] = v.clone()
        else:
            out[k] = synth_tensor(k, v.shape, base_seed).to(v.dtype)
    if calibrated:
        for k in list(out):
            if k == "coords_prior_log_scale":
                out[k] = torch.tensor(float(coords_log_scale))
            if k == "velocs_prior_log_scale":
                out[k] = torch.tensor(float(velocs_log_scale))
            if ".out_mlp._layers.2." in k:
                out[k] = torch.zeros_like(out[k])
    return out


def make_template(
    spec: FlowSpec,
    atom_embedding_dim: int = 32,
    d_model: int = 128,
    dim_feedforward: int = 2048,
    mlp_hidden: Tuple[int, ...] = (256,),
    lengthscales: Tuple[float, ...] = (0.1, 0.2, 0.5, 0.7, 1.0, 1.2),
    rff_dim: int = 0,
    n_elements: int = 5,
    cheb_order: int = 0,
) -> StateDict:
    """Names and shapes of the reference state_dict, zeros except buffers.
    kernel: value_dim = d_model, heads = len(lengthscales) (custom_attention_encoder.py:170-189)."""
    t: StateDict = {}
    t["coords_prior_log_scale"] = torch.zeros(())
    t["velocs_prior_log_scale"] = torch.zeros(())
    t["flow.atom_embedder.weight"] = torch.zeros(n_elements, atom_embedding_dim)
    in_dim = atom_embedding_dim + 9 + (rff_dim if spec.variant == "dense" else 0)

    def add_mlp(prefix: str, din: int, dout: int):
        dims = [din, *mlp_hidden, dout]
        for i in range(len(dims) - 1):
            t[f"{prefix}._layers.{2 * i}.weight"] = torch.zeros(dims[i + 1], dims[i])
            t[f"{prefix}._layers.{2 * i}.bias"] = torch.zeros(dims[i + 1])

    h = len(lengthscales)
    for c in range(spec.num_coupling_layers):
        if spec.variant == "dense":
            t[f"flow.chain.{c}.position_encoder.gaussian_vectors"] = torch.zeros(3, rff_dim // 2)
        for net in ("scale_transformer", "shift_transformer"):
            p = f"flow.chain.{c}.{net}"
            add_mlp(f"{p}.in_mlp", in_dim, d_model)
            for l in range(spec.num_transformer_layers):
                if spec.variant == "kernel":
                    q = f"{p}.encoder_layers.{l}"
                    t[f"{q}.self_attn.values_proj.weight"] = torch.zeros(h * d_model, d_model)
                    if spec.attention_type == "chebyshev_kernel":  # kernel_attention.py:300-304
                        t[f"{q}.self_attn.attention.cheb_coeffs"] = torch.zeros(h, cheb_order)
                    t[f"{q}.self_attn.attention.lengthscales"] = torch.tensor(lengthscales, dtype=torch.float32)
                    t[f"{q}.self_attn.attention._out_projection.weight"] = torch.zeros(d_model, h * d_model)
                else:
                    q = f"{p}.transformer.layers.{l}"
                    t[f"{q}.self_attn.in_proj_weight"] = torch.zeros(3 * d_model, d_model)
                    t[f"{q}.self_attn.in_proj_bias"] = torch.zeros(3 * d_model)
                    t[f"{q}.self_attn.out_proj.weight"] = torch.zeros(d_model, d_model)
                    t[f"{q}.self_attn.out_proj.bias"] = torch.zeros(d_model)
                t[f"{q}.linear1.weight"] = torch.zeros(dim_feedforward, d_model)
                t[f"{q}.linear1.bias"] = torch.zeros(dim_feedforward)
                t[f"{q}.linear2.weight"] = torch.zeros(d_model, dim_feedforward)
                t[f"{q}.linear2.bias"] = torch.zeros(d_model)
                for n in ("norm1", "norm2"):
                    t[f"{q}.{n}.weight"] = torch.zeros(d_model)
                    t[f"{q}.{n}.bias"] = torch.zeros(d_model)
            add_mlp(f"{p}.out_mlp", d_model, 3)
    return t
