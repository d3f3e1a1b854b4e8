"""Host-side weight packing: reference-named state_dict -> the flat fp32 "raw layout" that
libtimewarp_hip.so reads (order documented in include/timewarp_hip.h and mirrored from
csrc/tw_kernels.hip::raw_layout).  Pure torch-CPU host logic, testable without a GPU."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

from ._lib import FlowDesc

KERNEL, DENSE = 0, 1
# Pseudo-key of the raw layout's lengthscale block [2, H]: row 0 is used by the forward pass
# (log_likelihood), row 1 by the reverse pass (sampling).  The reference computes the attention scores
# once per flow call and reuses them in all 48 attention layers - its cache key ignores the lengthscales
# (model_constructor.py:192-195) - so a flow call uses the lengthscales of the attention layer it
# evaluates FIRST: chain[0].scale_transformer.encoder_layers[0] going forward, chain[n-1]'s going in
# reverse.  For attention_type "kernel" every layer holds the same buffer and the two rows are equal;
# for "learnable_kernel" they are exp(log_lengthscales) of those two layers.
LENGTHSCALES = "@lengthscales"


@dataclass(frozen=True)
class FlowDims:
    """Hyper-parameters of one flow; `to_desc()` gives the C struct."""

    variant: int
    n_coupling: int
    n_layers: int
    d_model: int
    d_ff: int
    d_hidden: int
    d_emb: int
    n_heads: int
    d_rff: int = 0
    n_elements: int = 5
    pos_mod2: int = 0
    displacement: bool = True
    ignore_cond_velocity: bool = False
    normalise: bool = True
    ln_eps: float = 1e-5
    learnable_lengthscales: bool = False  # attention_type "learnable_kernel" (host-side only, see LENGTHSCALES)
    cheb_order: int = 0                   # attention_type "chebyshev_kernel": order of the rational Chebyshev basis
    cheb_force_zero: bool = False         # force_asymptotic_zero

    @property
    def d_in(self) -> int:
        return self.d_emb + 9 + (self.d_rff if self.variant == DENSE else 0)

    def to_desc(self) -> FlowDesc:
        return FlowDesc(
            self.variant, self.n_coupling, self.n_layers, self.d_model, self.d_ff, self.d_hidden, self.d_emb,
            self.n_heads, self.d_rff, self.n_elements, self.pos_mod2, int(self.displacement),
            int(self.ignore_cond_velocity), int(self.normalise), self.ln_eps, self.cheb_order, int(self.cheb_force_zero),
        )


def raw_entries(d: FlowDims) -> List[Tuple[str, Tuple[int, ...]]]:
    """(state_dict key, shape) in raw-layout order.  Keys follow SURVEY.md section 8b."""
    dm, ff, hid, H = d.d_model, d.d_ff, d.d_hidden, d.n_heads
    out: List[Tuple[str, Tuple[int, ...]]] = [("flow.atom_embedder.weight", (d.n_elements, d.d_emb))]
    if d.variant == KERNEL:
        out.append((LENGTHSCALES, (2, H)))
    out.append(("coords_prior_log_scale", ()))
    out.append(("velocs_prior_log_scale", ()))
    for c in range(d.n_coupling):
        if d.variant == DENSE and d.d_rff > 0:
            out.append((f"flow.chain.{c}.position_encoder.gaussian_vectors", (3, d.d_rff // 2)))
        for net in ("scale_transformer", "shift_transformer"):
            p = f"flow.chain.{c}.{net}"
            out += [
                (f"{p}.in_mlp._layers.0.weight", (hid, d.d_in)),
                (f"{p}.in_mlp._layers.0.bias", (hid,)),
                (f"{p}.in_mlp._layers.2.weight", (dm, hid)),
                (f"{p}.in_mlp._layers.2.bias", (dm,)),
            ]
            for l in range(d.n_layers):
                if d.variant == KERNEL:
                    q = f"{p}.encoder_layers.{l}"
                    out += [
                        (f"{q}.self_attn.values_proj.weight", (H * dm, dm)),
                        (f"{q}.self_attn.attention._out_projection.weight", (dm, H * dm)),
                    ]
                    if d.cheb_order > 0:
                        out.append((f"{q}.self_attn.attention.cheb_coeffs", (H, d.cheb_order)))
                else:
                    q = f"{p}.transformer.layers.{l}"
                    out += [
                        (f"{q}.self_attn.in_proj_weight", (3 * dm, dm)),
                        (f"{q}.self_attn.in_proj_bias", (3 * dm,)),
                        (f"{q}.self_attn.out_proj.weight", (dm, dm)),
                        (f"{q}.self_attn.out_proj.bias", (dm,)),
                    ]
                out += [
                    (f"{q}.linear1.weight", (ff, dm)),
                    (f"{q}.linear1.bias", (ff,)),
                    (f"{q}.linear2.weight", (dm, ff)),
                    (f"{q}.linear2.bias", (dm,)),
                    (f"{q}.norm1.weight", (dm,)),
                    (f"{q}.norm1.bias", (dm,)),
                    (f"{q}.norm2.weight", (dm,)),
                    (f"{q}.norm2.bias", (dm,)),
                ]
            out += [
                (f"{p}.out_mlp._layers.0.weight", (hid, dm)),
                (f"{p}.out_mlp._layers.0.bias", (hid,)),
                (f"{p}.out_mlp._layers.2.weight", (3, hid)),
                (f"{p}.out_mlp._layers.2.bias", (3,)),
            ]
    return out


def raw_numel(d: FlowDims) -> int:
    n = 0
    for _, shape in raw_entries(d):
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


def pack_raw(state_dict: Dict[str, torch.Tensor], d: FlowDims) -> torch.Tensor:
    """Concatenate the state_dict into the raw layout (fp32, CPU).  Raises KeyError / ValueError on
    a missing key or a shape mismatch, so a wrong checkpoint fails loudly."""
    parts = []
    for key, shape in raw_entries(d):
        if key == LENGTHSCALES:
            att = "flow.chain.{}.scale_transformer.encoder_layers.0.self_attn.attention."
            if d.learnable_lengthscales:
                rows = [torch.exp(state_dict[att.format(c) + "log_lengthscales"].detach().float()) for c in (0, d.n_coupling - 1)]
            else:
                rows = [state_dict[att.format(0) + "lengthscales"].detach().float()] * 2
            t = torch.stack([r.cpu() for r in rows])
        else:
            t = state_dict[key]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{key}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        parts.append(t.detach().to(device="cpu", dtype=torch.float32).reshape(-1))
    return torch.cat(parts)


def strip_module_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """DeepSpeed / LossWrapper checkpoints carry `module.` prefixes (losses.py:247-258)."""
    out = {}
    for k, v in sd.items():
        while k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    return out
