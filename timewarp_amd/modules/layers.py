"""Parameter containers with the reference's attribute names, so that `state_dict()` /
`load_state_dict()` round-trip reference checkpoints unchanged.

These modules hold weights only: the arithmetic of every layer below runs in
libtimewarp_hip.so (csrc/tw_netblock.hip, csrc/tw_kernels.hip).  Calling one of them directly is a
programming error and raises."""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.nn as nn


class _WeightsOnly(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} only stores weights; the computation runs in the HIP library "
            "through ConditionalFlowDensityModel (there is no eager/CPU fallback)."
        )


class MLP(_WeightsOnly):
    """Linear/SiLU stack; reference keys `_layers.{0,2,...}.{weight,bias}` (layers/mlp.py:6-26)."""

    def __init__(self, input_dim: int, out_dim: int, hidden_layer_dims: Sequence[int]):
        super().__init__()
        mods: List[nn.Module] = []
        cur = input_dim
        for h in hidden_layer_dims:
            mods += [nn.Linear(cur, h), nn.SiLU()]
            cur = h
        mods.append(nn.Linear(cur, out_dim))
        self._layers = nn.Sequential(*mods)


class KernelAttention(_WeightsOnly):
    """`lengthscales` buffer + bias-free `_out_projection` (layers/kernel_attention.py:159-183)."""

    def __init__(self, value_dim: int, output_dim: int, lengthscales: Sequence[float], normalise_kernel_values: bool):
        super().__init__()
        self.register_buffer("lengthscales", torch.tensor(list(lengthscales), dtype=torch.float32), persistent=True)
        self.normalise_kernel_values = normalise_kernel_values
        self._out_projection = nn.Linear(value_dim * len(lengthscales), output_dim, bias=False)


class LearnableLengthscaleKernelAttention(KernelAttention):
    """KernelAttention + `log_lengthscales` parameter, initialised to log(lengthscales)
    (layers/kernel_attention.py:217-252).  The `lengthscales` buffer stays in the state_dict, unused."""

    def __init__(self, value_dim: int, output_dim: int, lengthscales: Sequence[float], normalise_kernel_values: bool):
        super().__init__(value_dim, output_dim, lengthscales, normalise_kernel_values)
        self.log_lengthscales = nn.Parameter(torch.log(torch.tensor(list(lengthscales), dtype=torch.float32)))


class KernelSelfAttention(_WeightsOnly):
    """Bias-free `values_proj` + `attention` (layers/kernel_self_attention.py:12-27)."""

    def __init__(self, input_dim: int, num_heads: int, value_dim: int, attention: KernelAttention):
        super().__init__()
        self.num_heads, self.input_dim, self.value_dim = num_heads, input_dim, value_dim
        self.values_proj = nn.Linear(input_dim, num_heads * value_dim, bias=False)
        self.attention = attention


class CustomTransformerEncoderLayer(_WeightsOnly):
    """`self_attn`, `linear1/2`, `norm1/2` (layers/custom_attention_encoder.py:49-73)."""

    def __init__(self, d_model: int, self_attention: nn.Module, dim_feedforward: int, layer_norm_eps: float = 1e-5):
        super().__init__()
        self.d_model, self.dim_feedforward = d_model, dim_feedforward
        self.self_attn = self_attention
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)


class CustomAttentionTransformerBlock(_WeightsOnly):
    """`in_mlp`, `encoder_layers`, `out_mlp` (layers/custom_transformer_block.py:20-44)."""

    def __init__(self, input_dim: int, output_dim: int, mlp_hidden_layer_dims: Sequence[int],
                 transformer_encoder_layers: Sequence[CustomTransformerEncoderLayer]):
        super().__init__()
        d_model = transformer_encoder_layers[0].d_model
        self.in_mlp = MLP(input_dim, d_model, mlp_hidden_layer_dims)
        self.encoder_layers = nn.ModuleList(transformer_encoder_layers)
        self.out_mlp = MLP(d_model, output_dim, mlp_hidden_layer_dims)


class TransformerBlock(_WeightsOnly):
    """Dense variant: `in_mlp`, `transformer.layers.{l}.*`, `out_mlp` (layers/transformer_block.py:23-56).
    torch's own nn.TransformerEncoder is used as the container so the keys and the default
    initialisation are exactly those of the reference."""

    def __init__(self, input_dim: int, output_dim: int, latent_dim: int, mlp_hidden_layer_dims: Sequence[int],
                 num_transformer_layers: int, n_head: int, dim_feedforward: int, dropout: float):
        super().__init__()
        self.in_mlp = MLP(input_dim, latent_dim, mlp_hidden_layer_dims)
        self.transformer = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=latent_dim, nhead=n_head, dim_feedforward=dim_feedforward,
                                       dropout=dropout, activation="relu", batch_first=True),
            num_layers=num_transformer_layers,
        )
        self.out_mlp = MLP(latent_dim, output_dim, mlp_hidden_layer_dims)


def draw_rff_vectors(ndim: int, nsamples: int, scale_mean: float, scale_stddev: float) -> torch.Tensor:
    """Random Fourier directions with Gamma-distributed RBF scales, one scale per vector
    (layers/rff_position_encoder.py:15-38, 67-83): shape = mean*rate, rate = mean/std^2."""
    if nsamples == 0:
        return torch.zeros((ndim, 0))
    rate = scale_mean / (scale_stddev**2.0)
    shape = scale_mean * rate
    gamma = torch.distributions.Gamma(shape, rate)
    cols = []
    for _ in range(nsamples):
        rbf_scale = gamma.sample()
        cols.append(torch.normal(torch.zeros(ndim), torch.ones(ndim) / rbf_scale))
    return torch.stack(cols, dim=1)


class RFFPositionEncoder(_WeightsOnly):
    """`gaussian_vectors` buffer [3, enc/2] (layers/rff_position_encoder.py:86-118)."""

    def __init__(self, position_dim: int, encoding_dim: int, scale_mean: float, scale_stddev: float):
        super().__init__()
        assert encoding_dim % 2 == 0, "Number of encoding dimensions must be even."
        vecs = draw_rff_vectors(position_dim, encoding_dim // 2, scale_mean, scale_stddev).to(torch.float32)
        self.register_buffer("gaussian_vectors", vecs, persistent=True)


class CouplingLayer(_WeightsOnly):
    """`scale_transformer` / `shift_transformer` (+ `position_encoder` for the dense variant)
    (modules/custom_transformer_nvp.py:20-42, modules/transformer_nvp.py:18-56)."""

    def __init__(self, transformed_vars: str, scale_transformer: nn.Module, shift_transformer: nn.Module,
                 position_encoder: nn.Module = None):
        super().__init__()
        assert transformed_vars in ("positions", "velocities")
        self.transformed_vars = transformed_vars
        if position_encoder is not None:
            self.position_encoder = position_encoder
        self.scale_transformer = scale_transformer
        self.shift_transformer = shift_transformer


class ConditionalSequentialFlow(_WeightsOnly):
    """`chain` + `atom_embedder` (modules/model_wrappers/flow.py:44-49)."""

    def __init__(self, layers: Sequence[nn.Module], atom_embedder: nn.Module):
        super().__init__()
        self.atom_embedder = atom_embedder
        self.chain = nn.ModuleList(layers)
