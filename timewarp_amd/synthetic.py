"""Synthetic inputs for benchmarks and smoke runs: the alanine-dipeptide
topology and name-seeded random weights.  No trained checkpoint or trajectory exists offline."""
from __future__ import annotations

import math
from typing import Dict

import torch

from .forcefield import AD_ATOM_NAMES, alanine_dipeptide_masses

# coordinates (Angstrom) of simulation/testdata/alanine-dipeptide.pdb -- 22-atom ACE-ALA-NME, ideal geometry
AD_COORDS_ANGSTROM = [
    [2.000, 1.000, -0.000], [2.000, 2.090, 0.000], [1.486, 2.454, 0.890], [1.486, 2.454, -0.890],
    [3.427, 2.641, -0.000], [4.391, 1.877, -0.000], [3.555, 3.970, -0.000], [2.733, 4.556, -0.000],
    [4.853, 4.614, -0.000], [5.408, 4.316, 0.890], [5.661, 4.221, -1.232], [5.123, 4.521, -2.131],
    [6.630, 4.719, -1.206], [5.809, 3.141, -1.241], [4.713, 6.129, 0.000], [3.601, 6.653, 0.000],
    [5.846, 6.835, 0.000], [6.737, 6.359, -0.000], [5.846, 8.284, 0.000], [4.819, 8.648, 0.000],
    [6.360, 8.648, 0.890], [6.360, 8.648, -0.890],
]
_VOCAB = {"C": 0, "H": 1, "N": 2, "O": 3, "S": 4}


def alanine_dipeptide_state():
    """(atom_types int64 [22], coords nm float32 [22,3], masses float32 [22])."""
    coords = torch.tensor(AD_COORDS_ANGSTROM, dtype=torch.float32) * 0.1
    types = torch.tensor([_VOCAB[n[0]] for n in AD_ATOM_NAMES], dtype=torch.int64)
    return types, coords, torch.from_numpy(alanine_dipeptide_masses())


def _name_seed(name: str, base: int) -> int:
    h = 1469598103934665603  # FNV-1a over the parameter name
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ base) & 0x7FFFFFFF


def synth_tensor(name: str, shape, base_seed: int = 0) -> torch.Tensor:
    """N(0,1)/sqrt(fan_in) for matrices, N(0,0.1^2) for vectors/scalars, LayerNorm weights around 1."""
    g = torch.Generator().manual_seed(_name_seed(name, base_seed))
    shape = tuple(shape)
    if len(shape) == 2:
        return torch.randn(shape, generator=g) / math.sqrt(shape[1])
    if len(shape) == 0:
        return torch.randn((), generator=g) * 0.1
    t = torch.randn(shape, generator=g) * 0.1
    if ".norm" in name and name.endswith("weight"):
        t = t + 1.0
    return t


def synth_state_dict(template: Dict[str, torch.Tensor], base_seed: int = 0, calibrated: bool = False,
                     coords_log_scale: float = -5.0, velocs_log_scale: float = -5.0) -> Dict[str, torch.Tensor]:
    """Fill a state_dict (names/shapes from `template`, persistent buffers kept) with the
    name-seeded recipe.  calibrated=True: prior log-scales -5 and the last out_mlp layer zeroed, so
    proposals are ~7e-3 nm perturbations with a non-degenerate MH acceptance while every
    kernel still does its full work."""
    out = {}
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


def kernel_transformer_nvp_config():
    """configs/kernel_transformer_nvp.yaml:15-30 as a ModelConfig."""
    from .model_configs import model_config_from_dict

    return model_config_from_dict({
        "model_type": "custom_attention_transformer_nvp",
        "custom_transformer_nvp_config": {
            "atom_embedding_dim": 32, "latent_mlp_hidden_dims": [256], "num_coupling_layers": 8,
            "num_transformer_layers": 3,
            "encoder_layer_config": {"d_model": 128, "dim_feedforward": 2048, "num_heads": 6, "dropout": 0,
                                     "attention_type": "kernel", "lengthscales": [0.1, 0.2, 0.5, 0.7, 1.0, 1.2],
                                     "normalise_kernel_values": True},
        },
    })


def transformer_nvp_config():
    """configs/transformer_nvp.yaml:14-25 (dense softmax attention) as a ModelConfig."""
    from .model_configs import model_config_from_dict

    return model_config_from_dict({
        "model_type": "transformer_nvp",
        "transformer_nvp_config": {
            "atom_embedding_dim": 32, "transformer_hidden_dim": 128, "latent_mlp_hidden_dims": [256],
            "num_coupling_layers": 8, "num_transformer_layers": 3,
            "transformer_config": {"n_head": 8, "dim_feedforward": 2048, "dropout": 0},
        },
    })
