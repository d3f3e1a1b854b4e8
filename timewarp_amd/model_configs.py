"""Config dataclasses with the reference's names and fields (model_configs.py:14-88,
modules/layers/custom_attention_encoder.py:126-137, modules/layers/transformer_block.py:11-15,
modules/layers/rff_position_encoder.py:8-12, modules/model_wrappers/flow.py:339-343), restricted to
the model families on the hot path.  A reference YAML's `model_config:` block maps 1:1 onto these
(see `model_config_from_dict`)."""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Any, Dict, List, Optional


@dataclass
class ConditionalFlowDensityConfig:
    scale_requires_grad: bool = True
    ignore_conditional_velocity: bool = False
    use_displacement_as_target: bool = True


@dataclass
class CustomAttentionEncoderLayerConfig:
    d_model: int
    dim_feedforward: int
    dropout: float
    num_heads: int
    attention_type: str
    lengthscales: Optional[List[float]] = None
    max_radius: Optional[float] = None
    normalise_kernel_values: Optional[bool] = None
    cheb_order: Optional[int] = None
    force_asymptotic_zero: Optional[bool] = None


@dataclass
class TransformerConfig:
    n_head: int = 8
    dim_feedforward: int = 2048
    dropout: float = 0.0


@dataclass
class RFFPositionEncoderConfig:
    encoding_dim: int
    scale_mean: float
    scale_stddev: float


@dataclass
class CustomAttentionTransformerNVPConfig:
    atom_embedding_dim: int
    latent_mlp_hidden_dims: List[int]
    num_coupling_layers: int
    num_transformer_layers: int
    encoder_layer_config: CustomAttentionEncoderLayerConfig
    position_layer_index_mod_2: int = 0
    conditional_flow_density: ConditionalFlowDensityConfig = field(default_factory=ConditionalFlowDensityConfig)


@dataclass
class TransformerNVPConfig:
    atom_embedding_dim: int
    transformer_hidden_dim: int
    latent_mlp_hidden_dims: List[int]
    num_coupling_layers: int
    num_transformer_layers: int
    transformer_config: TransformerConfig
    rff_position_encoder_config: Optional[RFFPositionEncoderConfig] = None
    position_layer_index_mod_2: int = 0
    conditional_flow_density: ConditionalFlowDensityConfig = field(default_factory=ConditionalFlowDensityConfig)


@dataclass
class ModelConfig:
    model_type: str
    transformer_nvp_config: Optional[TransformerNVPConfig] = None
    custom_transformer_nvp_config: Optional[CustomAttentionTransformerNVPConfig] = None


_NESTED = {
    "custom_transformer_nvp_config": CustomAttentionTransformerNVPConfig,
    "transformer_nvp_config": TransformerNVPConfig,
    "encoder_layer_config": CustomAttentionEncoderLayerConfig,
    "transformer_config": TransformerConfig,
    "rff_position_encoder_config": RFFPositionEncoderConfig,
    "conditional_flow_density": ConditionalFlowDensityConfig,
}


def _build(cls, d: Dict[str, Any]):
    known = {f.name for f in fields(cls)}
    unknown = set(d) - known
    if unknown:  # same behaviour as the reference's structured configs: unknown keys raise
        raise KeyError(f"unknown keys for {cls.__name__}: {sorted(unknown)}")
    kw = {}
    for k, v in d.items():
        if k in _NESTED and isinstance(v, dict):
            v = _build(_NESTED[k], v)
        kw[k] = v
    return cls(**kw)


def model_config_from_dict(d: Dict[str, Any]) -> ModelConfig:
    """Build a ModelConfig from the `model_config:` mapping of a reference YAML
    (configs/kernel_transformer_nvp.yaml:15-30, configs/transformer_nvp.yaml:14-25)."""
    return _build(ModelConfig, d)


def duck_get(cfg: Any, name: str, default: Any = None) -> Any:
    """Read a field from our dataclasses, the reference's dataclasses or an OmegaConf node."""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)
