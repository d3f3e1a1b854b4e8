"""timewarp_amd: MI355X-native implementation of Timewarp's conditional-flow sampling hot path.

Public surface (mirrors the reference's names for this path):
    model_constructor, ModelConfig & friends          -- the drop-in factory
    ConditionalFlowDensityModel                       -- log_likelihood / conditional_sample(_with_logp)
    sample_with_model, ChainStats                     -- batched Metropolis-Hastings loop
    AmberPotentialEnergyTorch                         -- energy callable with `.kbT`
The arithmetic lives in lib/libtimewarp_hip.so (include/timewarp_hip.h); there is no CPU fallback.
"""
from .model_configs import (  # noqa: F401
    ConditionalFlowDensityConfig,
    CustomAttentionEncoderLayerConfig,
    CustomAttentionTransformerNVPConfig,
    ModelConfig,
    RFFPositionEncoderConfig,
    TransformerConfig,
    TransformerNVPConfig,
    model_config_from_dict,
)
from .model_constructor import (  # noqa: F401
    custom_transformer_nvp_constructor,
    model_constructor,
    transformer_nvp_constructor,
)
from .modules.flow import ConditionalFlowDensityModel  # noqa: F401

__all__ = [
    "model_constructor",
    "custom_transformer_nvp_constructor",
    "transformer_nvp_constructor",
    "ConditionalFlowDensityModel",
    "ModelConfig",
    "CustomAttentionTransformerNVPConfig",
    "CustomAttentionEncoderLayerConfig",
    "TransformerNVPConfig",
    "TransformerConfig",
    "RFFPositionEncoderConfig",
    "ConditionalFlowDensityConfig",
    "model_config_from_dict",
]
