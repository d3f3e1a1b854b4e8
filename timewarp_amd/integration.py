"""Point an importable checkout of the reference (`timewarp` package) at this implementation.

    import timewarp                      # the reference, on sys.path as usual
    import timewarp_amd.integration as twi
    twi.install()                        # before sample.py / evaluate.py's main() runs

After `install()` the reference's scripts run unchanged:
  * `model_constructor` is rebound where callers bound it by name at import time
    (`timewarp.model_constructor`, `timewarp.utils.training_utils`; evaluate.py:43,
    sample_trajectory.py:26, exploration.py:24, profile.py:22) -- model types outside this build's
    scope fall through to the reference's own constructor;
  * our model classes are registered as virtual subclasses of the reference's ABCs, so
    `isinstance(model, ConditionalDensityModel)` and the `functools.singledispatch` in
    utils/sampling_utils.py:17-68 and utils/loss_utils.py:91-141 resolve them;
  * `sample_with_model`, `sample_on_batches`, `sample_on_single_conditional` and `OpenmmPotentialEnergyTorch` are replaced by the HIP-backed versions
    (the energy one reads its tables out of the `openmm.System` it is given).
"""
from __future__ import annotations

import importlib
import sys

from .model_constructor import model_constructor as _tw_model_constructor
from .energy import AmberPotentialEnergyTorch
from .modules.baselines import EulerMaruyamaGaussian
from .modules.flow import ConditionalFlowDensityModel
from .utils import evaluation_utils as _eu

_SUPPORTED = ("custom_attention_transformer_nvp", "transformer_nvp", "euler_maruyama_gaussian")


def _try_import(name):
    try:
        return importlib.import_module(name)
    except Exception:  # optional reference modules may need packages that are not installed
        return sys.modules.get(name)


def install(replace_energy: bool = True, replace_mh_loop: bool = True) -> dict:
    """Rebind the reference's seams; returns what was patched (for logging/tests)."""
    patched = {}
    ref_mc = importlib.import_module("timewarp.model_constructor")
    original = getattr(ref_mc, "_timewarp_amd_original", ref_mc.model_constructor)

    def model_constructor(config):
        if getattr(config, "model_type", None) in _SUPPORTED:
            if config.model_type == "custom_attention_transformer_nvp":
                enc = config.custom_transformer_nvp_config.encoder_layer_config
                if getattr(enc, "attention_type", None) not in ("kernel", "learnable_kernel", "chebyshev_kernel"):
                    return original(config)  # local attention stays on the reference
            return _tw_model_constructor(config)
        return original(config)

    ref_mc._timewarp_amd_original = original
    ref_mc.model_constructor = model_constructor
    patched["timewarp.model_constructor.model_constructor"] = True
    tu = _try_import("timewarp.utils.training_utils")
    if tu is not None and hasattr(tu, "model_constructor"):
        tu.model_constructor = model_constructor
        patched["timewarp.utils.training_utils.model_constructor"] = True

    base = importlib.import_module("timewarp.modules.model_wrappers.density_model_base")
    base.ConditionalDensityModel.register(ConditionalFlowDensityModel)
    base.ConditionalDensityModelWithForce.register(EulerMaruyamaGaussian)
    patched["abc.register"] = True

    if replace_mh_loop:
        eu = _try_import("timewarp.utils.evaluation_utils")
        if eu is not None:
            eu.sample_with_model = _eu.sample_with_model
            eu.sample_on_batches = _eu.sample_on_batches
            eu.sample_on_single_conditional = _eu.sample_on_single_conditional
            patched["timewarp.utils.evaluation_utils.sample_with_model"] = True
            patched["timewarp.utils.evaluation_utils.sample_on_batches"] = True
            patched["timewarp.utils.evaluation_utils.sample_on_single_conditional"] = True
    if replace_energy:
        for name in ("timewarp.utils.openmm.openmm_bridge", "timewarp.utils.evaluation_utils"):
            mod = _try_import(name)
            if mod is not None and hasattr(mod, "OpenmmPotentialEnergyTorch"):
                mod.OpenmmPotentialEnergyTorch = AmberPotentialEnergyTorch.from_openmm
                patched[name + ".OpenmmPotentialEnergyTorch"] = True
    return patched
