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

# ONE object: a bound classmethod is a new object on every attribute access, and the sweep compares by identity
_ENERGY_FACTORY = AmberPotentialEnergyTorch.from_openmm
_SUPPORTED = ("custom_attention_transformer_nvp", "transformer_nvp", "euler_maruyama_gaussian")


def _try_import(name):
    try:
        return importlib.import_module(name)
    except Exception:  # optional reference modules may need packages that are not installed
        return sys.modules.get(name)


def _rebind(attr: str, originals, new, patched: dict) -> None:
    """Replace `attr` in every already-imported module that holds one of `originals` under that name.  The reference's
    callers bind these seams BY NAME at import time (`from timewarp.utils.openmm import OpenmmPotentialEnergyTorch`,
    evaluate.py:41, sample_trajectory.py:25, exploration.py:22; the package `utils/openmm/__init__.py:1` has copied the
    class before anyone can patch the bridge module), so patching the defining module alone is not enough."""
    originals = [o for o in originals if o is not None]
    for name, mod in list(sys.modules.items()):
        if mod is None or name.startswith("timewarp_amd"):
            continue
        d = getattr(mod, "__dict__", None)
        if not isinstance(d, dict) or attr not in d:
            continue
        if any(d[attr] is o for o in originals):
            setattr(mod, attr, new)
            patched[f"{name}.{attr}"] = True


def install(replace_energy: bool = True, replace_mh_loop: bool = True) -> dict:
    """Rebind the reference's seams; returns what was patched (for logging/tests).  Call it after `import timewarp`
    and before the script's `main()`; modules imported before OR after the call both end up with the replacements
    (before: swept through `sys.modules`; after: they import from the patched defining modules / packages)."""
    patched = {}
    ref_mc = importlib.import_module("timewarp.model_constructor")
    original = getattr(ref_mc, "_timewarp_amd_original", ref_mc.model_constructor)
    previous = ref_mc.model_constructor

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
    _try_import("timewarp.utils.training_utils")
    _rebind("model_constructor", [original, previous], model_constructor, patched)

    base = importlib.import_module("timewarp.modules.model_wrappers.density_model_base")
    base.ConditionalDensityModel.register(ConditionalFlowDensityModel)
    base.ConditionalDensityModelWithForce.register(EulerMaruyamaGaussian)
    patched["abc.register"] = True

    if replace_mh_loop:
        eu = _try_import("timewarp.utils.evaluation_utils")
        if eu is not None:
            for attr in ("sample_with_model", "sample_on_batches", "sample_on_single_conditional"):
                ours = getattr(_eu, attr)
                theirs = getattr(eu, "_timewarp_amd_original_" + attr, None) or getattr(eu, attr, None)
                if theirs is ours:
                    continue
                setattr(eu, "_timewarp_amd_original_" + attr, theirs)
                setattr(eu, attr, ours)
                patched[f"timewarp.utils.evaluation_utils.{attr}"] = True
                _rebind(attr, [theirs], ours, patched)
    if replace_energy:
        ours = _ENERGY_FACTORY
        bridge = _try_import("timewarp.utils.openmm.openmm_bridge")
        pkg = _try_import("timewarp.utils.openmm")
        theirs = None
        if bridge is not None:
            theirs = getattr(bridge, "_timewarp_amd_original_energy", None) or getattr(bridge, "OpenmmPotentialEnergyTorch", None)
            if theirs is not None and theirs is not ours:
                bridge._timewarp_amd_original_energy = theirs
            bridge.OpenmmPotentialEnergyTorch = ours
            patched["timewarp.utils.openmm.openmm_bridge.OpenmmPotentialEnergyTorch"] = True
        if pkg is not None and hasattr(pkg, "OpenmmPotentialEnergyTorch"):
            # the name evaluate.py:41, sample_trajectory.py:25 and exploration.py:22 import
            pkg.OpenmmPotentialEnergyTorch = ours
            patched["timewarp.utils.openmm.OpenmmPotentialEnergyTorch"] = True
        _rebind("OpenmmPotentialEnergyTorch", [theirs], ours, patched)
    return patched
