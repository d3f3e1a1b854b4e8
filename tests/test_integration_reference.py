"""The drop-in seam against the real reference package (only where /root/reference exists: the
build container; skipped on the GPU box)."""
import os
import sys
import types

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def _import_reference():
    link_dir = "/tmp/tw_oracle_ref"
    os.makedirs(link_dir, exist_ok=True)
    link = os.path.join(link_dir, "timewarp")
    if not os.path.islink(link):
        os.symlink(REF, link)
    if link_dir not in sys.path:
        sys.path.insert(0, link_dir)
    if REF not in sys.path:
        sys.path.append(REF)

    class _Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            m = _Stub(f"{self.__name__}.{k}")
            setattr(self, k, m)
            return m

        def __call__(self, *a, **k):
            return None

    for n in ("mdtraj", "pymol2", "torch.utils.tensorboard", "tensorboard"):
        sys.modules.setdefault(n, _Stub(n))


def test_install_rebinds_factory_and_dispatch():
    _import_reference()
    import timewarp.model_constructor as ref_mc
    from timewarp.model_configs import CustomAttentionTransformerNVPConfig, ModelConfig
    from timewarp.modules.layers.custom_attention_encoder import CustomAttentionEncoderLayerConfig
    from timewarp.modules.model_wrappers.density_model_base import ConditionalDensityModel
    from timewarp.utils import sampling_utils as ref_su

    import timewarp_amd.integration as twi
    from timewarp_amd.modules.flow import ConditionalFlowDensityModel

    patched = twi.install(replace_energy=False, replace_mh_loop=False)
    assert patched["timewarp.model_constructor.model_constructor"]
    enc = CustomAttentionEncoderLayerConfig(d_model=128, dim_feedforward=2048, dropout=0.0, num_heads=6,
                                            attention_type="kernel", lengthscales=[0.1, 0.2, 0.5, 0.7, 1.0, 1.2],
                                            normalise_kernel_values=True)
    cfg = ModelConfig(model_type="custom_attention_transformer_nvp",
                      custom_transformer_nvp_config=CustomAttentionTransformerNVPConfig(
                          atom_embedding_dim=32, latent_mlp_hidden_dims=[256], num_coupling_layers=8,
                          num_transformer_layers=3, encoder_layer_config=enc))
    model = ref_mc.model_constructor(cfg)  # the reference's factory, reference's config dataclasses
    assert isinstance(model, ConditionalFlowDensityModel)
    assert isinstance(model, ConditionalDensityModel)  # the reference's ABC
    # the reference's singledispatch picks the ConditionalDensityModel overload for our class
    assert ref_su.get_sample.dispatch(type(model)) is ref_su.get_sample.dispatch(ConditionalDensityModel)
    # a reference checkpoint state_dict loads key-for-key
    ref_model = ref_mc._timewarp_amd_original(cfg)
    missing = model.load_state_dict(ref_model.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    # out-of-scope model types fall through to the reference's own constructor
    other = ref_mc.model_constructor(ModelConfig(model_type="initial_state_gaussian"))
    assert type(other).__module__.startswith("timewarp.")


# ---------------------------------------------------------------------------------------------------------------------
# install() with its DEFAULTS against the real package, and the names the reference's scripts actually bind
# (r03 review: `from timewarp.utils.openmm import OpenmmPotentialEnergyTorch` - evaluate.py:41, sample_trajectory.py:25,
# exploration.py:22 - still gave the reference's class after install()).  Each scenario runs in a fresh interpreter so
# the order of imports is the one under test.  Third-party packages the scripts import at module scope and this image
# lacks (openmm, bgflow, mdtraj, lmdb, deepspeed, omegaconf, git, ...) are attribute-permissive placeholders; none of
# their functionality is exercised - only the reference's own import graph is.
_SEAM_PRELUDE = r'''
import os, sys, types, importlib
sys.path.insert(0, {root!r})
REF = "/root/reference"
link_dir = "/tmp/tw_oracle_ref"; os.makedirs(link_dir, exist_ok=True)
link = os.path.join(link_dir, "timewarp")
if not os.path.islink(link):
    os.symlink(REF, link)
sys.path.insert(0, link_dir); sys.path.append(REF)
class _Stub(types.ModuleType):
    __path__ = []
    def __getattr__(self, k):
        if k.startswith("__"): raise AttributeError(k)
        if k[:1].isupper():
            c = type(k, (), {{"__init__": lambda self, *a, **kw: None}}); setattr(self, k, c); return c
        m = _Stub(self.__name__ + "." + k); setattr(self, k, m); return m
    def __call__(self, *a, **k): return None
for n in ("mdtraj", "pymol2", "torch.utils.tensorboard", "tensorboard", "git", "git.types", "openmm", "openmm.app",
          "openmm.unit", "bgflow", "bgflow.distribution", "bgflow.distribution.energy", "bgflow.distribution.energy.openmm",
          "bgflow.distribution.energy.base", "bgflow.utils", "bgflow.utils.types", "matplotlib", "matplotlib.pyplot",
          "simtk", "simtk.unit", "simtk.openmm", "simtk.openmm.app", "lmdb", "deepspeed", "omegaconf"):
    sys.modules[n] = _Stub(n)
class _MM:  # `multimethod` is used as a decorator with .register (losses.py:173)
    def __init__(self, fn): self.fn = fn
    def register(self, *a, **k):
        return self if (a and callable(a[0]) and not isinstance(a[0], type)) else (lambda f: self)
    def __call__(self, *a, **k): return self.fn(*a, **k)
_mm = types.ModuleType("multimethod"); _mm.multimethod = _MM; sys.modules["multimethod"] = _mm
import timewarp_amd.integration as twi
from timewarp_amd.energy import AmberPotentialEnergyTorch
from timewarp_amd.utils import evaluation_utils as our_eu
SCRIPTS = ("timewarp.sample_trajectory", "timewarp.evaluate", "timewarp.exploration")
def check():
    import timewarp.model_constructor as ref_mc
    from timewarp.utils.openmm import OpenmmPotentialEnergyTorch as E
    assert E is twi._ENERGY_FACTORY, E
    import timewarp.utils.openmm.openmm_bridge as bridge
    assert bridge.OpenmmPotentialEnergyTorch is twi._ENERGY_FACTORY
    import timewarp.utils.evaluation_utils as ref_eu
    assert ref_eu.sample_with_model is our_eu.sample_with_model
    assert ref_eu.sample_on_batches is our_eu.sample_on_batches
    assert ref_eu.sample_on_single_conditional is our_eu.sample_on_single_conditional
    assert ref_eu.OpenmmPotentialEnergyTorch is twi._ENERGY_FACTORY
    import timewarp.utils.training_utils as tu
    assert tu.model_constructor is ref_mc.model_constructor and hasattr(ref_mc, "_timewarp_amd_original")
    assert ref_mc.model_constructor is not ref_mc._timewarp_amd_original
    for name in SCRIPTS:  # what the script's own module namespace resolves, i.e. what main() will call
        m = importlib.import_module(name)
        assert m.OpenmmPotentialEnergyTorch is twi._ENERGY_FACTORY, (name, m.OpenmmPotentialEnergyTorch)
        assert m.model_constructor is ref_mc.model_constructor, name
        if hasattr(m, "sample_with_model"):
            assert m.sample_with_model is our_eu.sample_with_model, name
        for attr in ("sample_on_batches", "sample_on_single_conditional"):
            if hasattr(m, attr):
                assert getattr(m, attr) is getattr(our_eu, attr), (name, attr)
    # the factory takes the scripts' call (evaluate.py:296-301) and returns the HIP-backed energy
    import inspect
    sig = inspect.signature(twi._ENERGY_FACTORY)
    sig.bind("system", "integrator", platform_name="CUDA", platform_properties=dict(CudaDeviceIndex="0"))
    assert twi._ENERGY_FACTORY.__self__ is AmberPotentialEnergyTorch
'''


def _run_seam_scenario(body: str):
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _SEAM_PRELUDE.format(root=root) + body
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_install_defaults_then_scripts_import():
    """launcher order of INTEGRATION.md: import timewarp, install(), then the script is imported / run."""
    out = _run_seam_scenario(r'''
import timewarp
import timewarp.utils.openmm            # the package __init__ copies the class (utils/openmm/__init__.py:1)
patched = twi.install()
assert patched["timewarp.utils.openmm.OpenmmPotentialEnergyTorch"]
assert patched["timewarp.utils.evaluation_utils.sample_with_model"]
check()
print("OK", len(patched))
''')
    assert "OK" in out


def test_install_defaults_after_scripts_were_imported():
    """The scripts' modules already hold the reference's objects by name: install() sweeps sys.modules."""
    out = _run_seam_scenario(r'''
mods = [importlib.import_module(n) for n in SCRIPTS]
import timewarp.utils.openmm.openmm_bridge as bridge
ref_energy = bridge.OpenmmPotentialEnergyTorch
assert all(m.OpenmmPotentialEnergyTorch is ref_energy for m in mods)
patched = twi.install()
for n in SCRIPTS:
    assert patched[n + ".OpenmmPotentialEnergyTorch"], sorted(patched)
    assert patched[n + ".model_constructor"], sorted(patched)
assert patched["timewarp.sample_trajectory.sample_with_model"] and patched["timewarp.evaluate.sample_on_batches"]
check()
again = twi.install()                   # idempotent: a second call finds nothing of the reference's left
check()
assert bridge._timewarp_amd_original_energy is ref_energy
print("OK", len(patched))
''')
    assert "OK" in out
