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
