"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol,
the weight layout, the model factory / state_dict contract, configs, force-field tables."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from oracle import mh_oracle as mo
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from timewarp_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "timewarp_hip.h")).read()
    declared = set(re.findall(r"\b(tw_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"tw_status"}
    assert len(declared) >= 18
    lib = C.CDLL(_lib.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/timewarp_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().tw_abi_version() == 1


def test_raw_layout_matches_library_and_oracle_template():
    import timewarp_amd as tw
    from timewarp_amd import _lib, synthetic
    from timewarp_amd.weights import pack_raw, raw_entries, raw_numel

    lib = _lib.load()
    m = tw.model_constructor(synthetic.kernel_transformer_nvp_config())
    d = m.dims.to_desc()
    assert lib.tw_flow_raw_floats(C.byref(d)) == raw_numel(m.dims)
    assert lib.tw_flow_packed_floats(C.byref(d)) > 0
    assert lib.tw_flow_workspace_bytes(C.byref(d), 1000, 22) > 0
    sd = m.state_dict()
    t = fo.make_template(H.FULL_KERNEL_SPEC)
    assert set(sd) == set(t) and all(sd[k].shape == t[k].shape for k in t)
    assert sum(p.numel() for p in m.parameters()) == 35971282  # 35.97 M
    # pack order: embedding first, then lengthscales, the two prior log-scales, chain 0 scale in_mlp
    sd = synthetic.synth_state_dict(sd, 0)
    raw = pack_raw(sd, m.dims)
    assert torch.equal(raw[:160], sd["flow.atom_embedder.weight"].reshape(-1))
    assert torch.equal(raw[160:166], torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2]))
    assert raw[166] == sd["coords_prior_log_scale"] and raw[167] == sd["velocs_prior_log_scale"]
    assert torch.equal(raw[168:168 + 256 * 41], sd["flow.chain.0.scale_transformer.in_mlp._layers.0.weight"].reshape(-1))
    assert raw_entries(m.dims)[-1][0] == "flow.chain.7.shift_transformer.out_mlp._layers.2.bias"
    with pytest.raises(ValueError):
        bad = dict(sd)
        bad["flow.atom_embedder.weight"] = torch.zeros(5, 31)
        pack_raw(bad, m.dims)


def test_dense_constructor_keys_and_module_prefix():
    import timewarp_amd as tw

    cfg = tw.ModelConfig("transformer_nvp", transformer_nvp_config=tw.TransformerNVPConfig(
        32, 128, [256], 8, 3, tw.TransformerConfig()))
    m = tw.model_constructor(cfg)
    t = fo.make_template(H.FULL_DENSE_SPEC)
    sd = m.state_dict()
    assert set(sd) == set(t)
    assert sum(p.numel() for p in m.parameters()) == 29704402  # SURVEY: 29.70 M
    # DeepSpeed / LossWrapper checkpoints prefix keys with `module.` (losses.py:247-258)
    m.load_state_dict({"module." + k: v for k, v in sd.items()})
    with pytest.raises(NotImplementedError):
        tw.model_constructor(tw.ModelConfig("equivariant_nvp"))


def test_holder_modules_refuse_to_compute():
    import timewarp_amd as tw
    from timewarp_amd import synthetic

    m = tw.model_constructor(synthetic.kernel_transformer_nvp_config())
    with pytest.raises(RuntimeError, match="only stores weights"):
        m.flow.chain[0].scale_transformer(torch.zeros(1, 22, 41))


def test_config_from_reference_yaml_mapping():
    import yaml
    import timewarp_amd as tw

    text = """
model_config:
  model_type: custom_attention_transformer_nvp
  custom_transformer_nvp_config:
    atom_embedding_dim: 32
    latent_mlp_hidden_dims: [256]
    num_coupling_layers: 8
    num_transformer_layers: 3
    encoder_layer_config:
      d_model: 128
      dim_feedforward: 2048
      num_heads: 6
      dropout: 0
      attention_type: kernel
      lengthscales: [0.1, 0.2, 0.5, 0.7, 1., 1.2]
      normalise_kernel_values: true
"""
    cfg = tw.model_config_from_dict(yaml.safe_load(text)["model_config"])
    m = tw.model_constructor(cfg)
    assert m.dims.n_heads == 6 and m.dims.d_ff == 2048
    bad = yaml.safe_load(text)["model_config"]
    bad["custom_transformer_nvp_config"]["no_such_key"] = 1
    with pytest.raises(KeyError):
        tw.model_config_from_dict(bad)


def test_synthetic_recipe_identical_to_oracle_recipe():
    from timewarp_amd import synthetic

    t = fo.make_template(fo.FlowSpec(num_coupling_layers=2, num_transformer_layers=1))
    a = fo.synth_state_dict(t, 3, calibrated=True, coords_log_scale=-7.0, velocs_log_scale=0.0)
    b = synthetic.synth_state_dict(t, 3, calibrated=True, coords_log_scale=-7.0, velocs_log_scale=0.0)
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    z = np.load(os.path.join(H.GOLDEN, "ad_topology.npz"))
    assert np.array_equal(types.numpy(), z["atom_types"]) and np.allclose(coords.numpy(), z["coords_nm"], atol=1e-7)


def test_forcefield_tables_alanine_dipeptide():
    from timewarp_amd.forcefield import alanine_dipeptide_amber99sb

    t = alanine_dipeptide_amber99sb()
    assert t.n_atoms == 22 and len(t.bond_idx) == 21 and len(t.angle_idx) == 36
    assert abs(t.atom_par[:, 0].sum()) < 1e-9  # neutral
    n12_13 = 21 + 36
    assert len(t.exc_idx) > n12_13 and (t.exc_par[:n12_13, 0] == 0).all()
    pairs = {tuple(p) for p in t.exc_idx.tolist()}
    assert len(pairs) == len(t.exc_idx)  # no duplicates


def test_proposal_step_schedule_matches_oracle():
    from timewarp_amd.utils.evaluation_utils import ChainStats, compute_num_proposal_steps

    for p in (0.0, 1e-3, 0.01, 0.1, 0.5, 0.9, 1.0):
        for mx in (1, 10, 100, 1000):
            assert compute_num_proposal_steps(p, max_num_proposal_steps=mx) == mo.compute_num_proposal_steps(p, max_steps=mx)
    s = ChainStats(*[np.arange(10) for _ in range(9)])
    assert len(s) == 10 and len(s.thin(3)) == 4 and len(s[2:5]) == 3


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "timewarp_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
