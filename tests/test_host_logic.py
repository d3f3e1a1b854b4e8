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
    assert _lib.load().tw_abi_version() == _lib.ABI_VERSION == 8


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
    ls = torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2])
    assert torch.equal(raw[160:172], torch.cat([ls, ls]))  # forward-pass row, reverse-pass row
    assert raw[172] == sd["coords_prior_log_scale"] and raw[173] == sd["velocs_prior_log_scale"]
    assert torch.equal(raw[174:174 + 256 * 41], sd["flow.chain.0.scale_transformer.in_mlp._layers.0.weight"].reshape(-1))
    assert raw_entries(m.dims)[-1][0] == "flow.chain.7.shift_transformer.out_mlp._layers.2.bias"
    with pytest.raises(ValueError):
        bad = dict(sd)
        bad["flow.atom_embedder.weight"] = torch.zeros(5, 31)
        pack_raw(bad, m.dims)
    # learnable lengthscales: row 0 from chain 0, row 1 from the last coupling layer (first layer evaluated
    # by a forward / reverse pass), and the extra parameter is part of the state_dict
    cfg = synthetic.kernel_transformer_nvp_config()
    cfg.custom_transformer_nvp_config.encoder_layer_config.attention_type = "learnable_kernel"
    ml = tw.model_constructor(cfg)
    key = "flow.chain.{}.scale_transformer.encoder_layers.0.self_attn.attention.log_lengthscales"
    sdl = ml.state_dict()
    assert key.format(3) in sdl and lib.tw_flow_raw_floats(C.byref(ml.dims.to_desc())) == raw_numel(ml.dims)
    sdl[key.format(0)] = torch.log(ls * 2.0)
    sdl[key.format(7)] = torch.log(ls * 3.0)
    rawl = pack_raw(sdl, ml.dims)
    assert torch.allclose(rawl[160:166], ls * 2.0) and torch.allclose(rawl[166:172], ls * 3.0)


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


def test_execution_path_from_environment(monkeypatch):
    import timewarp_amd as tw
    from timewarp_amd import _lib, synthetic
    from timewarp_amd.modules import flow

    cfg = synthetic.kernel_transformer_nvp_config()
    monkeypatch.delenv("TW_EXECUTION_PATH", raising=False)
    assert tw.model_constructor(cfg).execution_path == flow.PREFER_SPLIT_FP16  # the measured path is the default
    for name, want in [("auto", _lib.TW_PATH_AUTO), ("f32", _lib.TW_PATH_FUSED), ("simple", _lib.TW_PATH_SIMPLE),
                       ("H3", flow.PREFER_SPLIT_FP16)]:
        monkeypatch.setenv("TW_EXECUTION_PATH", name)
        assert tw.model_constructor(cfg).execution_path == want
    m = tw.model_constructor(cfg)
    assert m._path_for(22) == _lib.TW_PATH_FUSED_H3 and m._path_for(60) == _lib.TW_PATH_FUSED_H3
    monkeypatch.setenv("TW_EXECUTION_PATH", "fp8")
    with pytest.raises(ValueError, match="TW_EXECUTION_PATH"):
        tw.model_constructor(cfg)
    # every molecule a fused layout takes runs on the split-fp16 kernel (tw_flow_path_supported); above them (r06) the per-op path
    # with split-fp16 linears (TW_PATH_SIMPLE_H3 = 5); a shape only the exact-f32 fused kernel serves would get AUTO
    monkeypatch.setenv("TW_EXECUTION_PATH", "h3")
    assert [m._path_for(v) for v in (1, 7, 12, 17, 21, 30, 40, 48, 49, 64, 65, 160, 192, 193, 200, 691)] == [3] * 13 + [5] * 3
    dense = tw.model_constructor(synthetic.transformer_nvp_config())
    assert [dense._path_for(v) for v in (22, 48, 49, 64, 65, 200)] == [3, 3, 3, 3, 5, 5]
    # ... unless the score-fragment producer's LDS tile would not fit the CU: 48 atoms x 18 heads
    many = synthetic.kernel_transformer_nvp_config()
    many.custom_transformer_nvp_config.encoder_layer_config.lengthscales = [0.1 * (i + 1) for i in range(18)]
    many.custom_transformer_nvp_config.encoder_layer_config.num_heads = 18
    mm = tw.model_constructor(many)
    # (the wide layout's producer works one head at a time, so 25 .. 160 atoms stay on the split-fp16 kernel; 24 atoms and
    # fewer need the per-wave producer, whose tile would be 2 x 18 x 24 x 24 floats + ... and fits)
    assert mm._path_for(22) == 3 and mm._path_for(48) == 3 and mm._path_for(60) == 3


def test_path_supported_sweep():
    """tw_flow_path_supported over 1 .. 200 atoms: the split-fp16 kernel takes 1 .. 48 (48-token waves), the wide
    layout 25 .. 192 (161 .. 192: one molecule per workgroup, six-group windows; 81 .. 95 with a slot stride of 96: back to back, wave 1 would span two molecules over eleven key tiles =
    six key groups, the statement has five); the single-MFMA fast path the same set; the f32 kernel 1 .. 64."""
    import ctypes as C
    from timewarp_amd import _lib, synthetic
    import timewarp_amd as tw

    lib = _lib.load()
    desc = tw.model_constructor(synthetic.kernel_transformer_nvp_config()).dims.to_desc()
    sup = lambda path: [v for v in range(1, 201) if lib.tw_flow_path_supported(C.byref(desc), v, path) == 1]
    assert sup(3) == list(range(1, 193))
    assert sup(4) == sup(3)   # the single-MFMA fast path: wherever the split-fp16 kernel runs kernel attention
    assert sup(1) == list(range(1, 65))
    assert sup(2) == list(range(1, 201)) and sup(0) == list(range(1, 201))


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


def test_alanine_dipeptide_charges_against_the_pinned_conventions():
    """The ACE / ALA / NME charge rows are the one part of the alanine-dipeptide table the NNQQ known-answer file cannot
    exercise (tests/test_energy_kat.py pins every type, bond, angle, torsion, LJ and GB class they use).  What can
    still be checked: ff94 gives every non-terminal residue the same backbone N / H / C / O charges, so the ALA row must
    carry exactly the values the pinned ASN and GLN rows carry; ACE's C / O and NME's N / H are those same backbone
    charges (C of ACE -0.0001 as in the ff94 library, so that the cap is neutral); every residue is neutral; hydrogens on one carbon are equal."""
    from timewarp_amd.forcefield import RESIDUES

    q = {r: RESIDUES[r]["charges"] for r in ("ACE", "ALA", "NME", "ASN", "GLN")}
    for atom in ("N", "H", "C", "O"):
        assert q["ALA"][atom] == q["ASN"][atom] == q["GLN"][atom], atom
    assert q["ACE"]["O"] == q["ALA"]["O"] and abs(q["ACE"]["C"] - q["ALA"]["C"]) <= 1.0001e-4
    assert q["NME"]["N"] == q["ALA"]["N"] and q["NME"]["H"] == q["ALA"]["H"]
    for r in q:
        assert abs(sum(q[r].values())) < 1e-9, r
    assert q["ACE"]["HH31"] == q["ACE"]["HH32"] == q["ACE"]["HH33"]
    assert q["NME"]["HH31"] == q["NME"]["HH32"] == q["NME"]["HH33"]
    assert q["ALA"]["HB1"] == q["ALA"]["HB2"] == q["ALA"]["HB3"]


def test_every_residue_template_builds_in_any_neighbourhood():
    """The 18-residue amber99sb-ildn library (pinned on the protein file, tests/test_energy_kat.py): every template is neutral
    or carries its formal charge; every capped single residue ACE-X-NME and every pair X-PRO / PRO-X resolves all of its bond,
    angle, torsion, LJ and GB-radius types (the protein only holds some neighbour pairs); ILDN series attach where they should."""
    from timewarp_amd import forcefield as ff

    formal = {"ASP": -1, "GLU": -1, "LYS": 1, "ARG": 1, "NASN": 1, "NMET": 1, "CGLN": -1, "CGLY": -1}
    for name, tpl in ff.RESIDUES.items():
        assert abs(sum(tpl["charges"].values()) - formal.get(name, 0)) < 1e-9, name
    mids = [r for r in ff.RESIDUES if r not in ("ACE", "NME") and not (len(r) == 4 and r[0] in "NC")]
    assert len(mids) == 18 and "HIS" not in mids and "CYS" not in mids

    def chain(seq):
        names, res, rid = [], [], []
        for i, r in enumerate(seq):
            for a in ff.RESIDUES[r]["names"]:
                names.append(a); res.append(r); rid.append(i + 1)
        return ff.amber99sbildn_obc_tables(names, res, rid)

    for r in mids:
        t = chain(["ACE", r, "NME"])
        assert abs(t.atom_par[:, 0].sum() - formal.get(r, 0)) < 1e-9
        assert len(t.bond_idx) >= t.n_atoms - 1 and (t.atom_par[:, 3] > 0.1).all()
        chain(["ACE", r, "PRO", "NME"]); chain(["ACE", "PRO", r, "NME"])
    per = lambda t: sorted(t.torsion_par[:, 0].astype(int).tolist())
    assert per(chain(["ACE", "ASP", "NME"])).count(6) == 3 and per(chain(["ACE", "ASN", "NME"])).count(6) == 2
    assert per(chain(["ACE", "ALA", "NME"])).count(6) == 0


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


def test_transform_batch_is_rigid():
    """transform_batch: coordinates translate+rotate, velocities rotate; the flow is equivariant only up
    to the model, so just check the geometry (distances preserved, velocity norms preserved)."""
    from timewarp_amd.dataloader import single_state_batch, transform_batch

    g = torch.Generator().manual_seed(3)
    x, v = torch.randn(1, 9, 3, generator=g), torch.randn(1, 9, 3, generator=g)
    b = single_state_batch("m", torch.zeros(9, dtype=torch.int64), x, v)
    t = transform_batch(b)
    assert torch.allclose(torch.cdist(t.atom_coords, t.atom_coords), torch.cdist(x, x), atol=1e-5)
    assert torch.allclose(t.atom_velocs.norm(dim=-1), v.norm(dim=-1), atol=1e-5)
    assert not torch.allclose(t.atom_coords, x)


def test_find_chirality_centers_alanine_dipeptide_like():
    """A carbon with four bonds, three of them to heavy atoms, is a centre; a methyl carbon is not
    (reference utils/chirality.py:14-38)."""
    from timewarp_amd.utils.chirality import compute_chirality_sign, find_chirality_centers

    # atoms: 0 C(alpha) 1 N 2 C' 3 C(beta) 4 H | methyl 3: 5 H 6 H 7 H | 8 H on N, 9 O on C'
    types = torch.tensor([[0, 2, 0, 0, 1, 1, 1, 1, 1, 3]])
    adj = torch.tensor([[0, 1], [0, 2], [0, 3], [0, 4], [3, 5], [3, 6], [3, 7], [1, 8], [2, 9]])
    cen = find_chirality_centers(adj, types)
    assert cen.tolist() == [[0, 1, 2, 3]]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 10, 3, generator=g)
    s = compute_chirality_sign(x, cen)
    mirrored = x * torch.tensor([1.0, 1.0, -1.0])
    assert s.shape == (2, 1) and torch.equal(compute_chirality_sign(mirrored, cen), -s)
    assert torch.equal(mo.compute_chirality_sign(x, cen), s)


def test_sample_trajectory_segments_and_resume(tmp_path):
    """Segment files, thinning by 10, time field, and resume from the last saved row
    (reference sample_trajectory.py:234-279), with the sampler injected so no GPU is needed."""
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.sample_trajectory import resume_point, sample_trajectory, segment_path

    V = 4
    calls = []

    def fake_sampler(batch, model, device, energy, masses, num_samples, accept, **kw):
        start = batch.atom_coords.numpy().astype(np.float32)
        calls.append(start.copy())
        steps = np.arange(num_samples + 1, dtype=np.float32)[:, None, None]
        coords = start + steps * 0.001  # state j = start + j*0.001
        return coords, coords, 0, None

    out = str(tmp_path / "chain")
    b = single_state_batch("pep", torch.zeros(V, dtype=torch.int64), torch.zeros(1, V, 3))
    n = sample_trajectory(b, None, "cpu", None, None, out, "pep", num_samples=40, saving_interval=20,
                          sampler=fake_sampler, verbose=False)
    assert n == 2 and sorted(os.listdir(out)) == ["pep_trajectory_model_0.npz", "pep_trajectory_model_1.npz"]
    z0 = np.load(segment_path(out, "pep", 0))
    assert z0["positions"].shape == (3, V, 3) and float(z0["time"]) >= 0.0       # rows 0, 10, 20 of 21 states
    assert np.allclose(z0["positions"][:, 0, 0], [0.0, 0.010, 0.020], atol=1e-6)
    assert np.allclose(calls[1][0, 0, 0], 0.020, atol=1e-6)                     # segment 1 starts at the last state
    # resume: nothing left to do for the same length, two more segments for a longer chain
    assert sample_trajectory(b, None, "cpu", None, None, out, "pep", 40, 20, sampler=fake_sampler, verbose=False) == 0
    done, last = resume_point(out, "pep")
    assert done == 2 and np.allclose(last[0, 0, 0], 0.040, atol=1e-6)
    b2 = single_state_batch("pep", torch.zeros(V, dtype=torch.int64), torch.zeros(1, V, 3))
    assert sample_trajectory(b2, None, "cpu", None, None, out, "pep", 80, 20, sampler=fake_sampler, verbose=False) == 2
    assert np.allclose(calls[-2][0, 0, 0], 0.040, atol=1e-6)                    # resumed from the saved row
    assert len(os.listdir(out)) == 4


def test_recorded_draws_replay_in_order():
    """The split-fp16 range guard redoes the iterations since the last read-back on the f32 kernels with the SAME random
    numbers: RecordingNoise keeps every draw in order (latents written into caller buffers are kept as copies - the
    buffers become the proposals), `mark()` starts a new window, ReplayDraws hands the window back draw by draw, also
    through the other latents entry point."""
    from timewarp_amd.utils.evaluation_utils import DeviceNoise, RecordingNoise, ReplayDraws

    S, V = 5, 4
    rec = RecordingNoise(DeviceNoise("cpu", seed=3))
    rec.uniform(S)                                   # an iteration before the read-back: forgotten by mark()
    rec.mark()
    zc, zv = torch.zeros(S + 1, V, 3), torch.zeros(S + 1, V, 3)
    rec.latents_into(zc, zv, S, 0.5, 2.0)
    kept_c, kept_v = zc[:S].clone(), zv[:S].clone()
    zc[:S] += 7.0                                    # the kernel overwrites the buffers with the proposals
    u = rec.uniform(S)
    r = rec.randn_like(torch.zeros(1, V, 3))
    a, b = rec.latents(S, 1, V, 0.5, 2.0)
    q = rec.rotation()
    assert len(rec.log) == 5

    rep = RecordingNoise(ReplayDraws(rec.log))       # the redo records again (a second overflow would replay the replay)
    zc2, zv2 = torch.zeros(S + 1, V, 3), torch.zeros(S + 1, V, 3)
    rep.latents_into(zc2, zv2, S, 0.5, 2.0, scale_c=0.5, scale_v=2.0)
    assert torch.equal(zc2[:S], kept_c) and torch.equal(zv2[:S], kept_v) and float(zc2[S].abs().max()) == 0.0
    assert torch.equal(rep.uniform(S), u)
    assert torch.equal(rep.randn_like(torch.zeros(1, V, 3)), r)
    a2, b2 = rep.latents(S, 1, V, 0.5, 2.0)
    assert torch.equal(a2, a) and torch.equal(b2, b) and a2.shape == (S, 1, V, 3)
    assert torch.equal(rep.rotation(), q)
    assert len(rep.log) == 5 and not rep.inner.log
    # the draws themselves are the seeded stream of the wrapped source
    ref = DeviceNoise("cpu", seed=3)
    ref.uniform(S)
    zr, zs = torch.zeros(S + 1, V, 3), torch.zeros(S + 1, V, 3)
    ref.latents_into(zr, zs, S, 0.5, 2.0)
    assert torch.equal(zr[:S], kept_c) and torch.equal(ref.uniform(S), u)


def test_generated_asm_includes_are_current(tmp_path):
    """The committed tw_h3_*_asm.inc / *_clobbers.inc files are exactly what the generators in tools/ emit (a build
    needs hipcc only; this keeps the committed text from going stale)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("H3_")}
    for args in (["tools/gen_h3_ffn_asm.py", "--shape=ffn"], ["tools/gen_h3_ffn_asm.py", "--shape=in"],
                 ["tools/gen_h3_ffn_asm.py", "--shape=out"], ["tools/gen_h3_attn_asm.py"],
                 ["tools/gen_h3_attn_asm.py", "--mode=windowed"], ["tools/gen_h3_attn_wide_asm.py"],
                 ["tools/gen_h3_dense_attn_asm.py"], ["tools/gen_h3_enc_asm.py"], ["tools/gen_h3_enc_asm.py", "--mode=windowed"],
                 # the single-MFMA variant (TW_PATH_FUSED_H1): tw_h1_*
                 ["tools/gen_h3_ffn_asm.py", "--shape=in", "--h1"], ["tools/gen_h3_ffn_asm.py", "--shape=out", "--h1"],
                 ["tools/gen_h3_enc_asm.py", "--h1"], ["tools/gen_h3_enc_asm.py", "--mode=windowed", "--h1"],
                 ["tools/gen_h3_ffn_asm.py", "--shape=ffn", "--h1"], ["tools/gen_h3_attn_wide_asm.py", "--h1"],
                 # ... on the six-slot ring (one barrier per pair of FFN stages): tw_h1r_*
                 ["tools/gen_h3_ffn_asm.py", "--shape=in", "--h1", "--ring6"], ["tools/gen_h3_ffn_asm.py", "--shape=out", "--h1", "--ring6"],
                 ["tools/gen_h3_enc_asm.py", "--h1", "--ring6"], ["tools/gen_h3_enc_asm.py", "--mode=windowed", "--h1", "--ring6"],
                 # wide layout, 65-96 atoms at the 96-slot stride: three-group windows, tw_h?_attns3_*
                 ["tools/gen_h3_attn_wide_asm.py", "--ng=3"], ["tools/gen_h3_attn_wide_asm.py", "--ng=3", "--h1"],
                 # ... 161-192 atoms, one molecule per workgroup: six-group windows, tw_h?_attns6_*
                 ["tools/gen_h3_attn_wide_asm.py", "--ng=6"], ["tools/gen_h3_attn_wide_asm.py", "--ng=6", "--h1"],
                 # 64-token waves (49-64 atoms): tw_h3n4_*
                 ["tools/gen_h3_ffn_asm.py", "--shape=ffn", "--nt=4"], ["tools/gen_h3_attn_asm.py", "--nt=4"],
                 ["tools/gen_h3_ffn_asm.py", "--shape=in", "--nt=4"], ["tools/gen_h3_ffn_asm.py", "--shape=out", "--nt=4"],
                 ["tools/gen_h3_ffn_asm.py", "--shape=ffn", "--nt=4", "--h1"], ["tools/gen_h3_attn_asm.py", "--nt=4", "--h1"],
                 ["tools/gen_h3_ffn_asm.py", "--shape=in", "--nt=4", "--h1"], ["tools/gen_h3_ffn_asm.py", "--shape=out", "--nt=4", "--h1"],
                 # r05: the encoder stack of the 64-token build as one statement, tw_h?n4_enc_*
                 ["tools/gen_h3_enc_asm.py", "--nt=4"], ["tools/gen_h3_enc_asm.py", "--nt=4", "--h1"],
                 # ... and of the wide layout (five- / three- / six-group key windows), tw_h?w{,3,6}_enc_*
                 ["tools/gen_h3_enc_asm.py", "--wide"], ["tools/gen_h3_enc_asm.py", "--wide", "--h1"],
                 ["tools/gen_h3_enc_asm.py", "--wide", "--ng=3"], ["tools/gen_h3_enc_asm.py", "--wide", "--ng=3", "--h1"],
                 ["tools/gen_h3_enc_asm.py", "--wide", "--ng=6"], ["tools/gen_h3_enc_asm.py", "--wide", "--ng=6", "--h1"],
                 # ... and of the dense softmax model, tw_h?d_enc_*
                 ["tools/gen_h3_enc_asm.py", "--dense"], ["tools/gen_h3_enc_asm.py", "--dense", "--h1"],
                 # r06: ... and on 64-token waves (49-64 atoms): tw_h3n4d_enc_*
                 ["tools/gen_h3_enc_asm.py", "--dense", "--nt=4"],
                 # ... and of the paired 64-token layout (97-128 atoms), tw_h?n4p_enc_*
                 ["tools/gen_h3_enc_asm.py", "--nt=4", "--pair"], ["tools/gen_h3_enc_asm.py", "--nt=4", "--pair", "--h1"]):
        subprocess.run([sys.executable] + args + [f"--out-dir={tmp_path}"], cwd=root, check=True, env=env,
                       stdout=subprocess.DEVNULL)
    names = sorted(os.listdir(tmp_path))
    assert len(names) == 85
    for n in names:
        with open(os.path.join(tmp_path, n)) as a, open(os.path.join(root, "timewarp_amd", "csrc", n)) as b:
            assert a.read() == b.read(), n


def test_euler_maruyama_baseline_and_sample_driver():
    """BASELINE config 0 (gaussian_baseline.yaml plumbing): the product's EulerMaruyamaGaussian against the
    reference's distribution parameters and log-likelihood (golden euler_maruyama.npz), and the `sample.py` driver
    (utils/sampling_utils.sample) on the reference's smallest test molecule: 100 conditional samples, one at a time."""
    import timewarp_amd as tw
    from timewarp_amd.dataloader import DenseMolDynBatch, ELEMENT_VOCAB
    from timewarp_amd.utils.sampling_utils import sample, sample_from_trajectory

    d, sd = H.load("euler_maruyama")
    m = tw.model_constructor(tw.ModelConfig(model_type="euler_maruyama_gaussian")).eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        pc, pv = m._get_y_dist(d["atom_types"], d["x_coords"], d["x_velocs"], d["x_forces"])
        ll = m.log_likelihood(atom_types=d["atom_types"], x_coords=d["x_coords"], x_velocs=d["x_velocs"], x_forces=d["x_forces"],
                              y_coords=d["y_coords"], y_velocs=d["y_velocs"], adj_list=None, edge_batch_idx=None,
                              masked_elements=d["masked"])
    for got, key in ((pc.loc, "coord_mean"), (pc.scale, "coord_std"), (pv.loc, "veloc_mean"), (pv.scale, "veloc_std"), (ll, "loglik")):
        assert H.rel_err(got, d[key]) < 2e-6, key
    cm, cs, vm, vs = fo.euler_maruyama_dist(sd, d["atom_types"], d["x_coords"], d["x_velocs"], d["x_forces"])
    assert H.rel_err(pc.loc, cm) < 2e-6 and H.rel_err(pv.scale, vs) < 2e-6

    z = np.load(os.path.join(H.GOLDEN, "smallest_molecule.npz"))
    at = torch.tensor([[ELEMENT_VOCAB[e] for e in z["elements"]]])
    x, v, f = (torch.from_numpy(z[k][:1]) for k in ("positions", "velocities", "forces"))
    V = x.shape[1]
    batch = DenseMolDynBatch(names=["2olx"], atom_types=at, adj_list=torch.zeros((0, 2), dtype=torch.int64),
                             edge_batch_idx=torch.zeros((0,), dtype=torch.int64), atom_coords=x, atom_velocs=v, atom_forces=f,
                             atom_coord_targets=x, atom_veloc_targets=v, atom_force_targets=f,
                             masked_elements=torch.zeros(1, V, dtype=torch.bool))
    torch.manual_seed(0)
    yc, yv = sample(m, batch, 100)
    assert yc.shape == (100, V, 3) and yv.shape == (100, V, 3) and yc.dtype == np.float64
    with torch.no_grad():
        pc, pv = m._get_y_dist(at, x, v, f)
    # sample mean within 5 standard errors of the analytic mean, everywhere
    assert (np.abs(yc.mean(0) - pc.loc[0].numpy()) < 5 * pc.scale[0].numpy() / 10 + 1e-6).all()
    assert (np.abs(yv.mean(0) - pv.loc[0].numpy()) < 5 * pv.scale[0].numpy() / 10 + 1e-6).all()
    cs_, vs_ = sample_from_trajectory(m, [batch, batch], 3)
    assert len(cs_) == 2 and cs_[0].shape == (3, V, 3)


# ---------------------------------------------------------------------------------------------
# The drop-in energy route: an openmm.System as the reference's scripts build it -> tw_forcefield tables
# (evaluate.py:290-301, sample_trajectory.py:190-202, utils/openmm/openmm_bridge.py:252-307).  OpenMM is not in this
# image: the stand-ins below carry the OpenMM 7.7 getters the reader uses and return Quantity-like values in
# NON-MD units (Angstrom, kcal/mol, degrees), so the unit conversion is exercised as well.
# ---------------------------------------------------------------------------------------------
class _Q:
    """Quantity stand-in: `to_md` = factor from the unit the value is written in to OpenMM's MD unit system."""

    def __init__(self, value, to_md):
        self.value, self.to_md = value, to_md

    def value_in_unit_system(self, system):
        assert system == "md_unit_system"
        return self.value * self.to_md


def _install_fake_openmm_unit(monkeypatch):
    import sys
    import types

    mm, unit = types.ModuleType("openmm"), types.ModuleType("openmm.unit")
    unit.md_unit_system = "md_unit_system"
    mm.unit = unit
    monkeypatch.setitem(sys.modules, "openmm", mm)
    monkeypatch.setitem(sys.modules, "openmm.unit", unit)


def _mock_openmm_system(t, extra_forces=(), obc1_custom=False, method=1):
    ANG, KCAL, DEG = 0.1, 4.184, np.pi / 180.0

    class HarmonicBondForce:
        def getNumBonds(self): return len(t.bond_idx)
        def getBondParameters(self, b):
            return int(t.bond_idx[b, 0]), int(t.bond_idx[b, 1]), _Q(t.bond_par[b, 0] / ANG, ANG), _Q(t.bond_par[b, 1] / (KCAL / ANG**2), KCAL / ANG**2)

    class HarmonicAngleForce:
        def getNumAngles(self): return len(t.angle_idx)
        def getAngleParameters(self, a):
            return (*map(int, t.angle_idx[a]), _Q(t.angle_par[a, 0] / DEG, DEG), _Q(t.angle_par[a, 1] / KCAL, KCAL))

    class PeriodicTorsionForce:
        def getNumTorsions(self): return len(t.torsion_idx)
        def getTorsionParameters(self, i):
            return (*map(int, t.torsion_idx[i]), int(t.torsion_par[i, 0]), _Q(t.torsion_par[i, 1] / DEG, DEG), _Q(t.torsion_par[i, 2] / KCAL, KCAL))

    class NonbondedForce:
        NoCutoff, CutoffNonPeriodic, CutoffPeriodic = 0, 1, 2
        def getNonbondedMethod(self): return method
        def getNumParticles(self): return t.n_atoms
        def getParticleParameters(self, i):
            return _Q(t.atom_par[i, 0], 1.0), _Q(t.atom_par[i, 1] / ANG, ANG), _Q(t.atom_par[i, 2] / KCAL, KCAL)
        def getNumExceptions(self): return len(t.exc_idx)
        def getExceptionParameters(self, e):
            return int(t.exc_idx[e, 0]), int(t.exc_idx[e, 1]), _Q(t.exc_par[e, 0], 1.0), _Q(t.exc_par[e, 1] / ANG, ANG), _Q(t.exc_par[e, 2] / KCAL, KCAL)
        def getCutoffDistance(self): return _Q(t.cutoff / ANG, ANG)
        def getReactionFieldDielectric(self): return t.rf_dielectric

    class GBSAOBCForce:
        def getNumParticles(self): return t.n_atoms
        def getParticleParameters(self, i): return _Q(t.atom_par[i, 0], 1.0), _Q(t.atom_par[i, 3] / ANG, ANG), t.atom_par[i, 4]
        def getSoluteDielectric(self): return t.solute_dielectric
        def getSolventDielectric(self): return t.solvent_dielectric
        def getSurfaceAreaEnergy(self): return _Q(t.surface_area_energy / (KCAL / ANG**2), KCAL / ANG**2)

    class CustomGBForce:  # GBSA-OBC I as openmm.app.internal.customgbforces.GBSAOBC1Force lays it out (plain floats)
        def getNumComputedValues(self): return 2
        def getComputedValueParameters(self, i):
            return [("I", "step(r+sr2-or1)*0.5*(1/L-1/U+0.25*(r-sr2^2/r)*(1/(U^2)-1/(L^2))+0.5*log(L/U)/r); ...", 1),
                    ("B", "1/(1/or-tanh(0.8*psi+2.909125*psi^3)/radius); psi=I*or; radius=or+offset; offset=0.009", 0)][i]
        def getNumEnergyTerms(self): return 2
        def getEnergyTermParameters(self, i):
            return [("28.3919551*(radius+0.14)^2*(radius/B)^6-0.5*138.935485*(1/soluteDielectric-1/solventDielectric)*charge^2/B; radius=or+offset; offset=0.009", 0),
                    ("-138.935485*(1/soluteDielectric-1/solventDielectric)*charge1*charge2/f; ...", 2)][i]
        def getNumPerParticleParameters(self): return 3
        def getPerParticleParameterName(self, i): return ["charge", "or", "sr"][i]
        def getNumParticles(self): return t.n_atoms
        def getParticleParameters(self, i):
            o_r = t.atom_par[i, 3] - 0.009
            return (t.atom_par[i, 0], o_r, t.atom_par[i, 4] * o_r)
        def getNumGlobalParameters(self): return 2
        def getGlobalParameterName(self, i): return ["solventDielectric", "soluteDielectric"][i]
        def getGlobalParameterDefaultValue(self, i): return [t.solvent_dielectric, t.solute_dielectric][i]

    class CMMotionRemover:
        pass

    forces = [HarmonicBondForce(), HarmonicAngleForce(), PeriodicTorsionForce(), NonbondedForce(),
              CustomGBForce() if obc1_custom else GBSAOBCForce(), CMMotionRemover(), *extra_forces]

    class System:
        def getNumParticles(self): return t.n_atoms
        def getForces(self): return forces

    return System()


def _tables_equal(a, b):
    for f in ("bond_idx", "angle_idx", "torsion_idx", "exc_idx"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    for f in ("bond_par", "angle_par", "torsion_par", "exc_par", "atom_par"):
        assert np.allclose(getattr(a, f), getattr(b, f), rtol=1e-12, atol=1e-15), f
    for f in ("has_gbsa", "cutoff", "rf_dielectric", "solute_dielectric", "solvent_dielectric", "surface_area_energy"):
        assert getattr(a, f) == pytest.approx(getattr(b, f), rel=1e-12), f


def test_tables_from_openmm_system_round_trip(monkeypatch):
    """System -> tables must give back alanine_dipeptide_amber99sb() exactly (indices) / to rounding (unit-converted
    parameters), for the GBSAOBCForce preset and for GBSA-OBC I delivered as a CustomGBForce."""
    import dataclasses
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.forcefield import alanine_dipeptide_amber99sb, tables_from_openmm_system

    _install_fake_openmm_unit(monkeypatch)
    t = alanine_dipeptide_amber99sb()
    _tables_equal(tables_from_openmm_system(_mock_openmm_system(t)), t)
    t1 = dataclasses.replace(t, has_gbsa=2, solvent_dielectric=78.5)
    _tables_equal(tables_from_openmm_system(_mock_openmm_system(t1, obc1_custom=True)), t1)
    # NoCutoff: the cutoff stays 0 (= none)
    assert tables_from_openmm_system(_mock_openmm_system(t, method=0)).cutoff == 0.0

    class LangevinIntegrator:
        def getTemperature(self): return _Q(36.85, 1.0)  # value already in kelvin here

    integ = LangevinIntegrator()
    e = AmberPotentialEnergyTorch.from_openmm(_mock_openmm_system(t), integ, platform_name="CUDA", platform_properties={})
    assert e.num_particles == 22 and e.get_integrator() is integ
    assert e.kbT == pytest.approx(8.314462618e-3 * 36.85)
    _tables_equal(e.tables, t)
    assert AmberPotentialEnergyTorch.alanine_dipeptide().kbT == pytest.approx(2.57748, abs=1e-5)


def test_tables_from_openmm_system_refuses_unknown_forces(monkeypatch):
    """A force whose energy the kernel would not evaluate is an error (it would bias the acceptance silently)."""
    from timewarp_amd.forcefield import alanine_dipeptide_amber99sb, tables_from_openmm_system

    _install_fake_openmm_unit(monkeypatch)
    t = alanine_dipeptide_amber99sb()
    for name in ("CMAPTorsionForce", "CustomTorsionForce", "CustomNonbondedForce", "CustomBondForce"):
        with pytest.raises(NotImplementedError, match=name):
            tables_from_openmm_system(_mock_openmm_system(t, extra_forces=(type(name, (), {})(),)))
    tables_from_openmm_system(_mock_openmm_system(t, extra_forces=(type("MonteCarloBarostat", (), {})(),)))  # energy-free
    with pytest.raises(NotImplementedError, match="periodic"):
        tables_from_openmm_system(_mock_openmm_system(t, method=2))
    with pytest.raises(NotImplementedError):
        tables_from_openmm_system(_mock_openmm_system(t, obc1_custom=True), allow_custom_gb_obc1=False)


def test_energy_callable_checks_shapes():
    from timewarp_amd.energy import AmberPotentialEnergyTorch

    e = AmberPotentialEnergyTorch.alanine_dipeptide()
    with pytest.raises(AssertionError, match="number of particles"):
        e(torch.zeros(3, 21, 3))
    with pytest.raises(AssertionError, match="size 3"):
        e(torch.zeros(3, 22, 2))


def test_openmm_step_drives_the_callers_simulation():
    """evaluation_utils.py:439-466: positions (and velocities, or a temperature draw) in, `num_steps` steps, float32
    tensors shaped like the input back; nothing but the Simulation's own methods is touched."""
    from oracle import mh_oracle as mo
    from oracle.fake_sim import FakeSimulation
    from timewarp_amd.utils.evaluation_utils import openmm_step

    g = torch.Generator().manual_seed(3)
    c, v = torch.randn(1, 7, 3, generator=g) * 0.3, torch.randn(1, 7, 3, generator=g)
    a = openmm_step(FakeSimulation(), c, v, num_steps=4)
    b = mo.openmm_step(FakeSimulation(), c, v, num_steps=4)
    assert a[0].shape == c.shape and a[0].dtype == torch.float32
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert not torch.equal(a[0], c)

    class Integrator:
        def getTemperature(self):
            return 310.0

    class ThermalSim(FakeSimulation):
        def __init__(self):
            super().__init__()
            ctx = self.context
            ctx.setVelocitiesToTemperature = lambda t: ctx.setVelocities(np.full(ctx.pos.shape, t / 310.0))

    w = openmm_step(ThermalSim(), c, None, num_steps=1, integrator=Integrator())
    assert torch.isfinite(w[0]).all() and w[1].shape == c.shape
    with pytest.raises(ValueError):
        openmm_step(FakeSimulation(), c, None)


def test_wrong_result_debug_switches_are_not_in_the_product_library():
    """r03 review: tw_debug_set_flags bits 0, 1, 6, 7 (and 11) are timing experiments that make results wrong.  The
    product build refuses them; only a -DTW_EXPERIMENTS developer build has them."""
    from timewarp_amd import _lib

    lib = _lib.load()
    try:
        for bit in (1, 2, 64, 128, 2048, 1 | 8):
            assert lib.tw_debug_set_flags(bit) != 0
            assert b"TW_EXPERIMENTS" in lib.tw_last_error()
        for bit in (4, 8, 16, 32, 1024, 4096, 8192, 16384, 32768):
            assert lib.tw_debug_set_flags(bit) == 0
    finally:
        assert lib.tw_debug_set_flags(0) == 0


def test_split_fp16_workspace_covers_every_layout_a_launch_can_take():
    """the split-fp16 kernel picks its layout per launch (48-token waves, 64-token waves, wide) from the row
    count and tw_debug_set_flags; a workspace sized once for up to n_rows rows must hold whichever is taken.  The sizing
    takes the maximum over the layouts that exist for the size (the 64-token one explicitly since r05), so it must not
    depend on the flags, must grow with the row count, and every size from 1 to 192 atoms must be supported."""
    import timewarp_amd as tw
    from timewarp_amd import _lib, synthetic

    lib = _lib.load()
    d = tw.model_constructor(synthetic.kernel_transformer_nvp_config()).dims.to_desc()
    flags = (0, 16384, 32768, 65536, 131072, 262144, 32768 | 262144, 65536 | 4096, 1048576, 4096)
    try:
        for V in list(range(1, 70)) + [80, 96, 97, 128, 160, 161, 192]:
            assert lib.tw_flow_path_supported(C.byref(d), V, _lib.TW_PATH_FUSED_H3) == 1, V
            sizes = []
            for f in flags:
                lib.tw_debug_set_flags(f)
                sizes.append([lib.tw_flow_workspace_bytes(C.byref(d), n, V) for n in (1, 100, 512, 1000)])
            assert all(s == sizes[0] for s in sizes), (V, sizes)
            assert sizes[0] == sorted(sizes[0]) and sizes[0][0] > 0, (V, sizes[0])
        assert lib.tw_flow_path_supported(C.byref(d), 193, _lib.TW_PATH_FUSED_H3) == 0
    finally:
        lib.tw_debug_set_flags(0)


# Instantiations a flow call can select that may use scratch, by name, with the scratch they may use.  Empty since r06: the dense
# model's 64-token build was the last one (r05: its compiled attention block, 724 B/lane) until its encoder-stack statement
# (tools/gen_h3_enc_asm.py --dense --nt=4) replaced it.
KNOWN_SPILLING = {}


def selectable_netblock_kernels():
    from timewarp_amd import _lib

    """Every net-block instantiation the launch code can select WITHOUT debug flags - asked of the library itself
    (tw_flow_selected_kernel: the launch branch run dry), for 1 .. 192 atoms x several row counts (the layout choice counts
    rounds of the chip) x {kernel attention, dense softmax, dense + position features} x {split-fp16, single-MFMA}."""
    import timewarp_amd as tw
    from timewarp_amd import synthetic

    lib = _lib.load()
    lib.tw_debug_set_flags(0)
    rff = synthetic.transformer_nvp_config()
    rff.transformer_nvp_config.rff_position_encoder_config = tw.RFFPositionEncoderConfig(128, 1.0, 1.0)
    picked = {}
    for tag, cfg in (("kernel", synthetic.kernel_transformer_nvp_config()), ("dense", synthetic.transformer_nvp_config()), ("dense+rff", rff)):
        d = tw.model_constructor(cfg).dims.to_desc()
        for path, pn in ((_lib.TW_PATH_FUSED_H3, "h3"), (_lib.TW_PATH_FUSED_H1, "h1")):
            for V in range(1, 193):
                for rows in (1, 100, 512, 1000, 4096):
                    name = lib.tw_flow_selected_kernel(C.byref(d), V, rows, path).decode()
                    if name:
                        picked.setdefault(name, set()).add((tag, pn, V))
    return picked


def test_product_kernels_do_not_spill():
    """Every instantiation of the split-fp16 / single-MFMA net-block kernel that a flow call can SELECT compiles to ScratchSize
    0 B/lane, and hipcc has nothing to say about the inline asm.  r05 defined "product" as the ENC = true template argument and
    so could not see the one selectable kernel that spills; now the library is asked which instantiations
    its launch code takes (tw_flow_selected_kernel); exceptions would be listed by name with their scratch size pinned (none left).
    Compiles csrc/tw_netblock_h3.hip once with -Rpass-analysis=kernel-resource-usage (~90 s, no GPU needed)."""
    import shutil
    import subprocess
    import sys

    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    picked = selectable_netblock_kernels()
    # the families that exist: 48-token, wide (5 / 6 groups), 64-token, paired - each on both paths; dense 48-token on both, with
    # position features on both, dense 64-token on the split path
    assert len(picked) >= 15, sorted(picked)
    atoms = lambda name: sorted({v for _, _, v in picked[name]})
    assert atoms("tw::netblock_h3_kernel<4, true, false, true, false, true, false, false>") == list(range(97, 129))      # paired
    assert atoms("tw::netblock_h3_kernel<3, true, false, true, false, true, false, true>") == list(range(161, 193))       # six groups
    assert atoms("tw::netblock_h3_kernel<4, true, true, false, false, true, false, false>") == list(range(49, 65))        # dense, 64-token waves
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "resource_usage.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r"\s*(\d+)\s+(\d+)\s+(-?\d+)\s+(\d+)\s+(tw::netblock_h3_kernel<[^>]*>)", line)
        if m:
            rows[m.group(5)] = int(m.group(4))
    missing = [k for k in picked if k not in rows]
    assert not missing, missing                         # every selectable name is an instantiation the compiler reported on
    spills = {k: rows[k] for k in picked if rows[k] != 0}
    assert set(spills) <= set(KNOWN_SPILLING), spills
    assert all(v <= KNOWN_SPILLING[k] for k, v in spills.items()), spills
    assert "# warnings: 0" in out.stdout, out.stdout[-1500:]


def test_philox_restatement_known_answers():
    """tests/helpers.philox4x32_10 - the numpy restatement the GPU suite holds tw_mh_iteration_chains' generator to - against
    the known-answer vectors published with the algorithm (Random123 kat_vectors: philox4x32 10)."""
    import numpy as np

    def run(c, k):
        return [int(x) for x in H.philox4x32_10(np.array([c], dtype=np.uint64), np.array([k], dtype=np.uint64))[0]]

    assert run([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    # streams: distinct (chain, iteration, kind) give unrelated words; the normals are standard
    a = H.chain_draw_words(7, 0, 0, 0, 64)
    assert not np.array_equal(a, H.chain_draw_words(7, 0, 1, 0, 64)) and not np.array_equal(a, H.chain_draw_words(7, 1, 0, 0, 64))
    assert not np.array_equal(a, H.chain_draw_words(7, 0, 0, 1, 64)) and not np.array_equal(a, H.chain_draw_words(8, 0, 0, 0, 64))
    n = H.chain_draw_normals(5, 3, 1, 0, 200000)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01 and abs((n ** 4).mean() - 3) < 0.1
    u = H.chain_draw_uniforms(5, 3, 1, 100000)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def test_range_guard_words_do_not_travel_with_a_copied_or_saved_model():
    """the per-device range-guard words (and the packed device weights / workspace) are process-local scratch - a
    deep copy or a pickle of the model must not carry them, and must still hold every parameter."""
    import copy
    import io

    import timewarp_amd as tw
    from timewarp_amd import synthetic

    m = tw.model_constructor(synthetic.kernel_transformer_nvp_config())
    m._range_flags[0] = torch.zeros(1, dtype=torch.int32)     # stands in for the device word
    m._workspace = torch.zeros(8, dtype=torch.uint8)
    m._dirty = False
    c = copy.deepcopy(m)
    assert c._range_flags == {} and c._workspace is None and c._dev_weights is None and c._dirty
    assert m._range_flags and m._workspace is not None        # the original keeps its own
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert r._range_flags == {} and r._workspace is None and r._dirty
    sd, sd_c, sd_r = m.state_dict(), c.state_dict(), r.state_dict()
    assert list(sd) == list(sd_c) == list(sd_r) and all(torch.equal(sd[k], sd_c[k]) and torch.equal(sd[k], sd_r[k]) for k in sd)
    assert c.dims == m.dims and r.dims == m.dims
