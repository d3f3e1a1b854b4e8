"""The dataset / preset -> force-field mapping of simulation/md.py:31-37, 153-159 without OpenMM.  amber99sb-ildn +
GBSA-OBC II is pinned by the reference's known-answer file (tests/test_energy_kat.py).  amber14-all + GBSA-OBC I (the
T1B-peptides / 4AA preset) is PARITY UNPINNED - there is nothing of the reference's to hold it against - so what is
checked here is what can be: charge sums, that every parameter resolves, that the amber14 tables differ from the pinned
amber99 ones exactly where ff14SB and obc1 differ, the refusal of residues whose ff14SB side-chain torsions are not in
the file, and (GPU) the kernel against the C oracle and whole MH iterations against the oracle loop on a 52-atom
capped tetra-alanine."""
import numpy as np
import pytest
import torch

from tests import helpers as H


def tetra_alanine():
    z = np.load(H.GOLDEN + "/tetra_alanine_capped.npz")
    return list(z["atom_names"]), list(z["residue_names"]), [int(i) for i in z["residue_ids"]], z["positions"]


def chain(*residues):
    from timewarp_amd import forcefield as ff

    names, res, rid = [], [], []
    for i, r in enumerate(residues):
        for n in ff._RESIDUES_FF14SB[r]["names"]:
            names.append(n)
            res.append(r[-3:])
            rid.append(i + 1)
    return names, res, rid


def test_amber14_templates_are_neutral_and_resolve():
    from timewarp_amd import forcefield as ff

    for name, tpl in ff._RESIDUES_FF14SB.items():
        q = sum(tpl["charges"].values())
        want = 1.0 if name.startswith("N") and name != "NME" else (-1.0 if name.startswith("C") else 0.0)
        assert abs(q - want) < 2e-4, (name, q)
    for residues, charge in ((("ACE", "ALA", "ALA", "ALA", "ALA", "NME"), 0.0), (("NALA", "GLY", "ALA", "CGLY"), 0.0),
                             (("NGLY", "ALA", "CALA"), 0.0), (("ACE", "GLY", "NME"), 0.0)):
        t = ff.tables_for_preset("T1B-peptides", *chain(*residues))
        assert t.has_gbsa == 2 and abs(t.atom_par[:, 0].sum() - charge) < 1e-3
        assert np.isfinite(t.torsion_par).all() and (t.atom_par[:, 3] > 0.1).all()
        e, terms = H.oracle_energy(t, np.random.default_rng(0).normal(size=(2, t.n_atoms, 3)).astype(np.float32))
        assert np.isfinite(e).all()


def test_amber14_differs_from_the_pinned_amber99_tables_only_where_ff14sb_and_obc1_do():
    from timewarp_amd import forcefield as ff

    rid = [{"ACE": 1, "ALA": 2, "NME": 3}[r] for r in ff.AD_RESIDUES]
    a99 = ff.tables_for_preset("alanine-dipeptide", ff.AD_ATOM_NAMES, ff.AD_RESIDUES, rid)
    a14 = ff.tables_for_preset("T1B-peptides", ff.AD_ATOM_NAMES, ff.AD_RESIDUES, rid)
    assert a99.has_gbsa == 1 and a14.has_gbsa == 2
    for f in ("bond_idx", "bond_par", "angle_idx", "angle_par", "torsion_idx", "exc_idx", "exc_par"):
        assert np.array_equal(getattr(a99, f), getattr(a14, f)), f
    assert np.array_equal(a99.atom_par[:, :3], a14.atom_par[:, :3]) and np.array_equal(a99.atom_par[:, 4], a14.atom_par[:, 4])
    changed = np.flatnonzero((a99.torsion_par != a14.torsion_par).any(axis=1))
    names = ff.AD_ATOM_NAMES
    for i in changed:  # only phi' = C(ACE)-N-CA-CB
        assert [names[a] for a in a14.torsion_idx[i]] in (["C", "N", "CA", "CB"], ["CB", "CA", "N", "C"])
    assert len(changed) == 2
    assert set(np.round(a14.atom_par[:, 3], 3)) == {0.12, 0.13, 0.15, 0.155, 0.17}  # mbondi2
    assert (a14.solvent_dielectric, a99.solvent_dielectric) == (78.5, 78.3)


def test_amber14_refuses_what_it_does_not_know():
    from timewarp_amd import forcefield as ff
    from tests.test_energy_kat import kat

    z = kat()  # NNQQ: fine under amber99, not offered under amber14 (ff14SB refits ASN / GLN side-chain torsions)
    args = (list(z["atom_names"]), list(z["residue_names"]), list(z["residue_ids"]))
    assert ff.tables_for_preset("T1-peptides", *args).n_atoms == 65
    with pytest.raises(NotImplementedError, match="amber14"):
        ff.tables_for_preset("T1B-peptides", *args)
    with pytest.raises(ValueError, match="unknown dataset"):
        ff.tables_for_preset("amber14-explicit", *args)


@pytest.mark.gpu
def test_amber14_obc1_kernel_vs_c_oracle_on_tetra_alanine():
    from timewarp_amd.energy import AmberPotentialEnergyTorch

    names, res, rid, pos = tetra_alanine()
    e = AmberPotentialEnergyTorch.from_preset("T1B-peptides", names, res, rid)
    assert e.tables.n_atoms == 52 and e.tables.has_gbsa == 2
    g = torch.Generator().manual_seed(2)
    x = torch.from_numpy(pos)[None] + torch.randn(32, 52, 3, generator=g) * 0.01
    out, terms = e.energy_and_terms(x.cuda(), want_terms=True)
    ref, ref_terms = H.oracle_energy(e.tables, x.numpy())
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-10, atol=1e-8)
    assert np.allclose(terms.cpu().numpy(), ref_terms, rtol=1e-9, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("path", [0, 3])
def test_mh_iterations_on_tetra_alanine_amber14_vs_oracle(path):
    """Whole MH iterations under the 4AA preset (amber14 + GBSA-OBC I tables, parity unpinned) on capped tetra-alanine,
    52 atoms: fused f32 kernel / split-fp16 wide layout, energy kernel in OBC-I mode, tw_mh_iteration - against the
    oracle loop with the C energy oracle on the same tables and noise."""
    from oracle import mh_oracle as mo
    from tests.test_mh_gpu import _assert_chain_matches_oracle
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.forcefield import ELEMENT_MASSES
    from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain, sample_with_model

    names, res, rid, pos = tetra_alanine()
    energy = AmberPotentialEnergyTorch.from_preset("T1B-peptides", names, res, rid)
    vocab = {"C": 0, "H": 1, "N": 2, "O": 3, "S": 4}
    types = torch.tensor([vocab[n[0]] for n in names])
    masses = torch.tensor([ELEMENT_MASSES[n[0]] for n in names], dtype=torch.float32)
    coords = torch.from_numpy(pos).float()
    sd = H.mh_state_dict("scaled", True, out_scale=3e-5, coords_log_scale=-7.5)
    S, N = 32, 60
    kw = dict(accept=True, num_proposal_steps=S, random_velocs=True, resample_velocs=True)
    ref = mo.sample_with_model(types[None], coords[None], torch.zeros(1, 52, 3), torch.zeros(1, 52, dtype=torch.bool),
                               mo.OracleModel(sd, H.FULL_KERNEL_SPEC), H.OracleAmberEnergy(energy.tables), masses, N,
                               H.HostNoise(3), **kw)
    model = H.tw_kernel_model(sd, path=path)
    dev = torch.device("cuda")
    chain = MetropolisHastingsChain(single_state_batch("a4", types, coords), model, dev, energy, masses,
                                    noise=H.HostNoise(3, "cuda"), **kw)
    assert chain._fused
    got = sample_with_model(single_state_batch("a4", types, coords), model, dev, energy, masses, N, disable_tqdm=True,
                            noise=H.HostNoise(3, "cuda"), **kw)
    assert ref[2] >= 1
    H.assert_not_demoted(model)
    _assert_chain_matches_oracle(got, ref, tol=1e-5, stat_tol=2e-4)


def test_amber14_preset_says_that_it_is_unpinned():
    """an energy built from the amber14 preset carries parity = "unpinned" and warns once per process; the
    amber99 presets (pinned on the reference's OpenMM known-answer files) do neither."""
    import warnings

    from timewarp_amd import energy as E
    from timewarp_amd.forcefield import AD_ATOM_NAMES, AD_RESIDUES

    rid = [{"ACE": 1, "ALA": 2, "NME": 3}[r] for r in AD_RESIDUES]
    E._UNPINNED_WARNED[0] = False
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        a99 = E.AmberPotentialEnergyTorch.from_preset("alanine-dipeptide", AD_ATOM_NAMES, AD_RESIDUES, rid)
        assert a99.parity == "pinned" and not seen
        a14 = E.AmberPotentialEnergyTorch.from_preset("T1B-peptides", AD_ATOM_NAMES, AD_RESIDUES, rid)
        assert a14.parity == "unpinned" and len(seen) == 1 and "UNPINNED" in str(seen[0].message)
        E.AmberPotentialEnergyTorch.from_preset("amber14-implicit", AD_ATOM_NAMES, AD_RESIDUES, rid)
        assert len(seen) == 1                      # once per process
    assert E.AmberPotentialEnergyTorch.alanine_dipeptide().parity == "pinned"
