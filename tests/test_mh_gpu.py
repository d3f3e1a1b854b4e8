"""The product MH loop (HIP) replayed on the reference's recorded traces, plus the energy kernel
against the C oracle."""
import numpy as np
import pytest
import torch

from oracle import mh_oracle as mo
from tests import helpers as H
from tests.test_mh_oracle import SCENARIOS, check_against_golden, load_mh, replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_sample_with_model_replays_reference(name):
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.evaluation_utils import sample_with_model

    z, sd = load_mh()
    kw = dict(SCENARIOS[name])
    extra = {}
    if kw.pop("chirality", False):
        extra = dict(chirality_centers=torch.from_numpy(z["centres"]),
                     reference_signs=torch.from_numpy(z[name + "/reference_signs"]))
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    batch = single_state_batch("tiny", torch.from_numpy(z["atom_types"]), x0, v0)
    energy = mo.SyntheticEnergy(x0.clone().cuda())  # a torch callable with .kbT: the loop's only requirement
    coords, velocs, accepted, stats = sample_with_model(
        batch, model, torch.device("cuda"), energy, torch.from_numpy(z["masses"]), disable_tqdm=True,
        noise=replay(z, name, device="cuda"), **kw, **extra)
    check_against_golden(z, name, coords, velocs, accepted, stats)


oracle_energy = H.oracle_energy


@pytest.mark.parametrize("gb", [1, 2, 0])
def test_amber_energy_kernel_vs_c_oracle(gb):
    """gb = 1: GBSA-OBC II (amber99_obc.xml), 2: GBSA-OBC I (implicit/obc1.xml coefficients), 0: no implicit solvent."""
    import dataclasses
    from timewarp_amd.energy import AmberPotentialEnergyTorch

    e = AmberPotentialEnergyTorch.alanine_dipeptide()
    assert abs(e.kbT - 2.57748) < 1e-4  # R * 310 K
    if gb != 1:
        e = AmberPotentialEnergyTorch(dataclasses.replace(e.tables, has_gbsa=gb))
    d, _ = H.load("kernel_full_ad")
    g = torch.Generator().manual_seed(1)
    x = d["x_coords"] + torch.randn(64, 22, 3, generator=g) * 0.01
    x[5] = d["x_coords"][0] * 3.0  # stretched: some pairs beyond the 2 nm cutoff
    out, terms = e.energy_and_terms(x.cuda(), want_terms=True)
    ref, ref_terms = oracle_energy(e.tables, x.numpy())
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-10, atol=1e-8)
    assert np.allclose(terms.cpu().numpy(), ref_terms, rtol=1e-9, atol=1e-8)
    assert e(x.cuda()).shape == (64, 1) and e(x.cuda()).dtype == torch.float32


def test_accept_kernel_first_index_and_clipping():
    from timewarp_amd.utils.evaluation_utils import _mh_accept

    S = 1000
    g = torch.Generator().manual_seed(0)
    energy = torch.randn(S, generator=g) * 3 + 4
    pxy, pyx, u = torch.randn(S, generator=g), torch.randn(S, generator=g), torch.rand(S, generator=g)
    ex, p_acc, acc, res = _mh_accept(energy.cuda(), pxy.cuda(), pyx.cuda(), u.cuda())
    e_ref = energy + pxy - pyx
    p_ref = torch.clamp(torch.exp(-e_ref), max=1.0)
    a_ref = u < p_ref
    assert torch.allclose(ex.cpu(), e_ref) and torch.allclose(p_acc.cpu(), p_ref, rtol=1e-6)
    assert (acc.cpu().bool() == a_ref).all()
    assert int(res[0]) == int(a_ref.nonzero()[0]) and int(res[1]) == 1
    ex, p_acc, acc, res = _mh_accept((energy + 1e4).cuda(), pxy.cuda(), pyx.cuda(), u.cuda())
    assert int(res[0]) == S - 1 and int(res[1]) == 0


@pytest.mark.parametrize("tag,random_velocs", [("fixedv", False), ("randv", True)])
def test_sample_on_batches_replays_reference(tag, random_velocs):
    """Product sample_on_batches against vectors recorded from the reference's own function."""
    from tests.test_mh_oracle import check_sob, load_sob, sob_replay
    from timewarp_amd.dataloader import DenseMolDynBatch
    from timewarp_amd.utils.evaluation_utils import sample_on_batches

    z, sd, raw = load_sob()
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    batches = []
    for b, r in enumerate(raw):
        zero = torch.zeros_like(r["x"])
        batches.append(DenseMolDynBatch(
            names=[f"tiny{b}"], atom_types=r["atom_types"], adj_list=torch.zeros((0, 2), dtype=torch.int64),
            edge_batch_idx=torch.zeros((0,), dtype=torch.int64), atom_coords=r["x"], atom_velocs=r["v"], atom_forces=zero,
            atom_coord_targets=r["y"], atom_veloc_targets=r["w"], atom_force_targets=zero,
            masked_elements=torch.zeros(1, r["x"].shape[1], dtype=torch.bool)))
    energy = mo.SyntheticEnergy(torch.from_numpy(z["x_ref"]).clone().cuda())
    res = sample_on_batches(batches, model, torch.device("cuda"), energy, False, torch.from_numpy(z["masses"]),
                            random_velocs=random_velocs, noise=sob_replay(z, tag, device="cuda"))
    check_sob(z, tag, res)


def test_sample_trajectory_writes_and_resumes(tmp_path):
    """Two segments of a real MH chain on the GPU, then a resumed third one."""
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.sample_trajectory import sample_trajectory, segment_path

    z, sd = load_mh()
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    batch = single_state_batch("tiny", torch.from_numpy(z["atom_types"]), x0, v0)
    energy = mo.SyntheticEnergy(x0.clone().cuda())
    out = str(tmp_path / "traj")
    masses = torch.from_numpy(z["masses"])
    assert sample_trajectory(batch, model, torch.device("cuda"), energy, masses, out, "tiny", 40, 20, mh=True,
                             num_proposal_steps=10, verbose=False) == 2
    p1 = np.load(segment_path(out, "tiny", 1))["positions"]
    assert p1.shape[1:] == (7, 3) and p1.shape[0] >= 3 and np.isfinite(p1).all()
    batch2 = single_state_batch("tiny", torch.from_numpy(z["atom_types"]), x0, v0)
    assert sample_trajectory(batch2, model, torch.device("cuda"), energy, masses, out, "tiny", 60, 20, mh=True,
                             num_proposal_steps=10, verbose=False) == 1
    p2 = np.load(segment_path(out, "tiny", 2))["positions"]
    assert np.allclose(p2[0], p1[-1])  # resumed from the last saved row


def test_deferred_iterations_equal_synchronous_ones():
    """sample_with_model with the accept results read back every 4 iterations (the chain state moved on the
    device by tw_mh_accept) gives bit-identical chains and statistics to one read-back per iteration."""
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.evaluation_utils import sample_with_model

    z, sd = load_mh()
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    energy = mo.SyntheticEnergy(x0.clone().cuda())
    masses = torch.from_numpy(z["masses"])
    outs = []
    for sync_every in (1, 4):
        for kw in (dict(), dict(random_velocs=True, resample_velocs=True)):
            batch = single_state_batch("tiny", torch.from_numpy(z["atom_types"]), x0, v0)
            torch.manual_seed(99)
            outs.append(sample_with_model(batch, model, torch.device("cuda"), energy, masses, 120, accept=True,
                                          num_proposal_steps=10, disable_tqdm=True, sync_every=sync_every, **kw))
    for a, b in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert a[0].shape == b[0].shape and a[2] == b[2] and a[2] > 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for f in ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin",
                  "energies_pot_delta", "energies_kin_delta"):
            assert np.array_equal(getattr(a[3], f), getattr(b[3], f)), f


def test_multichain_equals_single_chains():
    """Three chains evaluated in lock-step (one flow call of 3 x S rows per iteration, per-chain accept on the device)
    reproduce, chain by chain and bit for bit, three separate sample_with_model runs driven by the same per-chain
    noise streams - states, velocities, accept counts and all nine ChainStats arrays, including the final clip."""
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.evaluation_utils import DeviceNoise, sample_with_model
    from timewarp_amd.utils.multichain import sample_with_model_chains

    z, sd = load_mh()
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    at = torch.from_numpy(z["atom_types"])
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    g = torch.Generator().manual_seed(31)
    starts = [(x0 + 0.03 * torch.randn(x0.shape, generator=g), v0 + 0.1 * torch.randn(v0.shape, generator=g)) for _ in range(3)]
    energy = mo.SyntheticEnergy(x0.clone().cuda())
    masses = torch.from_numpy(z["masses"])
    dev = torch.device("cuda")
    N, S = 70, 10
    for kw in (dict(), dict(random_velocs=True, resample_velocs=True)):
        singles = []
        for c, (xc, vc) in enumerate(starts):
            singles.append(sample_with_model(single_state_batch("t", at, xc, vc), model, dev, energy, masses, N, accept=True,
                                             num_proposal_steps=S, disable_tqdm=True, noise=DeviceNoise(dev, seed=500 + c), **kw))
        multi = sample_with_model_chains([single_state_batch("t", at, xc, vc) for xc, vc in starts], model, dev, energy, masses,
                                         N, S, noises=[DeviceNoise(dev, seed=500 + c) for c in range(3)], sync_every=4, **kw)
        for a, b in zip(singles, multi):
            assert a[0].shape == b[0].shape and a[2] == b[2]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            for f in ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin",
                      "energies_pot_delta", "energies_kin_delta"):
                assert np.array_equal(getattr(a[3], f), getattr(b[3], f)), f


def test_multichain_full_size_split_fp16():
    """Same equivalence on the product configuration: full-size flow on the split-fp16 kernels, AMBER energy kernel,
    alanine dipeptide, two chains x 16 proposals."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.evaluation_utils import DeviceNoise, sample_with_model
    from timewarp_amd.utils.multichain import sample_with_model_chains

    sd = synthetic.synth_state_dict(H.full_kernel_sd(), 0, calibrated=True, coords_log_scale=-7.0, velocs_log_scale=0.0)
    model = H.tw_kernel_model(sd, path=3)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(1)
    starts = [coords + 0.002 * torch.randn(coords.shape, generator=g) for _ in range(2)]
    kw = dict(random_velocs=True, resample_velocs=True)
    singles = [sample_with_model(single_state_batch("ad", types, xc), model, dev, energy, masses, 12, accept=True,
                                 num_proposal_steps=16, disable_tqdm=True, noise=DeviceNoise(dev, seed=40 + c), **kw)
               for c, xc in enumerate(starts)]
    multi = sample_with_model_chains([single_state_batch("ad", types, xc) for xc in starts], model, dev, energy, masses, 12, 16,
                                     noises=[DeviceNoise(dev, seed=40 + c) for c in range(2)], sync_every=2, **kw)
    for a, b in zip(singles, multi):
        assert a[0].shape == b[0].shape and a[2] == b[2]
        assert np.allclose(a[0], b[0], rtol=0, atol=1e-6) and np.allclose(a[3].exponent, b[3].exponent, rtol=1e-5, atol=1e-4)
