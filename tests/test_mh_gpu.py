"""The product MH loop (HIP) replayed on the reference's recorded traces, plus the energy kernel
against the C oracle."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from oracle import mh_oracle as mo
from tests import helpers as H
from tests.test_mh_oracle import OPENMM_SCENARIOS, SCENARIOS, check_against_golden, load_mh, replay

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


@pytest.mark.parametrize("name", list(OPENMM_SCENARIOS))
def test_sample_with_model_replays_reference_with_openmm_steps(name):
    """OpenMM steps on the current state / on the proposal (evaluation_utils.py:558-565, 594-602, 623-626): the runs were
    recorded from the reference with oracle/fake_sim.FakeSimulation in the Simulation's place; the product drives the
    same object through the same five calls."""
    from oracle.fake_sim import FakeSimulation
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.evaluation_utils import sample_with_model

    z, sd = load_mh("mh_tiny_openmm.npz")
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    batch = single_state_batch("tiny", torch.from_numpy(z["atom_types"]), x0, v0)
    sim = FakeSimulation()
    coords, velocs, accepted, stats = sample_with_model(
        batch, model, torch.device("cuda"), mo.SyntheticEnergy(x0.clone().cuda()), torch.from_numpy(z["masses"]),
        disable_tqdm=True, noise=replay(z, name, device="cuda"), sim=sim, **OPENMM_SCENARIOS[name])
    check_against_golden(z, name, coords, velocs, accepted, stats)
    assert sim.calls > 0


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


MH_FULL_CASES = [("bench", True, 0, 40), ("scaled", True, 1, 130), ("scaled", False, 1, 130)]


@pytest.mark.parametrize("path", [1, 3])
@pytest.mark.parametrize("kind,random_velocs,seed,num_samples", MH_FULL_CASES)
def test_full_size_mh_iterations_vs_oracle(path, kind, random_velocs, seed, num_samples):
    """Whole MH iterations at the bench configuration - full-size kernel_transformer_nvp flow (exact-f32 and
    split-fp16 fused kernels), AMBER energy kernel, alanine dipeptide, 64 proposals per iteration - against
    oracle/mh_oracle.sample_with_model + oracle/energy_oracle.c on the same host-drawn noise: emitted states and
    velocities, accept count, accept indicators (= first-accepted indices), and all eight per-state statistics
    (reference evaluation_utils.py:609-713, flow.py:242-336)."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.evaluation_utils import sample_with_model

    S = 64
    sd = H.mh_state_dict(kind, random_velocs)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    v0 = torch.randn(1, 22, 3, generator=torch.Generator().manual_seed(9)) * 0.05
    kw = dict(accept=True, num_proposal_steps=S)
    if random_velocs:
        kw.update(random_velocs=True, resample_velocs=True)
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    ref = mo.sample_with_model(types[None], coords[None], v0, torch.zeros(1, 22, dtype=torch.bool),
                               mo.OracleModel(sd, H.FULL_KERNEL_SPEC), H.OracleAmberEnergy(energy.tables), masses,
                               num_samples, H.HostNoise(seed), **kw)
    model = H.tw_kernel_model(sd, path=path)
    got = sample_with_model(single_state_batch("ad", types, coords, v0[0]), model, torch.device("cuda"), energy, masses,
                            num_samples, disable_tqdm=True, noise=H.HostNoise(seed, "cuda"), **kw)
    H.assert_not_demoted(model)
    (rc, rv, racc, rs), (gc, gv, gacc, gs) = ref, got
    assert racc >= 1 and (racc >= 2 or rc.shape[0] > S + 1)  # accepted proposals, and more than one iteration
    assert gc.shape == rc.shape and gacc == racc
    assert np.array_equal(gs.acceptance_indicator.astype(bool), rs.acceptance_indicator.astype(bool))
    assert H.rel_err(gc, rc) < 1e-5 and H.rel_err(gv, rv) < 1e-5
    assert H.rel_err(gs.p_xy, rs.p_xy) < 1e-5 and H.rel_err(gs.p_yx, rs.p_yx) < 1e-5
    assert H.elem_rel_err(gs.p_xy, rs.p_xy) < 1e-5 and H.elem_rel_err(gs.p_yx, rs.p_yx) < 1e-5
    assert H.rel_err(gs.energies_pot, rs.energies_pot) < 1e-5 and H.rel_err(gs.energies_kin, rs.energies_kin) < 1e-5
    # differences of O(100) quantities: absolute bars at 1e-5 of the magnitude of what is subtracted
    scale = float(np.abs(rs.p_xy).max() + np.abs(rs.energies_pot).max() + np.abs(rs.energies_kin).max())
    for f in ("exponent", "energies_pot_delta", "energies_kin_delta"):
        assert np.abs(getattr(gs, f) - getattr(rs, f)).max() < 1e-5 * scale, f
    assert np.abs(gs.acceptance - rs.acceptance).max() < 2e-5 * scale  # p_acc = min(1, e^-exponent)


@pytest.mark.parametrize("path", [1, 3])
@pytest.mark.parametrize("init_random,smoothing,seed", [(False, 0.3, 4), (True, 0.1, 5)])
def test_adaptive_parallelism_and_random_init_through_mh_iteration(path, init_random, smoothing, seed):
    """`adaptive_parallelism=True` (the proposal count S follows the smoothed acceptance rate, reference
    evaluation_utils.py:575-584, 685-697) and `initialize_randomly=True` (:540-553) on the product route: full-size flow
    on both fused kernels, AMBER energy kernel, tw_mh_iteration - whose workspace and per-S constants are re-sized
    whenever S changes - against oracle/mh_oracle.sample_with_model on shared host noise.  S has to change at least
    twice during the run.  From the data sample (scaled weights: accepts and rejections alternate) and from a random
    flow sample (a 22-atom cloud far from any minimum: downhill proposals are accepted at once, S falls every iteration)."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils import evaluation_utils as eu

    S_max, N = 64, 60
    sd = H.mh_state_dict("scaled", True)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    kw = dict(accept=True, num_proposal_steps=S_max, random_velocs=True, resample_velocs=True, adaptive_parallelism=True,
              acceptance_rate_smoothing_factor=smoothing, initialize_randomly=init_random)
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    ref = mo.sample_with_model(types[None], coords[None], torch.zeros(1, 22, 3), torch.zeros(1, 22, dtype=torch.bool),
                               mo.OracleModel(sd, H.FULL_KERNEL_SPEC), H.OracleAmberEnergy(energy.tables), masses, N,
                               H.HostNoise(seed), **kw)
    model = H.tw_kernel_model(sd, path=path)
    sizes = []
    real = eu.MetropolisHastingsChain._iteration_fused

    def spy(self):
        sizes.append(self.S)
        return real(self)

    eu.MetropolisHastingsChain._iteration_fused = spy
    try:
        got = eu.sample_with_model(single_state_batch("ad", types, coords), model, torch.device("cuda"), energy, masses, N,
                                   disable_tqdm=True, noise=H.HostNoise(seed, "cuda"), **kw)
    finally:
        eu.MetropolisHastingsChain._iteration_fused = real
    H.assert_not_demoted(model)
    changes = sum(1 for a, b in zip(sizes, sizes[1:]) if a != b)
    assert len(sizes) >= 3 and changes >= 2, sizes  # every iteration went through tw_mh_iteration; S changed
    assert ref[2] >= 2
    if init_random:  # the chain starts from the flow sample, not from the data sample
        assert np.abs(got[0][0] - coords.numpy()).max() > 0.1
    # random cloud: E_pot ~ 1.2e7 kJ/mol, where the energy callable's float32 output has a quantum of 1 kJ/mol = 0.4 kT
    # against energy differences of a few thousand kT - a proposal whose last coordinate bit differs shows up as 1e-4
    _assert_chain_matches_oracle(got, ref, tol=1e-5, stat_tol=5e-4 if init_random else 2e-4)


@pytest.mark.parametrize("mode", ["sync", "deferred", "multichain", "multichain-lag"])
def test_split_fp16_overflow_is_redone_on_the_f32_kernels(mode):
    """A checkpoint whose activations leave the fp16 range must not abort a chain hours in: when the range flag is up
    at a read-back, the model is demoted to the exact-f32 kernels and the iterations since the last read-back are
    replayed there from the recorded draws.  The resulting chain is bit for bit the chain the f32 kernels produce from
    the start with the same noise - one synchronous iteration at a time, with deferred read-backs (4 iterations parked
    when the flag is seen), for lock-step chains, and for lock-step chains on the kernel's own draws with the lagged
    read-back (r06: the flag is seen one window late, two windows are redone from the same counters)."""
    from tests.test_flow_gpu import _overflowing_sd
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.evaluation_utils import DeviceNoise, sample_with_model
    from timewarp_amd.utils.multichain import sample_with_model_chains

    sd = _overflowing_sd()
    for k in sd:
        if ".out_mlp._layers.2." in k:
            sd[k] = sd[k] * 1e-4
    sd["coords_prior_log_scale"] = torch.tensor(-7.0)
    sd["velocs_prior_log_scale"] = torch.tensor(0.0)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    dev = torch.device("cuda")
    kw = dict(random_velocs=True, resample_velocs=True)
    S, N = 16, 60

    def run(path):
        model = H.tw_kernel_model(sd, path=path)
        if mode.startswith("multichain"):
            g = torch.Generator().manual_seed(1)
            starts = [coords + 0.0005 * torch.randn(coords.shape, generator=g) for _ in range(2)]
            torch.cuda.manual_seed(321)   # (the initial velocities of the kernel-draws chains come from the default generator)
            draws = dict(seed=5150) if mode == "multichain-lag" else dict(noises=[DeviceNoise(dev, seed=40 + c) for c in range(2)])
            out = sample_with_model_chains([single_state_batch("ad", types, xc) for xc in starts], model, dev, energy, masses,
                                           N, S, sync_every=3, **draws, **kw)
        else:
            noise = H.HostNoise(8, "cuda") if mode == "sync" else None  # host noise forces one read-back per iteration
            torch.cuda.manual_seed(123)
            out = [sample_with_model(single_state_batch("ad", types, coords), model, dev, energy, masses, 5 * S + 3, accept=True,
                                     num_proposal_steps=S, disable_tqdm=True, noise=noise, sync_every=4, **kw)]
        return out, model

    ref, m32 = run(1)
    assert not m32.demoted
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        got, m = run(3)
    assert m.demoted
    for a, b in zip(got, ref):
        assert a[0].shape == b[0].shape and a[2] == b[2]
        assert np.isfinite(a[3].exponent).all()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for f in ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin",
                  "energies_pot_delta", "energies_kin_delta"):
            assert np.array_equal(getattr(a[3], f), getattr(b[3], f)), f


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


def test_accept_kernel_rejects_non_finite_exponents():
    """torch.min(1, exp(-exp)) propagates NaN and `rand < NaN` is False (reference evaluation_utils.py:665-668): a
    proposal whose exponent is NaN (or +inf) is rejected and the chain state stays put; -inf is accepted (p = 1).
    Checked for tw_mh_accept and tw_mh_accept_chains, with the state update done on the device."""
    from timewarp_amd import _lib
    from timewarp_amd.utils.evaluation_utils import _mh_accept

    S, V = 64, 5
    nan, inf = float("nan"), float("inf")
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    yc, yv = torch.randn(S, V, 3, generator=g), torch.randn(S, V, 3, generator=g)
    x0c, x0v = torch.randn(1, V, 3, generator=g), torch.randn(1, V, 3, generator=g)
    u = torch.rand(S, generator=g) * 0.5
    big = torch.full((S,), 1e4)  # exp(-1e4) = 0: rejected
    zero = torch.zeros(S)
    for field in range(3):  # the non-finite value arrives through the energy, p_xy or p_yx
        for bad in (nan, inf if field != 2 else -inf):
            args = [big.clone(), zero.clone(), zero.clone()]
            args[field][:] = bad
            xc, xv = x0c.clone().to(dev), x0v.clone().to(dev)
            ex, p_acc, acc, res = _mh_accept(args[0].to(dev), args[1].to(dev), args[2].to(dev), u.to(dev), yc.to(dev), yv.to(dev), xc, xv)
            e_ref = args[0] + args[1] - args[2]
            p_ref = torch.min(torch.tensor(1.0), torch.exp(-e_ref))
            assert (acc.cpu() == 0).all() and res[:2].tolist() == [S - 1, 0], (field, bad)
            assert torch.equal(torch.isnan(p_acc.cpu()), torch.isnan(p_ref))
            assert torch.equal(xc.cpu(), x0c) and torch.equal(xv.cpu(), x0v)
    # a NaN row in front of a genuinely accepted one: the first accepted index skips it
    e = big.clone()
    e[3], e[7] = nan, -inf
    xc, xv = x0c.clone().to(dev), x0v.clone().to(dev)
    ex, p_acc, acc, res = _mh_accept(e.to(dev), zero.to(dev), zero.to(dev), u.to(dev), yc.to(dev), yv.to(dev), xc, xv)
    assert res[:2].tolist() == [7, 1] and acc.cpu().nonzero().flatten().tolist() == [7]
    assert torch.equal(xc.cpu()[0], yc[7]) and torch.equal(xv.cpu()[0], yv[7])
    # two chains in one call (row = proposal * n_chains + chain): chain 0 all-NaN, chain 1 accepts proposal 5
    C_ = 2
    e2 = torch.full((S, C_), 1e4)
    e2[:, 0] = nan
    e2[5, 1] = -1.0
    yc2, yv2 = torch.randn(S, C_, V, 3, generator=g), torch.randn(S, C_, V, 3, generator=g)
    x2c, x2v = torch.randn(C_, V, 3, generator=g), torch.randn(C_, V, 3, generator=g)
    xc, xv = x2c.clone().to(dev), x2v.clone().to(dev)
    outs = [torch.empty(S * C_, dtype=dt, device=dev) for dt in (torch.float32, torch.float32, torch.uint8)]
    res = torch.empty(C_, 4, dtype=torch.int32, device=dev)
    z2 = torch.zeros(S * C_, device=dev)
    u2 = torch.full((S * C_,), 0.25, device=dev)
    lib = _lib.load()
    e2d, yc2d, yv2d = e2.reshape(-1).to(dev), yc2.to(dev), yv2.to(dev)  # named: the kernel reads them after this line
    _lib.check(lib.tw_mh_accept_chains(e2d.data_ptr(), z2.data_ptr(), z2.data_ptr(), u2.data_ptr(),
                                       yc2d.data_ptr(), yv2d.data_ptr(), xc.data_ptr(), xv.data_ptr(),
                                       outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), res.data_ptr(), S, C_, V,
                                       _lib.stream_ptr(dev)), "tw_mh_accept_chains")
    assert res.cpu()[:, :2].tolist() == [[S - 1, 0], [5, 1]]
    assert torch.equal(xc.cpu()[0], x2c[0]) and torch.equal(xc.cpu()[1], yc2[5, 1]) and torch.equal(xv.cpu()[1], yv2[5, 1])


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


@pytest.mark.parametrize("tag,random_velocs", [("fixedv", False), ("randv", True)])
def test_sample_on_single_conditional_replays_reference(tag, random_velocs):
    """Product sample_on_single_conditional (reference evaluation_utils.py:356-413) against vectors recorded from the
    reference's own function, oracle/fake_sim.FakeSimulation standing in for the OpenMM Simulation on both sides."""
    from oracle.fake_sim import FakeSimulation
    from tests.test_mh_oracle import check_sosc, load_sosc, sob_replay
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.evaluation_utils import sample_on_single_conditional

    z, sd = load_sosc()
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    batch = single_state_batch("tiny", torch.from_numpy(z["atom_types"]), torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"]))
    sim = FakeSimulation(allow_thermal=True)
    res = sample_on_single_conditional(batch, model, int(z["num_samples"]), sim, int(z["step_width"]), random_velocs,
                                       torch.device("cuda"), noise=sob_replay(z, tag, device="cuda"))
    check_sosc(z, tag, res)
    assert sim.calls == int(z["num_samples"])
    # without injected noise: device draws, same shapes
    res2 = sample_on_single_conditional(batch, model, 2, FakeSimulation(allow_thermal=True), 1, random_velocs, torch.device("cuda"))
    assert res2[0].shape == (2, 7, 3) and res2[2].shape == (2, 7, 3) and np.isfinite(res2[0]).all()


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


@pytest.mark.parametrize("scenario", ["omm_current", "omm_proposal"])
def test_sample_trajectory_forwards_the_hybrid_move_arguments(tmp_path, scenario):
    """sample_trajectory hands `sim`, `openmm_on_current`, `openmm_on_proposal`, `num_openmm_steps` to the sampler as the
    reference's script does (sample_trajectory.py:218, 246-266; r05 dropped them): a segment written with a fake Simulation
    equals the chain `sample_with_model` produces with the same options and device seed, the Simulation is driven, and it is
    NOT handed on when neither switch is set (`sim=simulation if needs_sim else None`)."""
    from oracle.fake_sim import FakeSimulation
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.sample_trajectory import sample_trajectory, segment_path
    from timewarp_amd.utils.evaluation_utils import sample_with_model

    z, sd = load_mh("mh_tiny_openmm.npz")
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    mk = lambda: single_state_batch("tiny", torch.from_numpy(z["atom_types"]), x0, v0)
    energy = mo.SyntheticEnergy(x0.clone().cuda())
    masses = torch.from_numpy(z["masses"])
    opts = {k: v for k, v in OPENMM_SCENARIOS[scenario].items() if k in ("openmm_on_current", "openmm_on_proposal", "num_openmm_steps")}
    assert opts.get("num_openmm_steps", 0) > 0 and (opts.get("openmm_on_current") or opts.get("openmm_on_proposal"))
    dev = torch.device("cuda")
    sim = FakeSimulation()
    torch.cuda.manual_seed(11)
    out = str(tmp_path / "hyb")
    S = OPENMM_SCENARIOS[scenario]["num_proposal_steps"]   # (openmm_on_proposal: one proposal per iteration, as in the reference)
    assert sample_trajectory(mk(), model, dev, energy, masses, out, "tiny", 30, 30, mh=True, num_proposal_steps=S, verbose=False,
                             sim=sim, **opts) == 1
    assert sim.calls > 0
    ref_sim = FakeSimulation()
    torch.cuda.manual_seed(11)
    want, _, _, _ = sample_with_model(mk(), model, dev, energy, masses, 30, True, num_proposal_steps=S, disable_tqdm=True,
                                      sim=ref_sim, **opts)
    got = np.load(segment_path(out, "tiny", 0))["positions"]
    assert ref_sim.calls == sim.calls and np.array_equal(got, want[::10])
    # neither switch set: the Simulation stays untouched and the chain is the plain one
    idle = FakeSimulation()
    torch.cuda.manual_seed(11)
    out2 = str(tmp_path / "plain")
    sample_trajectory(mk(), model, dev, energy, masses, out2, "tiny", 30, 30, mh=True, num_proposal_steps=S, verbose=False,
                      sim=idle, num_openmm_steps=opts["num_openmm_steps"])
    torch.cuda.manual_seed(11)
    plain, _, _, _ = sample_with_model(mk(), model, dev, energy, masses, 30, True, num_proposal_steps=S, disable_tqdm=True)
    assert idle.calls == 0 and np.array_equal(np.load(segment_path(out2, "tiny", 0))["positions"], plain[::10])
    assert not np.array_equal(got, plain[::10])


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


def _assert_chain_matches_oracle(got, ref, tol=2e-5, stat_tol=1e-4):
    (gc, gv, gacc, gs), (rc, rv, racc, rs) = got, ref
    assert gc.shape == rc.shape and gacc == racc
    assert np.array_equal(np.asarray(gs.acceptance_indicator).astype(bool), np.asarray(rs.acceptance_indicator).astype(bool))
    assert H.rel_err(gc, rc) < tol and H.rel_err(gv, rv) < tol
    for f in ("acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin", "energies_pot_delta", "energies_kin_delta"):
        a, b = np.asarray(getattr(gs, f), np.float64), np.asarray(getattr(rs, f), np.float64)
        assert a.shape == b.shape and H.rel_err(a, b) < stat_tol, (f, H.rel_err(a, b))


def test_multichain_vs_oracle_per_chain():
    """Lock-step chains against the ORACLE (not against the product's own single chains): every chain of
    sample_with_model_chains must equal oracle/mh_oracle.sample_with_model run on that chain alone with the same
    per-chain host noise - states, velocities, accept counts, indicators and statistics, including the final clip."""
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.multichain import sample_with_model_chains

    z, sd = load_mh()
    model = H.tw_kernel_model(sd, emb=4, d_model=8, ff=16, hidden=8, n_coupling=2, n_layers=2,
                              lengthscales=(0.1, 0.5, 1.2), path=2)
    at = torch.from_numpy(z["atom_types"])
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    g = torch.Generator().manual_seed(31)
    starts = [(x0 + 0.03 * torch.randn(x0.shape, generator=g), v0 + 0.1 * torch.randn(v0.shape, generator=g)) for _ in range(3)]
    masses = torch.from_numpy(z["masses"])
    mask = torch.zeros(1, x0.shape[1], dtype=torch.bool)
    dev = torch.device("cuda")
    N, S = 70, 10
    oracle_model = mo.OracleModel(sd, H.TINY_KERNEL_SPEC)
    for kw in (dict(), dict(random_velocs=True, resample_velocs=True)):
        refs = [mo.sample_with_model(at, xc, vc, mask, oracle_model, mo.SyntheticEnergy(x0.clone()), masses, N,
                                     H.HostNoise(700 + c), accept=True, num_proposal_steps=S, **kw)
                for c, (xc, vc) in enumerate(starts)]
        multi = sample_with_model_chains([single_state_batch("t", at, xc, vc) for xc, vc in starts], model, dev,
                                         mo.SyntheticEnergy(x0.clone().cuda()), masses, N, S,
                                         noises=[H.HostNoise(700 + c, "cuda") for c in range(3)], sync_every=4, **kw)
        assert sum(r[2] for r in refs) > 0
        for got, ref in zip(multi, refs):
            _assert_chain_matches_oracle(got, ref)


def test_multichain_full_size_vs_oracle():
    """The same on the product configuration: full-size flow (split-fp16 kernels), AMBER energy kernel, two chains x
    32 proposals, weights whose coupling nets matter ("scaled", tests/helpers.py), each chain against the oracle."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.multichain import sample_with_model_chains

    sd = H.mh_state_dict("scaled", True)
    model = H.tw_kernel_model(sd, path=3)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    g = torch.Generator().manual_seed(1)
    starts = [coords + 0.0005 * torch.randn(coords.shape, generator=g) for _ in range(2)]
    kw = dict(random_velocs=True, resample_velocs=True)
    N, S = 70, 32
    mask = torch.zeros(1, 22, dtype=torch.bool)
    refs = [mo.sample_with_model(types[None], xc[None], torch.zeros(1, 22, 3), mask, mo.OracleModel(sd, H.FULL_KERNEL_SPEC),
                                 H.OracleAmberEnergy(energy.tables), masses, N, H.HostNoise(40 + c), accept=True,
                                 num_proposal_steps=S, **kw) for c, xc in enumerate(starts)]
    multi = sample_with_model_chains([single_state_batch("ad", types, xc) for xc in starts], model, torch.device("cuda"),
                                     energy, masses, N, S, noises=[H.HostNoise(40 + c, "cuda") for c in range(2)],
                                     sync_every=2, **kw)
    H.assert_not_demoted(model)
    for got, ref in zip(multi, refs):
        _assert_chain_matches_oracle(got, ref, tol=1e-5, stat_tol=2e-4)


@pytest.mark.parametrize("path", [1, 3])
def test_sample_drivers_with_hip_flow_vs_oracle(path):
    """`sample` / `sample_from_trajectory` (reference utils/sampling_utils.py:17-181, the sample.py drivers) running the
    HIP flow: S x conditional_sample(num_samples=1) with the latents drawn on the device in the reference's order
    (coords, then velocities, flow.py:274-275).  Re-seeding the device generator reproduces those draws, which the
    oracle then turns into the expected samples."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.sampling_utils import sample, sample_from_trajectory

    sd = H.full_kernel_sd()
    model = H.tw_kernel_model(sd, path=path)
    types, coords, _ = synthetic.alanine_dipeptide_state()
    g = torch.Generator().manual_seed(3)
    dev = torch.device("cuda")
    batches = [single_state_batch("ad", types, coords + 0.01 * torch.randn(coords.shape, generator=g),
                                  0.5 * torch.randn(22, 3, generator=g)) for _ in range(2)]
    S = 5
    sc, sv = torch.exp(sd["coords_prior_log_scale"]), torch.exp(sd["velocs_prior_log_scale"])
    mask = torch.zeros(1, 22, dtype=torch.bool)

    def expected(seed, bs):
        torch.cuda.manual_seed(seed)
        out = []
        for b in bs:
            zc, zv = [], []
            for _ in range(S):
                zc.append((torch.randn((1, 1, 22, 3), device=dev) * sc.to(dev)).cpu())
                zv.append((torch.randn((1, 1, 22, 3), device=dev) * sv.to(dev)).cpu())
            yc, yv, _ = fo.conditional_sample_with_logp(sd, H.FULL_KERNEL_SPEC, b.atom_types, b.atom_coords, b.atom_velocs, mask,
                                                        torch.cat(zc), torch.cat(zv))
            out.append((yc.squeeze(1).numpy(), yv.squeeze(1).numpy()))
        return out

    torch.cuda.manual_seed(77)
    c, v = sample(model, batches[0], S, device=dev)
    assert c.shape == (S, 22, 3) and c.dtype == np.float64 and v.dtype == np.float64  # sampling_utils.py:117-141
    (ec, ev), = expected(77, batches[:1])
    assert H.rel_err(c, ec) < 1e-5 and H.rel_err(v, ev) < 1e-5
    torch.cuda.manual_seed(78)
    cs, vs = sample_from_trajectory(model, batches, S, device=dev)
    exp = expected(78, batches)
    assert len(cs) == len(vs) == 2
    for (ec, ev), c, v in zip(exp, cs, vs):
        assert H.rel_err(c, ec) < 1e-5 and H.rel_err(v, ev) < 1e-5


@pytest.mark.parametrize("random_velocs,chirality", [(True, False), (False, True)])
def test_fused_iteration_equals_op_by_op_route(monkeypatch, random_velocs, chirality):
    """tw_mh_iteration (one C-ABI call per MH iteration) against the op-by-op route (flow sample, energies, kinetic
    energies, chirality guard, reverse-move likelihood, accept kernel + elementwise steps) on the same device noise:
    bit-identical chains, velocities and statistics - synchronous and deferred read-back."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.chirality import compute_chirality_sign, find_chirality_centers
    from timewarp_amd.utils.evaluation_utils import DeviceNoise, MetropolisHastingsChain, sample_with_model

    sd = H.mh_state_dict("scaled", random_velocs)
    model = H.tw_kernel_model(sd, path=3)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    dev = torch.device("cuda")
    v0 = torch.randn(1, 22, 3, generator=torch.Generator().manual_seed(9)) * 0.05
    kw = dict(accept=True, num_proposal_steps=48)
    if random_velocs:
        kw.update(random_velocs=True, resample_velocs=True)
    if chirality:
        centres = torch.tensor([[8, 6, 10, 14]])  # CA with N, CB, C (alanine dipeptide atom order)
        kw.update(chirality_centers=centres, reference_signs=compute_chirality_sign(coords[None], centres))
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("TW_MH_FUSED", fused)
        batch = single_state_batch("ad", types, coords, v0[0])
        chain = MetropolisHastingsChain(batch, model, dev, energy, masses, noise=DeviceNoise(dev, seed=3), **kw)
        assert chain._fused == (fused == "1")
        with torch.no_grad():
            for it in range(4):
                chain.step_deferred()
            chain.flush()
            chain.step(5)  # synchronous iteration with the reference's clip
        outs[fused] = chain.result()
    a, b = outs["1"], outs["0"]
    assert a[2] == b[2] and a[2] >= 1 and a[0].shape == b[0].shape
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for f in ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin",
              "energies_pot_delta", "energies_kin_delta"):
        assert np.array_equal(getattr(a[3], f), getattr(b[3], f)), f


@pytest.mark.parametrize("path", [0, 3])
def test_mh_iterations_on_a_65_atom_peptide_vs_oracle(path):
    """The tetrapeptide NNQQ of the reference's own OpenMM test data (65 atoms, amber99sb-ildn + GBSA-OBC tables pinned by
    that file, tests/test_energy_kat.py): whole MH iterations with the full-size flow - above 64 atoms TW_PATH_AUTO runs
    the per-op kernels, the split-fp16 kernel its wide layout with two molecules per workgroup (scores from torch.cdist's
    matmul branch either way) - through tw_mh_iteration, against the oracle loop and the C energy oracle on shared host
    noise.  The BASELINE tetrapeptide configuration's MH half on a real molecule."""
    from tests.test_energy_kat import kat, kat_tables
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.forcefield import ELEMENT_MASSES
    from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain, sample_with_model

    z = kat()
    tables = kat_tables(z)
    vocab = {"C": 0, "H": 1, "N": 2, "O": 3, "S": 4}
    els = [str(e) for e in z["elements"]]
    types = torch.tensor([vocab[e] for e in els])
    masses = torch.tensor([ELEMENT_MASSES[e] for e in els], dtype=torch.float32)
    coords = torch.from_numpy(z["positions"][0]).float()
    sd = H.mh_state_dict("scaled", True, out_scale=3e-5, coords_log_scale=-7.5)  # 65 atoms: smaller moves for the same acceptance
    energy = AmberPotentialEnergyTorch(tables)
    S, N = 16, 40
    kw = dict(accept=True, num_proposal_steps=S, random_velocs=True, resample_velocs=True)
    ref = mo.sample_with_model(types[None], coords[None], torch.zeros(1, 65, 3), torch.zeros(1, 65, dtype=torch.bool),
                               mo.OracleModel(sd, H.FULL_KERNEL_SPEC), H.OracleAmberEnergy(tables), masses, N,
                               H.HostNoise(2), **kw)
    model = H.tw_kernel_model(sd, path=path)
    dev = torch.device("cuda")
    chain = MetropolisHastingsChain(single_state_batch("nnqq", types, coords), model, dev, energy, masses,
                                    noise=H.HostNoise(2, "cuda"), **kw)
    assert chain._fused and model._path_for(65) == path
    got = sample_with_model(single_state_batch("nnqq", types, coords), model, dev, energy, masses, N, disable_tqdm=True,
                            noise=H.HostNoise(2, "cuda"), **kw)
    assert ref[2] >= 1
    H.assert_not_demoted(model)
    _assert_chain_matches_oracle(got, ref, tol=1e-5, stat_tol=2e-4)


def test_mh_iterations_on_a_61_atom_peptide_in_64_token_waves_vs_oracle():
    """bench.py --config 4aa's molecule (NAQQ: the reference's OpenMM test peptide with one asparagine cut back to alanine, 61
    atoms) through tw_mh_iteration on the 64-token build of the split-fp16 kernel (forced: 16 proposals are one round of
    workgroups in either layout, where the launch code would take the wide one), against the oracle loop and the C energy
    oracle on shared host noise."""
    import bench
    from timewarp_amd import _lib
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain, sample_with_model

    name, types, coords, masses, energy = bench.molecule("4aa")
    assert name == "NAQQ" and len(types) == 61
    sd = H.mh_state_dict("scaled", True, out_scale=3e-5, coords_log_scale=-7.5)
    S, N = 16, 40
    kw = dict(accept=True, num_proposal_steps=S, random_velocs=True, resample_velocs=True)
    ref = mo.sample_with_model(types[None], coords[None], torch.zeros(1, 61, 3), torch.zeros(1, 61, dtype=torch.bool),
                               mo.OracleModel(sd, H.FULL_KERNEL_SPEC), H.OracleAmberEnergy(energy.tables), masses, N,
                               H.HostNoise(2), **kw)
    lib = _lib.load()
    dev = torch.device("cuda")
    try:
        lib.tw_debug_set_flags(65536)
        model = H.tw_kernel_model(sd, path=3)
        chain = MetropolisHastingsChain(single_state_batch("naqq", types, coords), model, dev, energy, masses,
                                        noise=H.HostNoise(2, "cuda"), **kw)
        assert chain._fused
        got = sample_with_model(single_state_batch("naqq", types, coords), model, dev, energy, masses, N, disable_tqdm=True,
                                noise=H.HostNoise(2, "cuda"), **kw)
    finally:
        lib.tw_debug_set_flags(0)
    assert ref[2] >= 1
    H.assert_not_demoted(model)
    _assert_chain_matches_oracle(got, ref, tol=1e-5, stat_tol=2e-4)


@pytest.mark.parametrize("path", [1, 3])
def test_dense_flow_mh_iterations_vs_oracle(path):
    """BASELINE config 4 as a sampler: whole MH iterations with the full-size dense-softmax flow (transformer_nvp) on the
    fused f32 and the split-fp16 dense kernels through tw_mh_iteration, AMBER energy kernel, alanine dipeptide, 64
    proposals, against the oracle loop on shared host noise (coupling nets scaled so that proposals are accepted)."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain, sample_with_model

    sd = dict(H.full_dense_sd())
    for k in sd:
        if ".out_mlp._layers.2." in k:
            sd[k] = sd[k] * 1e-4
    sd["coords_prior_log_scale"] = torch.tensor(-7.0)
    sd["velocs_prior_log_scale"] = torch.tensor(0.0)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    S, N = 64, 130
    kw = dict(accept=True, num_proposal_steps=S, random_velocs=True, resample_velocs=True)
    ref = mo.sample_with_model(types[None], coords[None], torch.zeros(1, 22, 3), torch.zeros(1, 22, dtype=torch.bool),
                               mo.OracleModel(sd, H.FULL_DENSE_SPEC), H.OracleAmberEnergy(energy.tables), masses, N,
                               H.HostNoise(6), **kw)
    model = H.tw_dense_model(sd, path=path)
    dev = torch.device("cuda")
    chain = MetropolisHastingsChain(single_state_batch("ad", types, coords), model, dev, energy, masses,
                                    noise=H.HostNoise(6, "cuda"), **kw)
    assert chain._fused
    got = sample_with_model(single_state_batch("ad", types, coords), model, dev, energy, masses, N, disable_tqdm=True,
                            noise=H.HostNoise(6, "cuda"), **kw)
    assert ref[2] >= 1
    H.assert_not_demoted(model)
    _assert_chain_matches_oracle(got, ref, tol=1e-5, stat_tol=2e-4)


@pytest.mark.parametrize("path", [1, 3])
def test_exploration_mode_vs_oracle(path):
    """The exploration loop (timewarp_amd/exploration.py after the reference's exploration.py:229-257: P parallel explorers,
    no MH correction, energy threshold, chirality guard) on the HIP flow / energy / chirality kernels against the oracle
    loop with the same host-drawn noise: the same stay / move decisions, positions and energies at 1e-5.  Weights whose
    nets act and a coordinate prior (e^-6 nm) for which, with a 60 kJ/mol threshold, both outcomes occur (23 moves of 40 after
    the first step)."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.exploration import explore
    from timewarp_amd.forcefield import alanine_dipeptide_amber99sb

    sd = H.mh_state_dict("scaled", True, coords_log_scale=-6.0)
    model = H.tw_kernel_model(sd, path=path)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    tables = alanine_dipeptide_amber99sb()
    adj = torch.from_numpy(tables.bond_idx.astype(np.int64))
    v0 = torch.randn(22, 3, generator=torch.Generator().manual_seed(4))
    batch = single_state_batch("ad", types, coords, v0, adj_list=adj)
    P, steps, thr = 8, 6, 60.0
    pos, en = explore(batch, model, "cuda", AmberPotentialEnergyTorch(tables), steps, P, thr, noise=H.HostNoise(11, "cuda"))
    from timewarp_amd.utils.chirality import find_chirality_centers

    centres = find_chirality_centers(adj, types[None])
    assert centres.tolist() == [[8, 6, 9, 14]]   # the alpha carbon with N, HA, C
    om = mo.OracleModel(sd, H.FULL_KERNEL_SPEC)
    rpos, ren = mo.explore(types[None], coords[None], v0[None], torch.zeros(1, 22, dtype=torch.bool), om,
                           H.OracleAmberEnergy(tables), centres, steps, P, thr, H.HostNoise(11))
    pos, en = pos.cpu(), en.cpu()
    assert pos.shape == (steps * P, 22, 3) and en.shape == (steps * P, 1)
    moved = (rpos[P:] != rpos[:-P]).any(-1).any(-1)
    assert 0 < int(moved.sum()) < moved.numel()            # both outcomes occur after the first step
    assert torch.equal((pos[P:] != pos[:-P]).any(-1).any(-1), moved)
    assert H.rel_err(pos, rpos) < 1e-5
    assert float((en - ren).abs().max()) < 2e-3            # kJ/mol, fp32 energies of ~ -50 .. +100


@pytest.mark.parametrize("window", [64, 2])
def test_exploration_mode_range_guard_replays_on_f32(window, monkeypatch):
    """Exploration with a checkpoint whose activations leave the fp16 range: one look at the range flag per WINDOW of steps (no
    synchronisation per model call; r05: windows of 64 steps instead of the whole run, so that only one window's draws are
    kept), then that window again on the f32 kernels with the recorded draws and the rest there too - bit for bit what the f32
    path gives from the start with the same device noise.  With a window of 2 the five steps are three windows: the replay
    starts from the window's own starting state and the later windows run unguarded."""
    from timewarp_amd import exploration as ex_mod

    monkeypatch.setattr(ex_mod, "RANGE_CHECK_WINDOW", window)
    from tests.test_flow_gpu import _overflowing_sd
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.exploration import explore
    from timewarp_amd.forcefield import alanine_dipeptide_amber99sb
    from timewarp_amd.utils.evaluation_utils import DeviceNoise

    sd = _overflowing_sd()
    for k in sd:
        if ".out_mlp._layers.2." in k:
            sd[k] = sd[k] * 1e-4
    sd["coords_prior_log_scale"], sd["velocs_prior_log_scale"] = torch.tensor(-6.0), torch.tensor(0.0)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    tables = alanine_dipeptide_amber99sb()
    batch = single_state_batch("ad", types, coords, adj_list=torch.from_numpy(tables.bond_idx.astype(np.int64)))
    energy = AmberPotentialEnergyTorch(tables)
    dev = torch.device("cuda")
    m32 = H.tw_kernel_model(sd, path=1)
    ref = explore(batch, m32, dev, energy, 5, 6, 60.0, noise=DeviceNoise(dev, seed=21))
    m = H.tw_kernel_model(sd, path=3)
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        got = explore(batch, m, dev, energy, 5, 6, 60.0, noise=DeviceNoise(dev, seed=21))
    assert m.demoted and not m32.demoted
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.isfinite(got[0]).all()


def test_mh_iterations_on_the_691_atom_protein_vs_oracle():
    """The reference's SECOND test molecule end to end (testdata/output/1hgv-traj-state0.pdb: a 46-residue, 691-atom protein;
    topology and a frame of coordinates from the committed known-answer fixture, the pinned amber99sb-ildn + OBC tables): whole
    MH iterations with the full-size kernel_transformer_nvp flow.  No fused layout holds 691 atoms, so the flow runs on the per-op
    path - which refused anything above ~200 atoms until r05 (row-wise scores, tiled MFMA mixing; r06: its split-fp16 form,
    TW_PATH_SIMPLE_H3, is what the model takes by default) - inside tw_mh_iteration,
    with the AMBER energy kernel on all 691 atoms, against the oracle loop and the C energy oracle on shared host noise."""
    from timewarp_amd.dataloader import elements_from_atom_names, single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.forcefield import ELEMENT_MASSES, amber99sbildn_obc_tables
    from timewarp_amd.utils.evaluation_utils import MetropolisHastingsChain, sample_with_model

    z = np.load(H.GOLDEN + "/energy_kat_1hgv.npz")
    names = [str(n) for n in z["atom_names"]]
    tables = amber99sbildn_obc_tables(names, [str(r) for r in z["residue_names"]], [int(i) for i in z["residue_ids"]],
                                      improper_neighbour_order="pyset")
    V = len(names)
    assert V == 691
    types = elements_from_atom_names(names)
    coords = torch.from_numpy(z["positions"][0].astype(np.float32))
    masses = torch.tensor([ELEMENT_MASSES[next(ch for ch in n if ch.isalpha())] for n in names], dtype=torch.float32)
    energy = AmberPotentialEnergyTorch(tables)
    # (2 073 degrees of freedom against stiff bonded terms: output layers scaled by 1e-6 and a coordinate prior of e^-9.5 nm give
    # acceptance probabilities between 0 and 1 - three accepted moves in these iterations)
    sd = H.mh_state_dict("scaled", True, out_scale=1e-6, coords_log_scale=-9.5)
    S, N = 4, 9
    kw = dict(accept=True, num_proposal_steps=S, random_velocs=True, resample_velocs=True)
    ref = mo.sample_with_model(types[None], coords[None], torch.zeros(1, V, 3), torch.zeros(1, V, dtype=torch.bool),
                               mo.OracleModel(sd, H.FULL_KERNEL_SPEC), H.OracleAmberEnergy(tables), masses, N, H.HostNoise(5), **kw)
    dev = torch.device("cuda")
    model = H.tw_kernel_model(sd, path=None)     # the constructor's default: split-fp16 fused kernels where a layout exists, else ...
    assert model._path_for(V) == 5               # ... (r06) TW_PATH_SIMPLE_H3: the per-op path with split-fp16 linears / mixing / fused FFN
    chain = MetropolisHastingsChain(single_state_batch("1hgv", types, coords), model, dev, energy, masses, noise=H.HostNoise(5, "cuda"), **kw)
    assert chain._fused                          # the whole iteration as one C-ABI call
    got = sample_with_model(single_state_batch("1hgv", types, coords), model, dev, energy, masses, N, disable_tqdm=True,
                            noise=H.HostNoise(5, "cuda"), **kw)
    print("691-atom protein: accepted", ref[2], "of", len(ref[3].acceptance), "emitted states")
    assert ref[2] >= 1
    (gc, gv, gacc, gs), (rc, rv, racc, rs) = got, ref
    assert gc.shape == rc.shape and gacc == racc
    assert np.array_equal(np.asarray(gs.acceptance_indicator).astype(bool), np.asarray(rs.acceptance_indicator).astype(bool))
    assert H.rel_err(gc, rc) < 1e-5 and H.rel_err(gv, rv) < 1e-5
    for f in ("p_xy", "p_yx", "energies_pot", "energies_kin"):
        a, b = np.asarray(getattr(gs, f), np.float64), np.asarray(getattr(rs, f), np.float64)
        assert H.rel_err(a, b) < 2e-4, (f, H.rel_err(a, b))
    for f, big in (("energies_pot_delta", "energies_pot"), ("energies_kin_delta", "energies_kin")):
        # differences of two fp32 values of the magnitude of `big` (E_pot / kT ~ -1000): a few ulps of THAT
        a, b = np.asarray(getattr(gs, f), np.float64), np.asarray(getattr(rs, f), np.float64)
        mag = float(np.abs(np.asarray(getattr(rs, big), np.float64)).max())
        assert np.abs(a - b).max() < 3e-6 * mag + 1e-6, (f, np.abs(a - b).max(), mag)
    # The exponent is a difference of fp32 quantities of ~2e4 (log-densities of 2 073 coordinates; one ulp = 2e-3) and ~1e3
    # (E / kT): it is resolved to a few ulps of the LARGER terms on either side - the reference's arithmetic included - and the
    # acceptance probability e^-exponent inherits that as a relative error.  Held to 3e-6 of the log-density's magnitude.
    scale = float(np.abs(np.asarray(rs.p_xy, np.float64)).max())
    d_exp = np.abs(np.asarray(gs.exponent, np.float64) - np.asarray(rs.exponent, np.float64)).max()
    print("exponent: max abs difference", d_exp, "of log-densities ~", scale)
    assert d_exp < 3e-6 * scale, (d_exp, scale)
    assert np.allclose(np.asarray(gs.acceptance), np.asarray(rs.acceptance), rtol=0, atol=3 * d_exp + 1e-6)


# ---- tw_mh_iteration_chains (ABI 8): one C-ABI call per lock-step iteration, draws from its own counter-based generator ----

class _KernelDrawsReplay:
    """The DeviceNoise protocol fed from tw_mh_draw_chains: what the fused call draws for (seed, chain, iteration), handed to
    the host-noise route (and to the oracle) draw by draw, in sample_with_model's order: velocities, latents, uniforms."""

    def __init__(self, model, seed, chain, V, device="cuda", init_seed=0):
        self.model, self.seed, self.chain, self.V, self.device, self.it = model, seed, chain, V, device, 0
        self.first = True
        self.g = torch.Generator().manual_seed(init_seed)

    def _draw(self, S):
        from timewarp_amd.utils.multichain import draw_chains

        return [t.to(self.device) for t in draw_chains(self.model, "cuda", self.seed, self.it, self.chain, S, 1, self.V)]

    def randn_like(self, t):
        if self.first:   # the initial velocities of the chain (construction): not one of the iteration's draws
            self.first = False
            return torch.randn(t.shape, generator=self.g).to(self.device)
        return self._draw(1)[3].reshape(t.shape)

    def latents(self, S, B, V, scale_c, scale_v):
        self.first = False
        zc, zv, _, _ = self._draw(S)
        return zc.reshape(S, B, V, 3), zv.reshape(S, B, V, 3)

    def uniform(self, S):
        u = self._draw(S)[2].reshape(S)
        self.it += 1
        return u


def test_chain_draws_match_the_numpy_philox_restatement():
    """tw_mh_draw_chains (= what mhc_begin_kernel generates) against tests/helpers.py's numpy Philox4x32-10 + Box-Muller:
    uniforms bit for bit, normals to float32 rounding; a chain's draws do not depend on how many chains run beside it."""
    from timewarp_amd.utils.multichain import draw_chains

    sd = H.full_kernel_sd()
    sd = {k: v.clone() for k, v in sd.items()}
    model = H.tw_kernel_model(sd, path=3)
    S, Cn, V = 37, 3, 22
    seed, it, first = (0x1234 << 32) | 0x9ABCDEF1, (3 << 32) | 17, 5
    zc, zv, u, v = [t.cpu().numpy() for t in draw_chains(model, "cuda", seed, it, first, S, Cn, V)]
    std_c = float(torch.exp(model.coords_prior_log_scale.detach()))
    std_v = float(torch.exp(model.velocs_prior_log_scale.detach()))
    for c in range(Cn):
        assert np.array_equal(u[:, c], H.chain_draw_uniforms(seed, it, first + c, S))
        for arr, kind, std in ((zc, 0, std_c), (zv, 1, std_v)):
            want = H.chain_draw_normals(seed, it, first + c, kind, S * 3 * V).reshape(S, V, 3) * std
            assert np.abs(arr[:, c] - want).max() < 4e-6 * std, (c, kind, np.abs(arr[:, c] - want).max())
        assert np.abs(v[c] - H.chain_draw_normals(seed, it, first + c, 2, 3 * V).reshape(V, 3)).max() < 4e-6
    # chain `first + 1` alone, more proposals: the same stream, longer
    zc1, _, u1, v1 = [t.cpu().numpy() for t in draw_chains(model, "cuda", seed, it, first + 1, S + 9, 1, V)]
    assert np.array_equal(zc1[:S, 0], zc[:, 1]) and np.array_equal(u1[:S, 0], u[:, 1]) and np.array_equal(v1[0], v[1])
    # moments over a large block
    big = draw_chains(model, "cuda", 99, 0, 0, 1000, 8, V)
    n = (big[1] / std_v).double()
    assert abs(float(n.mean())) < 5e-3 and abs(float(n.std()) - 1) < 5e-3 and abs(float((n ** 4).mean()) - 3) < 0.05
    assert abs(float(big[2].double().mean()) - 0.5) < 5e-3


@pytest.mark.parametrize("random_velocs", [True, False])
def test_kernel_draws_route_equals_host_noise_route(random_velocs):
    """MetropolisHastingsChains without `noises` (draws inside tw_mh_iteration_chains) against the same chains driven through
    the host-noise route with the draws tw_mh_draw_chains writes out - bit for bit - and each chain against a one-chain run
    with first_chain = c (a chain's numbers do not depend on its neighbours)."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.multichain import sample_with_model_chains

    sd = synthetic.synth_state_dict(H.full_kernel_sd(), 0, calibrated=True, coords_log_scale=-7.0, velocs_log_scale=0.0)
    model = H.tw_kernel_model(sd, path=3)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(1)
    starts = [coords + 0.002 * torch.randn(coords.shape, generator=g) for _ in range(3)]
    vel = [0.3 * torch.randn(coords.shape, generator=g) for _ in range(3)]
    kw = dict(random_velocs=random_velocs, resample_velocs=random_velocs)
    batches = lambda idx: [single_state_batch("ad", types, starts[c], vel[c]) for c in idx]
    N, S, seed = 40, 16, 77123
    a = sample_with_model_chains(batches(range(3)), model, dev, energy, masses, N, S, sync_every=3, seed=seed, first_chain=4, **kw)
    b = sample_with_model_chains(batches(range(3)), model, dev, energy, masses, N, S, sync_every=2,
                                 noises=[_KernelDrawsReplay(model, seed, 4 + c, 22) for c in range(3)], **kw)
    H.assert_not_demoted(model)
    assert sum(r[2] for r in a) > 0 or not random_velocs   # (fixed unit-scale velocities: this calibration accepts nothing)
    for ra, rb in zip(a, b):
        assert ra[2] == rb[2] and np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1][1:], rb[1][1:])
        for f in ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin"):
            assert np.array_equal(np.asarray(getattr(ra[3], f)), np.asarray(getattr(rb[3], f))), f
    one = sample_with_model_chains(batches([1]), model, dev, energy, masses, N, S, sync_every=4, seed=seed, first_chain=5, **kw)[0]
    assert one[2] == a[1][2] and one[0].shape == a[1][0].shape
    assert np.allclose(one[0], a[1][0], rtol=0, atol=1e-6) and np.allclose(one[3].exponent, a[1][3].exponent, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("random_velocs,chirality", [(True, False), (False, True)])
def test_fused_chains_iteration_equals_op_by_op_route(monkeypatch, random_velocs, chirality):
    """tw_mh_iteration_chains against the op-by-op lock-step route (TW_MH_FUSED=0: flow sample, energies, kinetic terms,
    chirality test, likelihood of the reverse move, tw_mh_accept_chains as separate calls) on the same per-chain noise."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.utils.chirality import find_chirality_centers, compute_chirality_sign
    from timewarp_amd.utils.multichain import MetropolisHastingsChains, sample_with_model_chains

    sd = synthetic.synth_state_dict(H.full_kernel_sd(), 0, calibrated=True, coords_log_scale=-7.0, velocs_log_scale=0.0)
    model = H.tw_kernel_model(sd, path=3)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    starts = [coords + 0.002 * torch.randn(coords.shape, generator=g) for _ in range(3)]
    vel = [0.3 * torch.randn(coords.shape, generator=g) for _ in range(3)]
    kw = dict(random_velocs=random_velocs, resample_velocs=random_velocs)
    if chirality:
        centres = torch.tensor([[8, 6, 10, 14]])   # CA with N, CB, C (alanine dipeptide atom order)
        kw.update(chirality_centers=centres, reference_signs=compute_chirality_sign(coords[None], centres))
    run = lambda: sample_with_model_chains([single_state_batch("ad", types, x, v) for x, v in zip(starts, vel)], model, dev, energy,
                                           masses, 30, 16, noises=[H.HostNoise(900 + c, "cuda") for c in range(3)], sync_every=2, **kw)
    fused = run()
    monkeypatch.setenv("TW_MH_FUSED", "0")
    probe = MetropolisHastingsChains([single_state_batch("ad", types, starts[0], vel[0])], model, dev, energy, masses, 4)
    assert not probe._fused
    plain = run()
    H.assert_not_demoted(model)
    for ra, rb in zip(fused, plain):
        assert ra[2] == rb[2] and np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
        for f in ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin",
                  "energies_pot_delta", "energies_kin_delta"):
            assert np.array_equal(np.asarray(getattr(ra[3], f)), np.asarray(getattr(rb[3], f))), f
