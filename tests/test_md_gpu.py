"""Force kernel and device Langevin dynamics (csrc/tw_md.hip, timewarp_amd/md.py): what the hybrid moves of
sample_with_model get from OpenMM through `openmm_step` (utils/evaluation_utils.py:439-466) - forces of the AMBER System
(simulation/md.py:292-298) and steps of the integrators simulation/md.py:213-231 builds.

Pinned: the analytic forces against the 40 x 65 x 3 force components of the reference's own OpenMM known-answer file
(simulation/testdata/implicit-2olx-traj-cpu-arrays.npz; tolerances of simulation/tests/test_md.py:47: rtol 0.05 / atol 1e-2)
and against central differences of oracle/energy_oracle.c in all three implicit-solvent modes.  The integrators use
their own noise stream, so they are checked statistically (energy conservation without friction, the thermostat's
temperature), not trace for trace against OpenMM."""
import dataclasses

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_energy_kat import FORCE_RMS_TOL, kat, kat_tables, numerical_forces

pytestmark = pytest.mark.gpu


def test_force_kernel_matches_openmm_known_answers():
    from timewarp_amd.energy import AmberPotentialEnergyTorch

    z = kat()
    e = AmberPotentialEnergyTorch(kat_tables(z))
    x = torch.from_numpy(z["positions"]).cuda()
    en, f = e.energy_and_forces(x)
    f = f.cpu().numpy()
    ref = z["forces"].astype(np.float64)
    assert f.shape == ref.shape == (40, 65, 3)
    rms = float(np.sqrt(((f - ref) ** 2).mean()))
    print(f"force kernel vs OpenMM known answers: rms {rms:.4f} kJ/mol/nm of {np.sqrt((ref ** 2).mean()):.0f}, "
          f"max abs {np.abs(f - ref).max():.4f}")
    assert rms < FORCE_RMS_TOL
    assert np.allclose(f, ref, rtol=0.05, atol=1e-2)           # the reference's own tolerances (test_md.py:47)
    full, _ = e.energy_and_terms(x)
    assert np.allclose(en.cpu().numpy(), full.cpu().numpy(), rtol=1e-12, atol=1e-9)  # same energy as the energy kernel
    assert abs(f.sum(axis=1)).max() < 1e-6                      # no net force on an isolated molecule


@pytest.mark.parametrize("gb", [1, 2, 0])
def test_force_kernel_vs_finite_differences_of_the_c_oracle(gb):
    """gb = 1: GBSA-OBC II, 2: OBC I (the amber14 preset's mode), 0: vacuum; alanine dipeptide, perturbed and one
    stretched conformation (pairs beyond the 2 nm cutoff)."""
    from timewarp_amd.energy import AmberPotentialEnergyTorch

    e = AmberPotentialEnergyTorch.alanine_dipeptide()
    if gb != 1:
        e = AmberPotentialEnergyTorch(dataclasses.replace(e.tables, has_gbsa=gb))
    d, _ = H.load("kernel_full_ad")
    g = torch.Generator().manual_seed(2)
    x = (d["x_coords"] + torch.randn(6, 22, 3, generator=g) * 0.01).double()
    x[5] = d["x_coords"][0].double() * 2.2
    _, f = e.energy_and_forces(x.float().cuda())
    ref = numerical_forces(e.tables, x.float().double().numpy(), h=1e-5)
    err = np.abs(f.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err


def _total_energy(e, masses, x, v, dt):
    """E_pot(x_n) + E_kin at the SAME time: the integrators keep velocities half a step behind the positions
    (v_{n-1/2}; the state after a step is (x_n, v_{n-1/2})), so v_n = v_{n-1/2} + dt/2 F(x_n) / m.  (Taking the staggered
    velocity as it is leaves an O(dt) term dt/2 sum v.F that fluctuates by several kJ/mol.)"""
    ep, f = e.energy_and_forces(x)
    m = masses.cuda()[None, :, None].double()
    vn = v.double() + 0.5 * dt * f / m
    return ep + 0.5 * (m * vn ** 2).sum((1, 2))


@pytest.mark.parametrize("scheme", ["LangevinMiddleIntegrator", "LangevinIntegrator"])
def test_leapfrog_conserves_energy_without_friction(scheme):
    """friction = 0 makes both schemes plain leapfrog.  Alanine dipeptide at 310 K worth of velocities, 0.5 fs (the
    reference's time step, simulation/md.py:80,90): total energy over 4000 steps (2 ps) stays within 0.5 kJ/mol of a
    kinetic energy of ~85; at 0.25 fs the excursion shrinks about fourfold (second-order integrator)."""
    from timewarp_amd import synthetic
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.md import LangevinDynamics

    e = AmberPotentialEnergyTorch.alanine_dipeptide()
    types, coords, masses = synthetic.alanine_dipeptide_state()
    g = torch.Generator().manual_seed(4)
    x0 = coords[None].repeat(8, 1, 1).cuda()
    v0 = (torch.randn(8, 22, 3, generator=g) * (e.kbT / masses)[None, :, None].sqrt()).cuda()
    spread = {}
    for dt in (0.0005, 0.00025):
        md = LangevinDynamics(e, masses, timestep_ps=dt, friction_per_ps=0.0, integrator=scheme)
        x, v = md.step(x0, v0, 200)            # relax the start-up transient of the staggered velocities
        ref = _total_energy(e, masses, x, v, dt)
        worst = torch.zeros_like(ref)
        for _ in range(int(round(2.0 / (dt * 200)))):
            x, v = md.step(x, v, 200)
            worst = torch.maximum(worst, (_total_energy(e, masses, x, v, dt) - ref).abs())
        spread[dt] = float(worst.max())
        assert torch.isfinite(x).all()
    print(f"{scheme}, friction 0: max |E_total - E_total(0)| over 2 ps: {spread[0.0005]:.3f} kJ/mol at 0.5 fs, {spread[0.00025]:.3f} at 0.25 fs")
    assert spread[0.0005] < 0.5
    assert spread[0.00025] < 0.5 * spread[0.0005]


def test_sixteen_wave_dynamics_above_64_atoms():
    """Above 64 atoms a conformation's forces are shared by sixteen waves (amber_forces_block: rows of the pair matrices per wave, no
    cross-wave atomics).  NNQQ (65 atoms, the reference's OpenMM test peptide) and the 691-atom test protein: leapfrog without
    friction conserves the total energy; a thermostatted run is a function of its seed - bit for bit the same twice, different with
    another seed - which is what the row-owner formulation is for."""
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.forcefield import ELEMENT_MASSES, amber99sbildn_obc_tables
    from timewarp_amd.md import LangevinDynamics

    z = kat()
    e = AmberPotentialEnergyTorch(kat_tables(z))
    masses = torch.tensor([ELEMENT_MASSES[str(el)] for el in z["elements"]], dtype=torch.float32)
    g = torch.Generator().manual_seed(11)
    x0 = torch.from_numpy(z["positions"][:4].astype(np.float32)).cuda()
    v0 = (torch.randn(4, 65, 3, generator=g) * (e.kbT / masses)[None, :, None].sqrt()).cuda()
    dt = 0.0005
    md = LangevinDynamics(e, masses, timestep_ps=dt, friction_per_ps=0.0)
    x, v = md.step(x0, v0, 200)
    ref = _total_energy(e, masses, x, v, dt)
    worst = torch.zeros_like(ref)
    for _ in range(10):   # 1 ps
        x, v = md.step(x, v, 200)
        worst = torch.maximum(worst, (_total_energy(e, masses, x, v, dt) - ref).abs())
    print(f"NNQQ, sixteen waves, friction 0: max |E_total - E_total(0)| over 1 ps: {float(worst.max()):.3f} kJ/mol")
    assert torch.isfinite(x).all() and float(worst.max()) < 1.5   # (65 atoms: three times alanine dipeptide's kinetic energy)
    runs = []
    for seed in (5, 5, 6):
        md = LangevinDynamics(e, masses, timestep_ps=dt, friction_per_ps=0.3, seed=seed)
        runs.append(md.step(x0, v0, 300))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert not torch.equal(runs[0][0], runs[2][0])
    # the protein: 40 thermostatted steps twice, and the energy stays finite and close to where it started
    zp = np.load(H.GOLDEN + "/energy_kat_1hgv.npz")
    names = [str(n) for n in zp["atom_names"]]
    ep = AmberPotentialEnergyTorch(amber99sbildn_obc_tables(names, [str(r) for r in zp["residue_names"]], [int(i) for i in zp["residue_ids"]],
                                                            improper_neighbour_order="pyset"))
    mp = torch.tensor([ELEMENT_MASSES[next(ch for ch in n if ch.isalpha())] for n in names], dtype=torch.float32)
    xp = torch.from_numpy(zp["positions"][:2].astype(np.float32)).cuda()
    vp = (torch.randn(2, 691, 3, generator=g) * (ep.kbT / mp)[None, :, None].sqrt()).cuda()
    outs = [LangevinDynamics(ep, mp, timestep_ps=dt, friction_per_ps=0.3, seed=9).step(xp, vp, 40, want_energy=True) for _ in range(2)]
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])
    e0 = ep.energy_and_forces(xp)[0]
    assert torch.isfinite(outs[0][2]).all() and float((outs[0][2] - e0).abs().max()) < 0.05 * float(e0.abs().max())


@pytest.mark.parametrize("scheme", ["LangevinMiddleIntegrator", "LangevinIntegrator"])
def test_thermostat_reaches_the_target_temperature(scheme):
    """256 replicas of alanine dipeptide from rest, friction 10 / ps, 1 fs: after 6 ps the kinetic temperature over
    the next 4 ps is 310 K (66 degrees of freedom x 256 replicas x 20 snapshots: the mean is good to ~0.5 %)."""
    from timewarp_amd import synthetic
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.md import LangevinDynamics

    e = AmberPotentialEnergyTorch.alanine_dipeptide()
    types, coords, masses = synthetic.alanine_dipeptide_state()
    md = LangevinDynamics(e, masses, timestep_ps=0.001, friction_per_ps=10.0, integrator=scheme, seed=11)
    x = coords[None].repeat(256, 1, 1).cuda()
    v = torch.zeros_like(x)
    x, v = md.step(x, v, 6000)
    temps = []
    for _ in range(20):
        x, v = md.step(x, v, 200)
        ke = 0.5 * (masses.cuda()[None, :, None] * v.double() ** 2).sum((1, 2))      # kJ/mol
        temps.append(float((2.0 * ke / (66 * 8.314462618e-3)).mean()))
    t = float(np.mean(temps))
    print(f"{scheme}: kinetic temperature {t:.1f} K (target 310)")
    assert torch.isfinite(x).all() and abs(t - 310.0) < 0.03 * 310.0
    # replicas decorrelate: the noise differs per conformation
    assert float((x[0] - x[1]).abs().max()) > 1e-3
    # reproducible for a seed
    md2 = LangevinDynamics(e, masses, timestep_ps=0.001, friction_per_ps=10.0, integrator=scheme, seed=11)
    a, _ = md2.step(coords[None].repeat(4, 1, 1).cuda(), torch.zeros(4, 22, 3).cuda(), 50)
    md3 = LangevinDynamics(e, masses, timestep_ps=0.001, friction_per_ps=10.0, integrator=scheme, seed=11)
    b, _ = md3.step(coords[None].repeat(4, 1, 1).cuda(), torch.zeros(4, 22, 3).cuda(), 50)
    assert torch.equal(a, b)


@pytest.mark.parametrize("where", ["current", "proposal"])
def test_hybrid_moves_run_on_the_device_without_a_simulation(where):
    """sample_with_model's hybrid moves (evaluation_utils.py:558-565, 594-602, 623-626) with sim="device" (opt-in): the chain
    integrates by itself on the HIP force kernel (LangevinDynamics.from_preset with the integrator of the energy's dataset
    preset - alanine dipeptide: amber99-implicit-old = LangevinIntegrator, 310 K, 0.3 / ps, 0.5 fs - and a seed drawn from
    the chain's own noise source).  The reference needs the caller's OpenMM objects for these options; here the states move
    on the device - no host copy of coordinates happens inside `openmm_step`.  sim=None keeps the reference's meaning: the
    options are silently off."""
    from timewarp_amd import synthetic
    from timewarp_amd.dataloader import single_state_batch
    from timewarp_amd.energy import AmberPotentialEnergyTorch
    from timewarp_amd.md import LangevinDynamics
    from timewarp_amd.utils import evaluation_utils as eu

    sd = H.mh_state_dict("scaled", True)
    model = H.tw_kernel_model(sd, path=3)
    types, coords, masses = synthetic.alanine_dipeptide_state()
    energy = AmberPotentialEnergyTorch.alanine_dipeptide()
    S = 1 if where == "proposal" else 16   # openmm_on_proposal handles one proposal per iteration, as the reference
    kw = dict(accept=True, num_proposal_steps=S, random_velocs=True, resample_velocs=True, num_openmm_steps=5,
              openmm_on_current=(where == "current"), openmm_on_proposal=(where == "proposal"))
    off = eu.MetropolisHastingsChain(single_state_batch("ad", types, coords), model, torch.device("cuda"), energy, masses,
                                     noise=H.HostNoise(3, "cuda"), **kw)
    assert off.sim is None and not (off.omm_current or off.omm_proposal)   # as the reference: `... and sim is not None`
    chain = eu.MetropolisHastingsChain(single_state_batch("ad", types, coords), model, torch.device("cuda"), energy, masses,
                                       noise=H.HostNoise(3, "cuda"), sim="device", **kw)
    assert isinstance(chain.sim, LangevinDynamics) and (chain.omm_current or chain.omm_proposal)
    assert chain.sim.integrator == "LangevinIntegrator"   # alanine dipeptide's preset (simulation/md.py:31-37, 75-82)
    other = eu.MetropolisHastingsChain(single_state_batch("ad", types, coords), model, torch.device("cuda"), energy, masses,
                                       noise=H.HostNoise(4, "cuda"), sim="device", **kw)
    assert other.sim.seed != chain.sim.seed    # another chain's noise -> another thermostat stream
    seen = []
    real = eu.openmm_step

    def spy(sim, c, v=None, num_steps=1, integrator=None):
        assert c.is_cuda
        out = real(sim, c, v, num_steps, integrator)
        assert out[0].is_cuda and out[0].shape == c.shape
        seen.append(float((out[0] - c).abs().max()))
        return out

    eu.openmm_step = spy
    try:
        emitted = sum(chain.step() for _ in range(6))
    finally:
        eu.openmm_step = real
    c, v, accepted, stats = chain.result()
    assert len(seen) >= 6 and min(seen) > 0 and max(seen) < 0.05     # five 0.5 fs steps move atoms by ~1e-3 nm
    assert c.shape[0] == emitted + 1 and np.isfinite(stats.exponent).all()
    H.assert_not_demoted(model)
    if where == "current":
        assert accepted >= 1
