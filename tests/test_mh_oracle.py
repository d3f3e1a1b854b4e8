"""Pin oracle/mh_oracle.py against traces recorded from the reference's own sample_with_model
(oracle/gen_golden.py::gen_mh_goldens).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as fo
from oracle import mh_oracle as mo
from tests import helpers as H

SCENARIOS = {
    "s10": dict(accept=True, num_proposal_steps=10, num_samples=25),
    "s10_randv": dict(accept=True, num_proposal_steps=10, num_samples=25, random_velocs=True, resample_velocs=True),
    "adaptive": dict(accept=True, num_proposal_steps=10, num_samples=30, adaptive_parallelism=True),
    "noaccept_s1": dict(accept=False, num_proposal_steps=1, num_samples=6),
    "chirality": dict(accept=True, num_proposal_steps=10, num_samples=20, chirality=True),
    "init_random": dict(accept=True, num_proposal_steps=4, num_samples=8, initialize_randomly=True),
}
# OpenMM steps inside the chain, recorded from the reference with oracle/fake_sim.FakeSimulation as the Simulation
OPENMM_SCENARIOS = {
    "omm_current": dict(accept=True, num_proposal_steps=10, num_samples=20, num_openmm_steps=3, openmm_on_current=True),
    "omm_current_randv": dict(accept=True, num_proposal_steps=10, num_samples=20, random_velocs=True, resample_velocs=True,
                              num_openmm_steps=2, openmm_on_current=True),
    "omm_proposal": dict(accept=True, num_proposal_steps=1, num_samples=10, num_openmm_steps=2, openmm_on_proposal=True),
    "omm_both_noaccept": dict(accept=False, num_proposal_steps=1, num_samples=6, num_openmm_steps=1, openmm_on_proposal=True,
                              openmm_on_current=True),
}
STAT_FIELDS = ("acceptance_indicator", "acceptance", "p_xy", "p_yx", "exponent", "energies_pot", "energies_kin",
               "energies_pot_delta", "energies_kin_delta")


def load_mh(file="mh_tiny.npz"):
    import os
    z = np.load(os.path.join(H.GOLDEN, file))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return z, sd


def replay(z, name, device="cpu"):
    return mo.ReplayNoise(z[name + "/noise_normal"], z[name + "/noise_normal_sizes"], z[name + "/noise_rand"],
                          z[name + "/noise_rand_sizes"], z[name + "/noise_randn_like"], device=device)


def check_against_golden(z, name, coords, velocs, accepted, stats, tol=2e-5):
    assert coords.shape == z[name + "/coords"].shape
    assert int(accepted) == int(z[name + "/accepted"])
    assert H.rel_err(coords, z[name + "/coords"]) < tol
    assert H.rel_err(velocs, z[name + "/velocs"]) < tol
    for f in STAT_FIELDS:
        a, b = np.asarray(getattr(stats, f)), z[f"{name}/stats_{f}"]
        assert a.shape == b.shape, f
        if f == "acceptance_indicator":
            assert (a.astype(bool) == b.astype(bool)).all()
        else:
            assert H.rel_err(a.astype(np.float64), b.astype(np.float64)) < 1e-4, (f, H.rel_err(a, b))
    # the reference's own invariant (tests/test_evaluation_utils.py:138)
    assert len(coords) == len(stats.acceptance) + 1


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_mh_oracle_replays_reference(name):
    z, sd = load_mh()
    kw = dict(SCENARIOS[name])
    extra = {}
    if kw.pop("chirality", False):
        extra = dict(chirality_centers=torch.from_numpy(z["centres"]),
                     reference_signs=torch.from_numpy(z[name + "/reference_signs"]))
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    model = mo.OracleModel(sd, H.TINY_KERNEL_SPEC)
    energy = mo.SyntheticEnergy(x0.clone())
    coords, velocs, accepted, stats = mo.sample_with_model(
        torch.from_numpy(z["atom_types"]), x0, v0, torch.zeros(1, x0.shape[1], dtype=torch.bool), model, energy,
        torch.from_numpy(z["masses"]), noise=replay(z, name), **kw, **extra)
    check_against_golden(z, name, coords, velocs, accepted, stats)


@pytest.mark.parametrize("name", list(OPENMM_SCENARIOS))
def test_mh_oracle_replays_reference_with_openmm_steps(name):
    """evaluation_utils.py:558-565, 594-602, 623-626: the Simulation is the caller's object; the recorded runs used
    oracle/fake_sim.FakeSimulation, and so does the replay."""
    from oracle.fake_sim import FakeSimulation

    z, sd = load_mh("mh_tiny_openmm.npz")
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    sim = FakeSimulation()
    coords, velocs, accepted, stats = mo.sample_with_model(
        torch.from_numpy(z["atom_types"]), x0, v0, torch.zeros(1, x0.shape[1], dtype=torch.bool),
        mo.OracleModel(sd, H.TINY_KERNEL_SPEC), mo.SyntheticEnergy(x0.clone()), torch.from_numpy(z["masses"]),
        noise=replay(z, name), sim=sim, **OPENMM_SCENARIOS[name])
    check_against_golden(z, name, coords, velocs, accepted, stats)
    assert sim.calls > 0


def test_compute_num_proposal_steps():
    # evaluation_utils.py:32-64: ceil(log(1-target)/log(1-p)) clipped to [1, max]
    assert mo.compute_num_proposal_steps(1e-3, max_steps=100) == 100
    assert mo.compute_num_proposal_steps(0.5, max_steps=100) == 4
    assert mo.compute_num_proposal_steps(1.0, max_steps=100) == 1
    assert mo.compute_num_proposal_steps(0.0, max_steps=7) == 7


# ------------------------------------------------------------------ sample_on_batches (evaluation_utils.py:190-333)
SOB_NAMES = ("y_coords_model", "y_velocs_model", "traj_coords", "traj_velocs", "traj_coords_conditioning",
             "traj_velocs_conditioning", "ll_reverse", "ll_forward", "ll_reverse_training", "ll_forward_training",
             "acceptance")


def load_sob():
    import os
    z = np.load(os.path.join(H.GOLDEN, "sob_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    batches = [dict(atom_types=torch.from_numpy(z[f"batch{b}/atom_types"]), **{k: torch.from_numpy(z[f"batch{b}/{k}"]) for k in "xvyw"})
               for b in range(3)]
    return z, sd, batches


def sob_replay(z, tag, device="cpu"):
    return mo.ReplayNoise(z[tag + "/noise_normal"], z[tag + "/noise_normal_sizes"], np.zeros(0, np.float32), [],
                          z[tag + "/noise_randn_like"], device=device)


def check_sob(z, tag, res, tol=2e-5):
    for n, a in zip(SOB_NAMES, res):
        b = z[f"{tag}/{n}"]
        assert np.asarray(a).shape == b.shape, n
        assert H.rel_err(np.asarray(a, np.float64), b.astype(np.float64)) < (1e-4 if n.startswith("ll_") or n == "acceptance" else tol), n


@pytest.mark.parametrize("tag,random_velocs", [("fixedv", False), ("randv", True)])
def test_sample_on_batches_oracle_replays_reference(tag, random_velocs):
    z, sd, batches = load_sob()
    model = mo.OracleModel(sd, H.TINY_KERNEL_SPEC)
    energy = mo.SyntheticEnergy(torch.from_numpy(z["x_ref"]).clone())
    res = mo.sample_on_batches(batches, model, energy, torch.from_numpy(z["masses"]), sob_replay(z, tag), random_velocs)
    check_sob(z, tag, res)


# ------------------------------------------------------------------ sample_on_single_conditional (evaluation_utils.py:356-413)
SOSC_NAMES = ("y_coords_model", "y_velocs_model", "traj_coords", "traj_velocs", "traj_coords_conditioning")


def load_sosc():
    import os
    z = np.load(os.path.join(H.GOLDEN, "sosc_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return z, sd


def check_sosc(z, tag, res, tol=2e-5):
    for n, a in zip(SOSC_NAMES, res):
        b = z[f"{tag}/{n}"]
        assert np.asarray(a).shape == b.shape, n
        assert H.rel_err(np.asarray(a, np.float64), b.astype(np.float64)) < tol, n


@pytest.mark.parametrize("tag,random_velocs", [("fixedv", False), ("randv", True)])
def test_sample_on_single_conditional_oracle_replays_reference(tag, random_velocs):
    """Recorded from the reference's own function with oracle/fake_sim.FakeSimulation (thermal velocities allowed) as
    the Simulation: five model samples and five 3-step segments from one conditioning state."""
    from oracle.fake_sim import FakeSimulation

    z, sd = load_sosc()
    x0, v0 = torch.from_numpy(z["x0"]), torch.from_numpy(z["v0"])
    res = mo.sample_on_single_conditional(
        torch.from_numpy(z["atom_types"]), x0, v0, torch.zeros(1, x0.shape[1], dtype=torch.bool),
        mo.OracleModel(sd, H.TINY_KERNEL_SPEC), int(z["num_samples"]), FakeSimulation(allow_thermal=True), int(z["step_width"]),
        random_velocs, sob_replay(z, tag))
    check_sosc(z, tag, res)
