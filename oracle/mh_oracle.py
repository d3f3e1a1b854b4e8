"""CPU oracle for the batched Metropolis-Hastings loop.  TEST INFRASTRUCTURE ONLY.

Restates `sample_with_model` (utils/evaluation_utils.py:468-745), `compute_kinetic_energy`
(:416-436), `compute_num_proposal_steps` (:32-64) and `compute_chirality_sign` /
`check_symmetry_change` (utils/chirality.py:40-80) on torch-CPU, with every random draw taken
from an explicit noise source so that a run can be replayed.

Pinning: `oracle/gen_golden.py::gen_mh_goldens` runs the REAL reference function (imported from
/root/reference with its third-party imports stubbed; the energy is a synthetic callable passed in
as an argument) and records inputs, every random draw and all outputs in
tests/golden/mh_tiny.npz (and, with oracle/fake_sim.FakeSimulation as the OpenMM Simulation, in mh_tiny_openmm.npz);
tests/test_mh_oracle.py replays them through this file.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import flow_oracle as fo


def compute_num_proposal_steps(p_acc: float, target: float = 0.9, max_steps: int = 100) -> int:
    """evaluation_utils.py:32-64."""
    p_rej = min(max(1 - p_acc, 1e-3), 1 - 1e-3)
    with np.errstate(all="ignore"):
        n = np.nan_to_num(np.log(1 - target) / np.log(p_rej), nan=np.inf)
    return max(int(np.ceil(min(n, max_steps))), 1)


def compute_kinetic_energy(velocs, masses, random_velocs=False, kbT=None):
    """evaluation_utils.py:416-436."""
    if random_velocs:
        return 0.5 * ((velocs**2.0).sum(-1)).sum(-1)
    assert kbT
    return 0.5 * (masses * (velocs**2.0).sum(-1)).sum(-1) / kbT


def compute_chirality_sign(coords, centres):
    """chirality.py:40-62."""
    d = coords[:, centres[:, 1:], :] - coords[:, centres[:, [0]], :]
    s = torch.einsum("ijk,ijk->ij", d[:, :, 0], torch.cross(d[:, :, 1], d[:, :, 2], dim=-1))
    return torch.sign(s)


def check_symmetry_change(coords, centres, reference_signs):
    """chirality.py:65-80."""
    return (compute_chirality_sign(coords, centres) != reference_signs.to(coords)).any(dim=-1)


class SyntheticEnergy:
    """Same closed form as gen_golden.SyntheticEnergy: k*sum|x-x_ref|^2 + w*sum_{i<j} exp(-r_ij^2)."""

    def __init__(self, x_ref, k=40.0, w=3.0, kbT=2.5):
        self.x_ref, self.k, self.w, self.kbT = x_ref, k, w, kbT

    def __call__(self, coords):
        c = coords.reshape(-1, self.x_ref.shape[-2], 3)
        x_ref = self.x_ref.to(c)
        e = self.k * ((c - x_ref) ** 2).sum(dim=(-1, -2))
        d2 = ((c[:, :, None, :] - c[:, None, :, :]) ** 2).sum(-1)
        iu = torch.triu_indices(c.shape[1], c.shape[1], offset=1).to(c.device)
        e = e + self.w * torch.exp(-d2[:, iu[0], iu[1]]).sum(-1)
        return e[:, None]


class ReplayNoise:
    """Feeds back the draws recorded from a reference run, in the reference's order."""

    def __init__(self, normal, normal_sizes, rand, rand_sizes, randn_like, device="cpu"):
        self.normal = [torch.as_tensor(a) for a in np.split(np.asarray(normal), np.cumsum(normal_sizes)[:-1])] if len(normal_sizes) else []
        self.rand = [torch.as_tensor(a) for a in np.split(np.asarray(rand), np.cumsum(rand_sizes)[:-1])] if len(rand_sizes) else []
        self.randn_like_buf = torch.as_tensor(np.asarray(randn_like))
        self.pos = 0
        self.device = device

    def randn_like(self, t):
        n = t.numel()
        out = self.randn_like_buf[self.pos:self.pos + n].reshape(t.shape).to(t)
        self.pos += n
        return out

    def latents(self, S, B, V, scale_c, scale_v):
        zc = self.normal.pop(0).reshape(S, B, V, 3)
        zv = self.normal.pop(0).reshape(S, B, V, 3)
        return zc.to(self.device), zv.to(self.device)

    def uniform(self, S):
        u = self.rand.pop(0)
        assert u.numel() == S
        return u.to(self.device)

    def rotation(self):
        raise NotImplementedError


class OracleModel:
    """flow_oracle behind the two model calls the loop makes."""

    def __init__(self, sd, spec):
        self.sd, self.spec = sd, spec

    def scales(self):
        return torch.exp(self.sd["coords_prior_log_scale"]), torch.exp(self.sd["velocs_prior_log_scale"])

    def conditional_sample_with_logp(self, atom_types, x_coords, x_velocs, masked, z_coords, z_velocs):
        return fo.conditional_sample_with_logp(self.sd, self.spec, atom_types, x_coords, x_velocs, masked, z_coords, z_velocs)

    def log_likelihood(self, atom_types, x_coords, x_velocs, y_coords, y_velocs, masked):
        return fo.log_likelihood(self.sd, self.spec, atom_types, x_coords, x_velocs, y_coords, y_velocs, masked)


@dataclass
class ChainStats:
    acceptance_indicator: np.ndarray
    acceptance: np.ndarray
    p_xy: np.ndarray
    p_yx: np.ndarray
    exponent: np.ndarray
    energies_pot: np.ndarray
    energies_kin: np.ndarray
    energies_pot_delta: np.ndarray
    energies_kin_delta: np.ndarray


def openmm_step(sim, coords, velocs, num_steps=1):
    """evaluation_utils.py:439-466 (the branch with velocities, the only one the loop takes)."""
    sim.context.setPositions(coords.cpu().numpy().squeeze(0))
    sim.context.setVelocities(velocs.cpu().numpy().squeeze(0))
    sim.step(num_steps)
    state = sim.context.getState(getPositions=True, getVelocities=True)
    c = torch.from_numpy(state.getPositions(asNumpy=True)._value).reshape(coords.shape).to(coords)
    v = torch.from_numpy(state.getVelocities(asNumpy=True)._value).reshape(coords.shape).to(coords)
    return c, v


def sample_with_model(atom_types, x_coords, x_velocs, masked, model: OracleModel, energy, masses, num_samples: int,
                      noise, accept=False, random_velocs=False, resample_velocs=False, initialize_randomly=False,
                      num_proposal_steps=1, adaptive_parallelism=False, acceptance_rate_smoothing_factor=0.01,
                      reference_signs=None, chirality_centers=None, num_openmm_steps=0, sim=None,
                      openmm_on_proposal=False, openmm_on_current=False):
    """evaluation_utils.py:517-745 (the rotate option omitted: it raises in the reference).  `sim`: anything with the
    five calls of `openmm_step`.  Inputs are [1,V,...] tensors (B == 1 is asserted at :517)."""
    assert x_coords.size(0) == 1, "only batch-size of 1 is supported"
    names = ["ind", "acc", "pxy", "pyx", "exp", "epot", "ekin", "dpot", "dkin"]
    rec = {n: [] for n in names}
    x_coords = x_coords.contiguous()
    x_velocs = noise.randn_like(x_coords) if random_velocs else x_velocs.contiguous()  # :529-533
    sc, sv = model.scales()
    B, V = x_coords.shape[0], x_coords.shape[1]
    if initialize_randomly:  # :540-553
        rc, rv = noise.randn_like(x_coords), noise.randn_like(x_velocs)
        zc, zv = noise.latents(1, B, V, sc, sv)
        yc, yv, _ = model.conditional_sample_with_logp(atom_types, rc, rv, masked, zc, zv)
        x_coords, x_velocs = yc.squeeze(0), yv.squeeze(0)
    kbT = energy.kbT
    velocs_std = (kbT / masses.unsqueeze(0).unsqueeze(-1)).sqrt()  # :556
    on_current = openmm_on_current and num_openmm_steps > 0 and sim is not None

    def step_current(x_c, x_v):  # :558-565 and :594-602
        if random_velocs:
            return openmm_step(sim, x_c, x_v * velocs_std, num_openmm_steps)[0], x_v
        return openmm_step(sim, x_c, x_v, num_openmm_steps)

    if on_current:
        x_coords, x_velocs = step_current(x_coords, x_velocs)
    coords_out, velocs_out = [x_coords.numpy().copy()], [x_velocs.numpy().copy()]  # :566-567
    accepted = 0
    p_bar = 1e-3  # :575
    s_max = num_proposal_steps
    S = num_proposal_steps if not adaptive_parallelism else compute_num_proposal_steps(p_bar, max_steps=s_max)
    i = 0
    while i < num_samples:  # :589
        if random_velocs and resample_velocs:
            x_velocs = noise.randn_like(x_velocs)  # :590-592
        if on_current:
            x_coords, x_velocs = step_current(x_coords, x_velocs)
        zc, zv = noise.latents(S, B, V, sc, sv)
        y_c, y_v, p_xy = model.conditional_sample_with_logp(atom_types, x_coords, x_velocs, masked, zc, zv)  # :609-617
        y_c, y_v = y_c.squeeze(1), y_v.squeeze(1)
        X_c, X_v = x_coords.repeat(S, 1, 1), x_velocs.repeat(S, 1, 1)  # :620-621
        if openmm_on_proposal and sim is not None and num_openmm_steps > 0:  # :623-626
            y_c, _ = openmm_step(sim, y_c, y_v * velocs_std, num_openmm_steps)
        e_pot_x = (energy(X_c) / kbT).squeeze(-1)  # :628
        e_kin_x = compute_kinetic_energy(X_v, masses, random_velocs, kbT)
        e_kin_y = compute_kinetic_energy(y_v, masses, random_velocs, kbT)
        e_kin = e_kin_y - e_kin_x
        e_pot_y = (energy(y_c) / kbT).squeeze(-1)  # :635
        if chirality_centers is not None and reference_signs is not None:
            e_pot_y[check_symmetry_change(y_c, chirality_centers, reference_signs)] += 2000  # :638-642
        e_pot = e_pot_y - e_pot_x
        en = e_pot + e_kin
        sgn = 1.0 if random_velocs else -1.0
        p_yx = model.log_likelihood(atom_types.repeat(S, 1), y_c, sgn * y_v, X_c, sgn * X_v, masked.repeat(S, 1))  # :648-657
        p_xy = p_xy.reshape(p_yx.shape)
        ex = en + p_xy - p_yx  # :663
        p_acc = torch.min(torch.tensor(1.0), torch.exp(-ex))  # :665
        if accept:
            acc = noise.uniform(S).to(p_acc) < p_acc  # :668
            idx = acc.nonzero(as_tuple=True)[0]
            none = len(idx) == 0
            if none:
                k = S - 1
            else:
                k = int(idx[0])
                X_c[k], X_v[k] = y_c[k], y_v[k]  # :675-676
                accepted += 1
            k = min(k, num_samples - i)  # :680
            rec["ind"].append(acc[: k + 1].numpy())
            p_bar = acceptance_rate_smoothing_factor * (1 - none) + (1 - acceptance_rate_smoothing_factor) ** k * p_bar  # :685-689
            S_next = S if not adaptive_parallelism else compute_num_proposal_steps(p_bar, max_steps=s_max)
        elif S == 1:  # :698-705
            X_c, X_v = y_c, y_v
            accepted += 1
            k = 0
            rec["ind"].append(np.array([True]))
            S_next = S
        else:
            raise ValueError("Number of proposals has to be one if everything is accepted!")
        coords_out.append(X_c[: k + 1].numpy().copy())  # :709-710
        velocs_out.append(X_v[: k + 1].numpy().copy())
        x_coords, x_velocs = X_c[k].unsqueeze(0), X_v[k].unsqueeze(0)  # :712-713
        i += k + 1
        for n, t in (("acc", p_acc), ("pxy", p_xy), ("pyx", p_yx), ("exp", ex), ("epot", e_pot_y), ("ekin", e_kin_y),
                     ("dpot", e_pot), ("dkin", e_kin)):
            rec[n].append(t.numpy()[: k + 1])  # :721-728
        S = S_next
    stats = ChainStats(*[np.concatenate(rec[n], axis=0) for n in names])
    return np.concatenate(coords_out, axis=0), np.concatenate(velocs_out, axis=0), accepted, stats


def sample_on_batches(batches, model: OracleModel, energy, masses, noise, random_velocs=False):
    """Restatement of utils/evaluation_utils.py:190-333 (data_augmentation=False).  `batches` is a list
    of dicts with atom_types [1,V], x, v, y, w [1,V,3]; returns the reference's eleven arrays."""
    kbT = energy.kbT
    sgn = 1.0 if random_velocs else -1.0
    cols = {k: [] for k in ("y_c", "y_v", "t_c", "t_v", "c_c", "c_v", "p_xy", "p_yx", "p_xy_tr", "p_yx_tr", "acc")}
    for b in batches:
        at, x_c, y_t = b["atom_types"], b["x"], b["y"]
        mk = torch.zeros(at.shape, dtype=torch.bool)
        if random_velocs:
            x_v = noise.randn_like(x_c)          # :236
            w_t = noise.randn_like(y_t)          # :237
        else:
            x_v, w_t = b["v"], b["w"]
        sc, sv = model.scales()
        z_c, z_v = noise.latents(1, x_c.shape[0], x_c.shape[1], sc, sv)
        y_c, y_v, _ = model.conditional_sample_with_logp(at, x_c, x_v, mk, z_c, z_v)   # :242-250 (num_samples=1)
        y_c, y_v = y_c.squeeze(0), y_v.squeeze(0)
        p_xy = model.log_likelihood(at, x_c, x_v, y_c, y_v, mk)                        # :253-262
        e_kin = compute_kinetic_energy(y_v, masses, random_velocs, kbT) - compute_kinetic_energy(x_v, masses, random_velocs, kbT)
        e_pot = ((energy(y_c) - energy(x_c)) / kbT).view(-1)                           # :268-271
        p_yx = model.log_likelihood(at, y_c, sgn * y_v, x_c, sgn * x_v, mk)            # :275-284
        ex = e_pot + e_kin + p_xy - p_yx                                               # :288
        p_acc = torch.min(torch.tensor(1.0), torch.exp(-ex))                           # :289
        p_xy_tr = model.log_likelihood(at, x_c, x_v, y_t, w_t, mk)                     # :292-301
        p_yx_tr = model.log_likelihood(at, y_t, sgn * w_t, x_c, sgn * x_v, mk)         # :303-312
        for k, t in (("acc", p_acc), ("p_xy", p_xy), ("p_yx", p_yx), ("p_xy_tr", p_xy_tr), ("p_yx_tr", p_yx_tr),
                     ("y_c", y_c), ("y_v", y_v), ("c_c", x_c), ("c_v", x_v), ("t_c", y_t), ("t_v", b["w"])):
            cols[k].append(t.detach().numpy())
    arr = {k: np.array(v) for k, v in cols.items()}
    sq = lambda a: a.squeeze(1)
    return (sq(arr["y_c"]), sq(arr["y_v"]), sq(arr["t_c"]), sq(arr["t_v"]), sq(arr["c_c"]), sq(arr["c_v"]),
            arr["p_yx"], arr["p_xy"], arr["p_yx_tr"], arr["p_xy_tr"], arr["acc"])


def sample_on_single_conditional(atom_types, x_coords, x_velocs, masked, model: OracleModel, num_samples, sim, step_width,
                                 random_velocs, noise):
    """Restatement of utils/evaluation_utils.py:356-413: `num_samples` model samples and `num_samples` OpenMM segments of
    `step_width` steps, all from the one conditioning state.  Returns the reference's five arrays."""
    positions, velocities, yc_all, yv_all = [], [], [], []
    sc, sv = model.scales()
    B, V = x_coords.shape[0], x_coords.shape[1]
    for _ in range(num_samples):
        sim.context.setPositions(x_coords.numpy().squeeze(0))                      # :375
        if random_velocs:
            sim.context.setVelocitiesToTemperature(sim.integrator.getTemperature())  # :377
            sim.context.getState(getPositions=True, getVelocities=True)
            xv = noise.randn_like(x_velocs)                                         # :380
        else:
            xv = x_velocs
            sim.context.setVelocities(xv.numpy().squeeze(0))                        # :383
        zc, zv = noise.latents(1, B, V, sc, sv)
        y_c, y_v, _ = model.conditional_sample_with_logp(atom_types, x_coords, xv.float(), masked, zc, zv)  # :385-393
        sim.step(step_width)                                                        # :395
        state = sim.context.getState(getPositions=True, getVelocities=True)
        positions.append(state.getPositions(asNumpy=True)._value)
        velocities.append(state.getVelocities(asNumpy=True)._value)
        yc_all.append(y_c.numpy())
        yv_all.append(y_v.numpy())
    return (np.array(yc_all).squeeze(1).squeeze(1), np.array(yv_all).squeeze(1).squeeze(1), np.array(positions),
            np.array(velocities), np.array(x_coords.numpy()))


def explore(atom_types, x_coords, x_velocs, masked, model: OracleModel, energy, chirality_centres, num_steps: int,
            num_parallel_steps: int, energy_threshold: float, noise):
    """The exploration loop of the reference's exploration.py:229-257 (no MH correction: an explorer moves to the proposal
    unless the potential energy rises by more than `energy_threshold`; a flipped chirality centre adds 10000), written after
    the script line by line - the loop sits inline in its `main` and cannot be imported, so this restatement is pinned only
    through the model / energy / chirality calls it is made of (parity unpinned as a whole).  Returns (positions
    [num_steps * P, V, 3], energies [num_steps * P, 1])."""
    P = num_parallel_steps
    signs = compute_chirality_sign(x_coords, chirality_centres) if len(chirality_centres) else None   # :230
    y_c, y_v = x_coords, x_velocs                                                                     # :232-233
    energies = energy(y_c).repeat(P, 1)                                                               # :236-237
    y_c, y_v = y_c.repeat(P, 1, 1), y_v.repeat(P, 1, 1)
    at, mk = atom_types.repeat(P, 1), masked.repeat(P, 1)
    sc, sv = model.scales()
    traj, elog = [], []
    for _ in range(num_steps):
        z_c, z_v = noise.latents(1, P, y_c.shape[1], sc, sv)
        y_new, _, _ = model.conditional_sample_with_logp(at, y_c, y_v, mk, z_c, z_v)                  # :122-134
        y_new = y_new.squeeze(0)
        e_new = energy(y_new)                                                                         # :242
        if signs is not None:
            e_new = e_new.clone()
            e_new[check_symmetry_change(y_new, chirality_centres, signs)] += 10000                    # :243-245
        stay = e_new - energies > energy_threshold
        y_c = torch.where(stay.unsqueeze(-1), y_c, y_new)                                             # :246-248
        energies = torch.where(stay, energies, e_new)                                                 # :249
        traj.append(y_c)
        elog.append(energies)
        y_v = noise.randn_like(y_c)                                                                   # :253
    return torch.cat(traj, 0), torch.cat(elog, 0)
