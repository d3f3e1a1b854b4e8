"""torch float64 restatement of oracle/energy_oracle.c (GBSA-OBC II branch) on a ForceFieldTables, vectorised over
frames so that autograd gives forces.  Analysis tooling only (tools/pin_energy/fit_2olx.py)."""
import math

import torch

K_E = 138.935456


def dihedral(x, idx):
    a, b, c, d = (x[:, idx[:, k]] for k in range(4))
    r0, r1, r2 = a - b, c - b, c - d
    c0, c1 = torch.cross(r0, r1, dim=-1), torch.cross(r1, r2, dim=-1)
    y = (torch.cross(c0, c1, dim=-1) * r1).sum(-1) / r1.norm(dim=-1)
    phi = torch.atan2(y, (c0 * c1).sum(-1))
    return phi.abs() * torch.sign((r0 * c1).sum(-1))  # the oracle's sign convention


def energy_terms(x, t, eps_solvent=None, sa=None):
    """x [F,V,3] float64, t ForceFieldTables -> dict of [F] energies (kJ/mol)"""
    T = lambda a, dt=torch.float64: torch.as_tensor(a, dtype=dt)
    out = {}
    bi, bp = T(t.bond_idx, torch.long), T(t.bond_par)
    d = (x[:, bi[:, 0]] - x[:, bi[:, 1]]).norm(dim=-1) - bp[:, 0]
    out["bond"] = (0.5 * bp[:, 1] * d * d).sum(-1)
    ai, ap = T(t.angle_idx, torch.long), T(t.angle_par)
    v0, v1 = x[:, ai[:, 0]] - x[:, ai[:, 1]], x[:, ai[:, 2]] - x[:, ai[:, 1]]
    th = torch.acos(((v0 * v1).sum(-1) / (v0.norm(dim=-1) * v1.norm(dim=-1))).clamp(-1, 1))
    out["angle"] = (0.5 * ap[:, 1] * (th - ap[:, 0]) ** 2).sum(-1)
    ti, tp = T(t.torsion_idx, torch.long), T(t.torsion_par)
    out["torsion"] = (tp[:, 2] * (1 + torch.cos(tp[:, 0] * dihedral(x, ti) - tp[:, 1]))).sum(-1)
    V, rc = x.shape[1], t.cutoff
    q, sg, ep = T(t.atom_par[:, 0]), T(t.atom_par[:, 1]), T(t.atom_par[:, 2])
    iu = torch.triu_indices(V, V, 1)
    mask = torch.ones(V, V, dtype=torch.bool)
    ei = T(t.exc_idx, torch.long)
    mask[ei[:, 0], ei[:, 1]] = False
    mask[ei[:, 1], ei[:, 0]] = False
    keep = mask[iu[0], iu[1]]
    I, J = iu[0][keep], iu[1][keep]
    r = (x[:, I] - x[:, J]).norm(dim=-1)
    sr6 = (0.5 * (sg[I] + sg[J]) / r) ** 6
    krf = (1 / rc**3) * (t.rf_dielectric - 1) / (2 * t.rf_dielectric + 1)
    crf = (1 / rc) * 3 * t.rf_dielectric / (2 * t.rf_dielectric + 1)
    pair = 4 * torch.sqrt(ep[I] * ep[J]) * (sr6 * sr6 - sr6) + K_E * q[I] * q[J] * (1 / r + krf * r * r - crf)
    out["nb"] = (pair * (r < rc)).sum(-1)
    xp = T(t.exc_par)
    r = (x[:, ei[:, 0]] - x[:, ei[:, 1]]).norm(dim=-1)
    sr6 = (xp[:, 1] / r) ** 6
    out["nb14"] = (4 * xp[:, 2] * (sr6 * sr6 - sr6) + K_E * xp[:, 0] / r).sum(-1)
    rad, sc = T(t.atom_par[:, 3]), T(t.atom_par[:, 4])
    off = rad - 0.009
    eye = torch.eye(V, dtype=torch.bool)
    R = (x[:, :, None] - x[:, None]).norm(dim=-1) + eye.to(x.dtype)
    offi, srj = off[None, :, None], (off * sc)[None, None, :]
    l = 1 / torch.maximum(offi.expand_as(R), (R - srj).abs())
    u = 1 / (R + srj)
    term = l - u + 0.25 * R * (u * u - l * l) + 0.5 / R * torch.log(u / l) + 0.25 * srj * srj / R * (l * l - u * u)
    term = term + torch.where(offi < (srj - R), 2 * (1 / offi - l), torch.zeros_like(R))
    ok = (offi < R + srj) & ~eye[None] & (R <= rc)
    s = (term * ok).sum(-1) * 0.5 * off
    born = 1 / (1 / off - torch.tanh(s - 0.8 * s**2 + 4.85 * s**3) / rad)
    pre = -K_E * (1 / t.solute_dielectric - 1 / (eps_solvent or t.solvent_dielectric))
    e_sa = (4 * math.pi * (sa or t.surface_area_energy) * (rad + 0.14) ** 2 * (rad / born) ** 6).sum(-1)
    I, J = iu[0], iu[1]
    r = (x[:, I] - x[:, J]).norm(dim=-1)
    a2 = born[:, I] * born[:, J]
    qq = pre * q[I] * q[J]
    eg = (qq / torch.sqrt(r * r + a2 * torch.exp(-r * r / (4 * a2))) - qq / rc) * (r <= rc)
    out["gb"] = e_sa + (0.5 * pre * q * q / born).sum(-1) + eg.sum(-1)
    return out
