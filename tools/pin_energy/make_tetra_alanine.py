"""Fixture for the amber14 / GBSA-OBC I preset (parity unpinned): topology and a relaxed geometry of capped tetra-alanine
ACE-ALA4-NME (52 atoms).  No structure file of a 4AA peptide under that preset exists offline, so the geometry is made
here: atoms scattered along the chain, then the tables' own energy (bonds, angles, torsions, LJ + Coulomb; no GB) is
minimised with L-BFGS in float64 (tools/pin_energy/energy_t.py).  Chirality is whatever the minimiser finds - the
fixture serves arithmetic parity (HIP kernel == C oracle, MH iterations == oracle loop), not chemistry.

    python tools/pin_energy/make_tetra_alanine.py   ->   tests/golden/tetra_alanine_capped.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from energy_t import energy_terms  # noqa: E402
from timewarp_amd import forcefield as ff  # noqa: E402

names, res, rid = [], [], []
for r, i in [("ACE", 1)] + [("ALA", 2 + k) for k in range(4)] + [("NME", 6)]:
    for n in ff._RESIDUES_FF14SB[r]["names"]:
        names.append(n); res.append(r); rid.append(i)
t = ff.amber14_obc1_tables(names, res, rid)
g = torch.Generator().manual_seed(7)
x = torch.zeros(1, len(names), 3, dtype=torch.float64)
for a, r in enumerate(rid):
    x[0, a] = torch.tensor([0.36 * r, 0.0, 0.0], dtype=torch.float64) + 0.12 * torch.randn(3, generator=g, dtype=torch.float64)
x.requires_grad_(True)


def total(xx, soft):
    e = energy_terms(xx, t)
    return e["bond"] + e["angle"] + e["torsion"] + (0.0 if soft else e["nb"] + e["nb14"])


for soft, iters in ((True, 200), (False, 400)):
    opt = torch.optim.LBFGS([x], lr=0.5, max_iter=iters, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        e = total(x, soft).sum()
        e.backward()
        return e
    opt.step(closure)
e = energy_terms(x.detach(), t)
print({k: float(v) for k, v in e.items()})
b = t.bond_idx
d = (x.detach()[0, b[:, 0]] - x.detach()[0, b[:, 1]]).norm(dim=-1)
print("bond lengths nm: min %.3f max %.3f; min nonbonded distance %.3f" % (
    d.min(), d.max(), float(torch.cdist(x.detach()[0], x.detach()[0]).fill_diagonal_(9).min())))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tetra_alanine_capped.npz"), atom_names=np.array(names),
                    residue_names=np.array(res), residue_ids=np.array(rid), positions=x.detach()[0].numpy().astype(np.float32))
