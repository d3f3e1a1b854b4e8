"""How the energy oracle / parameter tables were pinned against the reference's OpenMM known-answer file
(tests/golden/energy_kat_2olx.npz = simulation/testdata/implicit-2olx-traj-cpu-arrays.npz + its PDB's atom names).

    python tools/pin_energy/fit_2olx.py            # all checks below

1. residual of the committed tables (energies, forces, worst atoms);
2. the two asparagine torsion series refitted as free cosine coefficients (linear least squares on the forces) --
   reproduces `_ASN_FITTED_TORSIONS`, and shows that sine terms are not needed;
3. GBSA-OBC radii / scale factors refitted per (element, bonds) class (non-linear least squares on the forces) --
   recovers 0.115/0.125/0.148/0.1625/0.17063/0.1875/0.19 nm and 0.85/0.72/0.79/0.85;
4. the amide-nitrogen improper constant refitted per class -- 1.1 kcal/mol for the backbone pattern, 1.0 for NH2;
5. the solvent dielectric scanned -- minimum at 78.5.
"""
import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from energy_t import dihedral, energy_terms  # noqa: E402

from timewarp_amd import forcefield as ff  # noqa: E402

KCAL = 4.184
z = np.load(os.path.join(ROOT, "tests", "golden", "energy_kat_2olx.npz"))
NAMES, RID = list(z["atom_names"]), list(z["residue_ids"])
TAB = ff.amber99sbildn_obc_tables(NAMES, list(z["residue_names"]), RID)
EREF, FREF = z["energies"][:, 0], z["forces"].astype(np.float64)
X = torch.tensor(z["positions"], dtype=torch.float64, requires_grad=True)
IDX = {(r, n): i for i, (n, r) in enumerate(zip(NAMES, RID))}


def force(e):
    return -torch.autograd.grad(e.sum(), X, retain_graph=True)[0].numpy()


def total(t, **kw):
    e = sum(energy_terms(X, t, **kw).values())
    return e.detach().numpy(), force(e)


def without_fitted(t):
    quads = {tuple(IDX[(r, n)] for n in q) for r in (1, 2) for q in ff._ASN_FITTED_TORSIONS}
    keep = np.array([tuple(q) not in quads and tuple(q[::-1]) not in quads for q in t.torsion_idx])
    return dataclasses.replace(t, torsion_idx=t.torsion_idx[keep], torsion_par=t.torsion_par[keep])


def report(tag, e, f):
    d, df = e - EREF, f - FREF
    pa = np.sqrt((df**2).sum(-1).mean(0))
    worst = [(NAMES[i], int(RID[i]), round(float(pa[i]), 3)) for i in np.argsort(-pa)[:4]]
    print(f"{tag}: dE mean {d.mean():.4f} std {d.std():.5f} | F rms diff {np.sqrt((df**2).mean()):.4f} (|F| rms "
          f"{np.sqrt((FREF**2).mean()):.0f}) worst atoms {worst}")


def fit_series(t0, spec, sin=False, **kw):
    """spec {atom-name quad: periodicities}; returns coefficients (kcal/mol) of cos (and sin) terms, shared by both ASN"""
    e0, f0 = total(t0, **kw)
    cols, ecols, labels = [], [], []
    for quad, ns in spec.items():
        ph = dihedral(X, torch.tensor([[IDX[(r, n)] for n in quad] for r in (1, 2)]))
        for n in ns:
            for fn, nm in ((torch.cos, "cos"),) + (((torch.sin, "sin"),) if sin else ()):
                e = fn(n * ph).sum(-1)
                cols.append(force(e).ravel()); ecols.append(e.detach().numpy()); labels.append(("-".join(quad), nm, n))
    A, Ec = np.stack(cols, 1), np.stack(ecols, 1)
    b = (FREF - f0).ravel()
    c, *_ = np.linalg.lstsq(A, b, rcond=None)
    d = e0 + Ec @ c - EREF
    print(f"   force rms after fit {np.sqrt(((A @ c - b)**2).mean()):.4f}; dE std {d.std():.5f}; constant needed "
          f"{-d.mean() / KCAL:.4f} kcal vs 2 x sum|k| = {2 * np.abs(c).sum() / KCAL:.4f}")
    for l, v in zip(labels, c):
        print("   ", l, f"{v / KCAL:+.5f} kcal/mol")
    return c


def main():
    print("1. committed tables")
    report("   all terms", *total(TAB))
    t0 = without_fitted(TAB)
    report("   without the fitted ASN series", *total(t0))
    print("2. refit of the ASN series (cos only), then with sine terms")
    spec = {q: [n for _, _, n in v] for q, v in ff._ASN_FITTED_TORSIONS.items()}
    fit_series(t0, spec)
    fit_series(t0, {("CA", "CB", "CG", "ND2"): [1, 2, 3, 4, 5, 6], ("C", "CA", "CB", "CG"): [1, 5]}, sin=True)
    print("3. GBSA-OBC radii per class / scales per element refitted")
    from scipy.optimize import least_squares

    rad0 = TAB.atom_par[:, 3]
    classes = sorted(set(np.round(rad0, 5)))
    ci = np.array([classes.index(round(r, 5)) for r in rad0])
    ei = np.array(["HCNO".index(n[0]) for n in NAMES])
    other = sum(v for k, v in energy_terms(X, TAB).items() if k != "gb")
    f_other = force(other)

    def resid(p):
        ap = TAB.atom_par.copy()
        ap[:, 3], ap[:, 4] = p[:len(classes)][ci], p[len(classes):][ei]
        return (force(energy_terms(X, dataclasses.replace(TAB, atom_par=ap))["gb"]) + f_other - FREF).ravel()

    sol = least_squares(resid, np.array(classes + [0.8, 0.8, 0.8, 0.8]) * 1.03, diff_step=1e-5)
    print("   start (all +3 %, scales 0.82):", [round(c * 1.03, 5) for c in classes])
    print("   fitted radii", np.round(sol.x[:len(classes)], 5), "scales H C N O", np.round(sol.x[len(classes):], 4),
          "rms %.4f" % np.sqrt((sol.fun**2).mean()))
    print("4. amide-N improper constants (kcal/mol) refitted per class")
    bonded = {tuple(sorted(b)) for b in TAB.bond_idx.tolist()}
    is_imp = np.array([all(tuple(sorted((int(q[2]), int(a)))) in bonded for a in (q[0], q[1], q[3])) and
                       NAMES[q[2]] in ("N", "ND2", "NE2") for q in TAB.torsion_idx])
    t_no = dataclasses.replace(TAB, torsion_idx=TAB.torsion_idx[~is_imp], torsion_par=TAB.torsion_par[~is_imp])
    _, f0 = total(t_no)
    cols = []
    for cls in (("N",), ("ND2", "NE2")):
        quads = [q for q, m in zip(TAB.torsion_idx, is_imp) if m and NAMES[q[2]] in cls]
        ph = dihedral(X, torch.tensor(np.array(quads)))
        cols.append(force((1 + torch.cos(2 * ph - np.pi)).sum(-1)).ravel())
    c, *_ = np.linalg.lstsq(np.stack(cols, 1), (FREF - f0).ravel(), rcond=None)
    print("   backbone N-H: %.4f   side-chain NH2: %.4f" % tuple(c / KCAL))
    print("5. solvent dielectric scan (force rms, dE std)")
    for eps in (78.0, 78.2, 78.3, 78.4, 78.5, 79.0):
        e, f = total(TAB, eps_solvent=eps)
        print("   %.1f: %.4f %.5f" % (eps, np.sqrt(((f - FREF)**2).mean()), (e - EREF).std()))


if __name__ == "__main__":
    main()
