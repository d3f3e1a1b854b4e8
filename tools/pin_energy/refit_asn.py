"""The two asparagine side-chain torsion series of the amber99sb-ildn tables, against ALL OpenMM data the reference holds
for its test peptide NNQQ - not only the 40-frame file they were fitted to in r02:

    simulation/testdata/implicit-2olx-traj-cpu-arrays.npz   40 frames   (tests/golden/energy_kat_2olx.npz; simulation/tests/test_md.py)
    simulation/testdata/implicit-2olx-traj-arrays.npz       200 frames  an independent trajectory of the same System
    testdata/output/2olx-traj-arrays.npz                    140 frames  a third one (1 ns later), visits other chi1 rotamers
    testdata/smallest_molecule/2olx-traj-arrays.npz         2 frames

Each holds positions, potential energies and forces written by OpenMM.  r03's review: the series were checked against the
data they were fitted to.  This script (build container only: it reads /root/reference)
  1. evaluates the COMMITTED tables on the three files that were never used: an independent check;
  2. refits the series on a training split over all files (forces only, linear least squares, cos + sin up to n = 6, on
     candidate carrier dihedrals) and reports energies and forces on the held-out split;
  3. writes tests/golden/energy_kat_2olx_more.npz: held-out frames as a data fixture for tests/test_energy_kat.py.
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
REF = "/root/reference"
FILES = {"cpu40": "simulation/testdata/implicit-2olx-traj-cpu-arrays.npz", "impl200": "simulation/testdata/implicit-2olx-traj-arrays.npz",
         "out140": "testdata/output/2olx-traj-arrays.npz", "small2": "testdata/smallest_molecule/2olx-traj-arrays.npz"}
g = np.load(os.path.join(ROOT, "tests", "golden", "energy_kat_2olx.npz"))
NAMES, RID = list(g["atom_names"]), list(g["residue_ids"])
IDX = {(r, n): i for i, (n, r) in enumerate(zip(NAMES, RID))}


def load():
    pos, en, frc, src = [], [], [], []
    for k, f in FILES.items():
        z = np.load(os.path.join(REF, f))
        pos.append(z["positions"]); en.append(z["energies"][:, 0]); frc.append(z["forces"]); src += [k] * len(z["positions"])
    return np.concatenate(pos), np.concatenate(en), np.concatenate(frc).astype(np.float64), np.array(src)


POS, EREF, FREF, SRC = load()
X = torch.tensor(POS, dtype=torch.float64, requires_grad=True)


def force(e):
    return -torch.autograd.grad(e.sum(), X, retain_graph=True)[0].numpy()


def total(t):
    e = sum(energy_terms(X, t).values())
    return e.detach().numpy(), force(e)


def without_fitted(t):
    quads = {tuple(IDX[(r, n)] for n in q) for r in (1, 2) for q in ff._ASN_FITTED_TORSIONS}
    keep = np.array([tuple(q) not in quads and tuple(q[::-1]) not in quads for q in t.torsion_idx])
    return dataclasses.replace(t, torsion_idx=t.torsion_idx[keep], torsion_par=t.torsion_par[keep])


def report(tag, e, f, sel):
    d, df = (e - EREF)[sel], (f - FREF)[sel]
    print(f"   {tag:34s} n={sel.sum():3d}  dE mean {d.mean():+8.4f} std {d.std():8.4f} max|dE - mean| {np.abs(d - d.mean()).max():8.4f} | "
          f"F rms diff {np.sqrt((df ** 2).mean()):8.4f} max {np.abs(df).max():8.3f}")


def basis(spec, sin=True):
    """columns of the design matrix: forces and energies of cos(n phi), sin(n phi) summed over both ASN"""
    fc, ec, lab = [], [], []
    for quad, ns in spec.items():
        ph = dihedral(X, torch.tensor([[IDX[(r, n)] for n in quad] for r in (1, 2)]))
        for n in ns:
            for fn, nm in ((torch.cos, "cos"),) + (((torch.sin, "sin"),) if sin else ()):
                e = fn(n * ph).sum(-1)
                fc.append(force(e)); ec.append(e.detach().numpy()); lab.append(("-".join(quad), nm, n))
    return np.stack(fc, -1), np.stack(ec, -1), lab


def main():
    tab = ff.amber99sbildn_obc_tables(NAMES, list(g["residue_names"]), RID)
    print(f"{len(POS)} frames: " + ", ".join(f"{k} {np.sum(SRC == k)}" for k in FILES))
    print("1. committed tables, per file (cpu40 is what the series were fitted to; the others are independent)")
    e, f = total(tab)
    for k in FILES:
        report(k, e, f, SRC == k)
    t0 = without_fitted(tab)
    e0, f0 = total(t0)
    print("   ... and without the fitted series:")
    for k in FILES:
        report(k + " (no ASN series)", e0, f0, SRC == k)
    # training / held-out split: every second frame of each file
    idx = np.arange(len(POS))
    train = np.zeros(len(POS), bool)
    for k in FILES:
        m = np.where(SRC == k)[0]
        train[m[::2]] = True
    test = ~train
    print(f"2. refit on {train.sum()} training frames (forces only), evaluated on {test.sum()} held-out frames")
    results = {}
    for name, spec in {
        "r02 carriers C-CA-CB-CG + CA-CB-CG-ND2": {("C", "CA", "CB", "CG"): range(1, 7), ("CA", "CB", "CG", "ND2"): range(1, 7)},
        "N-CA-CB-CG + CA-CB-CG-OD1": {("N", "CA", "CB", "CG"): range(1, 7), ("CA", "CB", "CG", "OD1"): range(1, 7)},
        "N-CA-CB-CG + CA-CB-CG-ND2": {("N", "CA", "CB", "CG"): range(1, 7), ("CA", "CB", "CG", "ND2"): range(1, 7)},
        "all four": {("N", "CA", "CB", "CG"): range(1, 7), ("C", "CA", "CB", "CG"): range(1, 7), ("CA", "CB", "CG", "OD1"): range(1, 7),
                     ("CA", "CB", "CG", "ND2"): range(1, 7)},
    }.items():
        Fc, Ec, lab = basis(spec)
        A = Fc[train].reshape(-1, Fc.shape[-1])
        b = (FREF - f0)[train].ravel()
        c, *_ = np.linalg.lstsq(A, b, rcond=None)
        e1, f1 = e0 + Ec @ c, f0 + Fc @ c
        const = -(e1 - EREF)[train].mean()
        print(f"   -- {name}")
        report("train", e1 + const, f1, train)
        report("HELD OUT", e1 + const, f1, test)
        results[name] = (c, lab, const)
    return tab, results, test


if __name__ == "__main__" and "--final" not in sys.argv:
    main()


# ---------------------------------------------------------------------------------------------------------------------
# r04 follow-up.  The residual the refit above leaves on testdata/output/2olx-traj-arrays.npz and testdata/smallest_molecule
# (force rms 3.3, max 220, all of it on CA / C / O / OXT of the C-terminal residue) is NOT a torsion series: those two files
# were written by an older OpenMM (7.4.1, their PDB headers), which matches the carboxylate improper X-O2-C-O2 as
# (CA, OXT, C, O) where 7.6 / 7.7 - the version the reference pins, and the one that wrote the two simulation/testdata files -
# gives (CA, O, C, OXT).  With the ordering of the file's own OpenMM version, k = 10.5000 kcal/mol comes out of either group
# and every force of all 382 frames is reproduced to 0.002 - 0.004 kJ/mol/nm rms (`final()`).
def swap_carboxylate_improper(t):
    C, CA, O, OXT = (IDX[(4, n)] for n in ("C", "CA", "O", "OXT"))
    idx = t.torsion_idx.copy()
    for i, q in enumerate(idx.tolist()):
        if q[2] == C and set(q) == {C, CA, O, OXT}:
            idx[i] = (CA, OXT, C, O)
    return dataclasses.replace(t, torsion_idx=idx)


OLD_OPENMM = ("out140", "small2")


def final(write_fixture=True):
    tab = ff.amber99sbildn_obc_tables(NAMES, list(g["residue_names"]), RID)
    t0 = without_fitted(tab)
    old = np.isin(SRC, OLD_OPENMM)
    eA, fA = total(t0)
    eB, fB = total(swap_carboxylate_improper(t0))
    e0, f0 = np.where(old, eB, eA), np.where(old[:, None, None], fB, fA)
    train = np.zeros(len(POS), bool)
    for k in FILES:
        train[np.where(SRC == k)[0][::2]] = True
    held = ~train
    spec = {("C", "CA", "CB", "CG"): range(1, 7), ("CA", "CB", "CG", "ND2"): range(1, 7)}
    Fc, Ec, lab = basis(spec)
    c, *_ = np.linalg.lstsq(Fc[train].reshape(-1, Fc.shape[-1]), (FREF - f0)[train].ravel(), rcond=None)
    e1, f1 = e0 + Ec @ c, f0 + Fc @ c
    const = -(e1 - EREF)[train].mean()
    print(f"3. final refit: 24 Fourier coefficients on {train.sum()} training frames (forces only) + one additive constant "
          f"({const:+.4f} kJ/mol from the training energies); the carboxylate improper in the ordering of each file's OpenMM version")
    for k in FILES:
        report(k + " train", e1 + const, f1, train & (SRC == k))
        report(k + " HELD OUT", e1 + const, f1, held & (SRC == k))
    # as PeriodicTorsionForce terms k (1 + cos(n phi - phase)), kcal/mol and degrees; the constants k of those terms are part of
    # what `const` has to make up for: E = sum k (1 + cos) + c0  with  c0 = const - sum k
    terms, ksum = {}, 0.0
    for (quad, _, n), a, b in zip(lab[0::2], c[0::2], c[1::2]):
        k = float(np.hypot(a, b))
        terms.setdefault(tuple(quad.split("-")), []).append((round(k / KCAL, 5), round(float(np.degrees(np.arctan2(b, a))), 3), int(n)))
        ksum += 2.0 * k   # two asparagines
    c0 = (const - ksum) / KCAL
    print("   _ASN_FITTED_TORSIONS = {")
    for q, v in terms.items():
        print(f"       {q!r}: {v},")
    print("   }")
    print(f"   additive constant needed beside those terms: {c0:+.5f} kcal/mol for the molecule (r02's local form needed one; "
          "with the full series the absolute energies follow from the k (1 + cos) form itself)")
    if write_fixture:
        sel = np.zeros(len(POS), bool)
        for k in FILES:
            if k != "cpu40":
                m = np.where(SRC == k)[0]
                sel[m[1::4]] = True
        sel &= held
        out = os.path.join(ROOT, "tests", "golden", "energy_kat_2olx_more.npz")
        np.savez_compressed(out, positions=POS[sel].astype(np.float32), energies=EREF[sel], forces=FREF[sel].astype(np.float32),
                            old_openmm=np.isin(SRC[sel], OLD_OPENMM), source=SRC[sel])
        print(f"   wrote {out}: {sel.sum()} held-out frames ({', '.join(f'{k} {np.sum(SRC[sel] == k)}' for k in FILES if k != 'cpu40')})")
    return terms, c0


if __name__ == "__main__" and "--final" in sys.argv:
    final()
