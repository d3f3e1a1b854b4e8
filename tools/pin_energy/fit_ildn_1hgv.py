"""The side-chain torsion series ff99SB-ILDN replaces for ILE, LEU and ASP - not recallable, like asparagine's - fitted to
the forces of the reference's 691-atom OpenMM file (testdata/output/1hgv-traj-arrays.npz: 3 ILE, 4 LEU, 1 ASP, 140 frames).
Everything else in the tables meets that file at its float32 noise (tools/pin_energy/pin_1hgv.py), so the residual force is
the difference between the true series and the parm99 terms the tables carry on those bonds.  Linear least squares on the
forces of every second frame: cos + sin coefficients up to n = NMAX on candidate carrier dihedrals; the other frames are
held out.  Build container only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from energy_t import dihedral  # noqa: E402
from pin_1hgv import REF, evaluate, topology  # noqa: E402

from timewarp_amd import forcefield as ff  # noqa: E402

CARRIERS = {
    "ILE": [("N", "CA", "CB", "CG1"), ("N", "CA", "CB", "CG2"), ("C", "CA", "CB", "CG1"), ("C", "CA", "CB", "CG2"),
            ("CA", "CB", "CG1", "CD1"), ("CG2", "CB", "CG1", "CD1")],
    "LEU": [("N", "CA", "CB", "CG"), ("C", "CA", "CB", "CG"), ("CA", "CB", "CG", "CD1"), ("CA", "CB", "CG", "CD2")],
    "ASP": [("N", "CA", "CB", "CG"), ("C", "CA", "CB", "CG"), ("CA", "CB", "CG", "OD1"), ("CA", "CB", "CG", "OD2")],
}
NMAX = int(os.environ.get("NMAX", "4"))


def main():
    names, res, rid = topology()
    z = np.load(os.path.join(REF, "1hgv-traj-arrays.npz"))
    pos, eref, fref = z["positions"].astype(np.float64), z["energies"][:, 0], z["forces"].astype(np.float64)
    t = ff.amber99sbildn_obc_tables(names, res, rid, improper_neighbour_order="pyset")
    cache = "/tmp/pin_1hgv_base.npz"
    if os.path.exists(cache):
        c = np.load(cache); e, f = c["e"], c["f"]
    else:
        torch.set_num_threads(8)
        e, f, _ = evaluate(t, pos)
        np.savez(cache, e=e, f=f)
    R = fref - f                      # what the missing series must supply
    dE = eref - e
    idx = {(r, n): i for i, (n, r) in enumerate(zip(names, rid))}
    only = os.environ.get("ONLY")
    carriers = {k: v for k, v in CARRIERS.items() if not only or k in only.split(",")}
    if os.environ.get("PICK"):   # e.g. PICK="ILE:0,4;LEU:1,2"
        for item in os.environ["PICK"].split(";"):
            k, ids = item.split(":")
            carriers[k] = [CARRIERS[k][int(i)] for i in ids.split(",")]
    cols, labels = [], []
    X = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    ecols = []
    for rs, quads in carriers.items():
        rids = sorted({r for n, r_, r in zip(names, res, rid) if r_ == rs})
        for q in quads:
            ii = torch.tensor([[idx[(r, a)] for a in q] for r in rids])
            phi = dihedral(X, ii)         # [F, instances]
            for n in range(1, NMAX + 1):
                for fn, tag in ((torch.cos, "cos"), (torch.sin, "sin")):
                    b = fn(n * phi).sum(-1)
                    g = torch.autograd.grad(b.sum(), X, retain_graph=True)[0]
                    cols.append((-g).numpy().reshape(len(pos), -1))
                    ecols.append(b.detach().numpy())
                    labels.append((rs, q, n, tag))
    A = np.stack(cols, -1)               # [F, 3V, K]
    Eb = np.stack(ecols, -1)             # [F, K]
    train = np.arange(0, len(pos), 2)
    test = np.arange(1, len(pos), 2)
    At = A[train].reshape(-1, A.shape[-1])
    coef, *_ = np.linalg.lstsq(At, R[train].reshape(-1), rcond=None)
    for split, name in ((train, "train"), (test, "held out")):
        res_f = R[split].reshape(len(split), -1) - A[split] @ coef
        de = dE[split] - Eb[split] @ coef
        print(f"{name}: force residual rms {np.sqrt((res_f ** 2).mean()):.4f} (before {np.sqrt((R[split] ** 2).mean()):.4f});  "
              f"E_ref - E: mean {de.mean():+.4f} spread {de.std():.4f} (before {dE[split].mean():+.4f} / {dE[split].std():.4f})")
    res_f = (R.reshape(len(pos), -1) - A @ coef).reshape(len(pos), -1, 3)
    per_atom = np.sqrt((res_f ** 2).sum(-1).mean(0))
    worst = np.argsort(-per_atom)[:10]
    print("largest remaining per-atom residuals:", [(res[i], names[i], round(float(per_atom[i]), 3)) for i in worst])
    k = 0
    for rs, quads in carriers.items():
        for q in quads:
            terms = []
            for n in range(1, NMAX + 1):
                c, s_ = coef[k], coef[k + 1]; k += 2
                amp, ph = np.hypot(c, s_), np.degrees(np.arctan2(s_, c))
                terms.append(f"n{n}: {amp / 4.184:.5f} kcal @ {ph:+.3f}")
            print(rs, "-".join(q), " | ".join(terms))
    np.savez("/tmp/fit_ildn_coef.npz", coef=coef, labels=np.array([str(l) for l in labels]))


if __name__ == "__main__":
    main()
