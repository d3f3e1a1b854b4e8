"""The amber99sb-ildn + OBC tables of timewarp_amd/forcefield.py against the SECOND OpenMM known-answer file the reference
holds: testdata/output/1hgv-traj-arrays.npz - 140 frames of a 46-residue protein (691 atoms; MET ... GLY, 18 residue types)
with potential energies and forces, written by the same OpenMM 7.4.1 run as the NNQQ file next to it (which the committed
tables meet at 2e-4 kJ/mol).  Build container only (reads /root/reference).

  python tools/pin_energy/pin_1hgv.py            residuals: energies, forces per atom type / residue / atom name
  python tools/pin_energy/pin_1hgv.py --fixture  writes tests/golden/energy_kat_1hgv.npz: topology + 12 of the ODD frames (the
                                                 fit of the ILE / LEU / ASP series, fit_ildn_1hgv.py, took the even ones) -
                                                 positions, energies, forces as OpenMM wrote them: data, for
                                                 tests/test_energy_kat.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from energy_t import energy_terms  # noqa: E402

from timewarp_amd import forcefield as ff  # noqa: E402

REF = "/root/reference/testdata/output"


def topology():
    names, res, rid = [], [], []
    for line in open(os.path.join(REF, "1hgv-traj-state0.pdb")):
        if line.startswith("ATOM"):
            names.append(line[12:16].strip()); res.append(line[17:20].strip()); rid.append(int(line[22:26]))
    return names, res, rid


def evaluate(t, pos, chunk=7):
    E, F, parts = [], [], []
    for i in range(0, len(pos), chunk):
        x = torch.tensor(pos[i:i + chunk], dtype=torch.float64, requires_grad=True)
        terms = energy_terms(x, t)
        e = sum(terms.values())
        F.append(-torch.autograd.grad(e.sum(), x)[0].numpy())
        E.append(e.detach().numpy())
        parts.append({k: v.detach().numpy() for k, v in terms.items()})
    return np.concatenate(E), np.concatenate(F), {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


def main():
    names, res, rid = topology()
    z = np.load(os.path.join(REF, "1hgv-traj-arrays.npz"))
    nf = int(os.environ.get("FRAMES", "14"))
    sel = np.linspace(0, len(z["positions"]) - 1, nf).astype(int)
    pos, eref, fref = z["positions"][sel].astype(np.float64), z["energies"][sel, 0], z["forces"][sel].astype(np.float64)
    if os.environ.get("NO_ILDN"):   # the tables as first written down: parm99 terms on the ILE / LEU / ASP side-chain bonds
        for r in ("ILE", "LEU", "ASP"):
            ff._ILDN_FITTED_TORSIONS.pop(r, None)
    t = ff.amber99sbildn_obc_tables(names, res, rid, improper_neighbour_order=os.environ.get("ORDER", "pyset"))
    print(f"improper neighbour order {os.environ.get('ORDER', 'pyset')}; ILDN series of ILE / LEU / ASP: "
          f"{'parm99 terms (as first written)' if os.environ.get('NO_ILDN') else 'fitted on the even frames'}")
    torch.set_num_threads(8)
    e, f, parts = evaluate(t, pos)
    de = e - eref
    print(f"frames {nf}: E - E_ref mean {de.mean():+.4f}  spread {de.std():.4f}  (E_ref {eref.mean():.1f});  "
          f"force rms residual {np.sqrt(((f - fref) ** 2).mean()):.4f} of rms {np.sqrt((fref ** 2).mean()):.1f}")
    print("terms (frame 0):", {k: round(float(v[0]), 3) for k, v in parts.items()})
    r = np.sqrt(((f - fref) ** 2).sum(-1))          # [frames, atoms]
    per_atom = np.sqrt((r ** 2).mean(0))
    by = {}
    for i, (n_, rs) in enumerate(zip(names, res)):
        key = ("N" if rid[i] == rid[0] else "C" if rid[i] == rid[-1] else "") + rs
        by.setdefault((key, n_), []).append(per_atom[i])
    rows = sorted(((np.sqrt(np.mean(np.square(v))), k) for k, v in by.items()), reverse=True)
    print("largest force residuals by (residue, atom):")
    for v, k in rows[:40]:
        print(f"   {k[0]:5s} {k[1]:5s} {v:10.3f}")
    byres = {}
    for (rs, n_), v in by.items():
        byres.setdefault(rs, []).extend(v)
    print("by residue type:", {k: round(float(np.sqrt(np.mean(np.square(v)))), 3) for k, v in sorted(byres.items())})


def fixture():
    names, res, rid = topology()
    z = np.load(os.path.join(REF, "1hgv-traj-arrays.npz"))
    sel = np.arange(1, 140, 12)
    assert len(sel) == 12 and (sel % 2 == 1).all()
    out = os.path.join(ROOT, "tests", "golden", "energy_kat_1hgv.npz")
    np.savez_compressed(out, atom_names=np.array(names), residue_names=np.array(res), residue_ids=np.array(rid, dtype=np.int32),
                        frames=sel.astype(np.int32), positions=z["positions"][sel], energies=z["energies"][sel, 0],
                        forces=z["forces"][sel], openmm_version=np.array("7.4.1"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    fixture() if "--fixture" in sys.argv else main()
