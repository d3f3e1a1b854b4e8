"""Force-field parameter tables for the HIP energy kernel (`tw_forcefield`, include/timewarp_hip.h).

The reference never holds these numbers itself: `simulation/md.py:150-173` asks OpenMM's
`ForceField("amber99sbildn.xml", "amber99_obc.xml").createSystem(...)` for them.  Two sources here:

* `tables_from_openmm_system(system)` -- reads the tables out of an `openmm.System` exactly as the
  scripts build it (the drop-in route; needs OpenMM importable, which it is not in this image);
* `amber99sbildn_obc_tables(...)` / `tables_from_pdb(path)` / `alanine_dipeptide_amber99sb()` -- the published
  parm99 / ff99SB / ff94-charge / OBC numbers for a small residue set (ACE, ALA, NME, ASN, GLN), written out here so
  the whole MH path runs without OpenMM.  Pinned against the reference's own OpenMM known-answer file (40 frames of
  NNQQ with energies and forces, simulation/tests/test_md.py:35-83); see the section comment below for what that
  covers and for the two asparagine torsion series that had to be fitted.

Units: nm, kJ/mol, elementary charge, radians.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from itertools import combinations
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from ._lib import ForceField

KCAL = 4.184


@dataclass
class ForceFieldTables:
    bond_idx: np.ndarray      # [nb,2] int32
    bond_par: np.ndarray      # [nb,2] (r0, k)
    angle_idx: np.ndarray     # [na,3]
    angle_par: np.ndarray     # [na,2] (theta0, k)
    torsion_idx: np.ndarray   # [nt,4]
    torsion_par: np.ndarray   # [nt,3] (n, phase, k)
    exc_idx: np.ndarray       # [ne,2]
    exc_par: np.ndarray       # [ne,3] (qq, sigma, eps)
    atom_par: np.ndarray      # [V,5] (q, sigma, eps, gb_radius, gb_scale)
    has_gbsa: int = 1  # 0: none, 1: GBSA-OBC II (GBSAOBCForce / amber99_obc.xml), 2: GBSA-OBC I (implicit/obc1.xml)
    cutoff: float = 2.0
    rf_dielectric: float = 1.0  # OpenMM sets the reaction-field dielectric to 1 when a GB force is present
    solute_dielectric: float = 1.0
    solvent_dielectric: float = 78.5
    surface_area_energy: float = 2.25936

    @property
    def n_atoms(self) -> int:
        return int(self.atom_par.shape[0])

    def to_device(self, device) -> "DeviceForceField":
        return DeviceForceField(self, device)


class DeviceForceField:
    """Device copies of the tables + the ctypes struct pointing at them."""

    def __init__(self, t: ForceFieldTables, device):
        def i32(a, w):
            return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32).reshape(-1, w), device=device)

        def f64(a, w):
            return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64).reshape(-1, w), device=device)

        self.tables = t
        self.keep = dict(
            bond_idx=i32(t.bond_idx, 2), bond_par=f64(t.bond_par, 2), angle_idx=i32(t.angle_idx, 3),
            angle_par=f64(t.angle_par, 2), torsion_idx=i32(t.torsion_idx, 4), torsion_par=f64(t.torsion_par, 3),
            exc_idx=i32(t.exc_idx, 2), exc_par=f64(t.exc_par, 3), atom_par=f64(t.atom_par, 5),
        )
        k = self.keep
        self.struct = ForceField(
            t.n_atoms, k["bond_idx"].shape[0], k["angle_idx"].shape[0], k["torsion_idx"].shape[0],
            k["exc_idx"].shape[0], int(t.has_gbsa), t.cutoff, t.rf_dielectric, t.solute_dielectric,
            t.solvent_dielectric, t.surface_area_energy,
            k["bond_idx"].data_ptr(), k["bond_par"].data_ptr(), k["angle_idx"].data_ptr(), k["angle_par"].data_ptr(),
            k["torsion_idx"].data_ptr(), k["torsion_par"].data_ptr(), k["exc_idx"].data_ptr(), k["exc_par"].data_ptr(),
            k["atom_par"].data_ptr(),
        )


# ---------------------------------------------------------------------------------------------------
# amber99sb-ildn + GBSA-OBC parameters for a small residue set (no OpenMM needed)
#
# Pinned against the reference's OpenMM known-answer file simulation/testdata/implicit-2olx-traj-cpu-arrays.npz
# (40 frames of the peptide NNQQ, E_pot and forces; simulation/tests/test_md.py:35-83): with the numbers below the
# energies agree to 1e-3 kJ/mol (of -1690) and the forces to 0.01 kJ/mol/nm rms (of 933), the float32 noise of
# the file (tests/test_energy_kat.py, tools/pin_energy/).  What that pins: every bond/angle/torsion/improper/LJ/charge
# entry below that NNQQ exercises (all the types alanine dipeptide uses among them), the GBSA-OBC radii rule, the
# improper atom ordering, the dielectric constants.  Two torsion series could not be recalled and were FITTED to the
# file (marked below); alanine dipeptide does not use them.
# ---------------------------------------------------------------------------------------------------
# parm99: Rmin/2 (Angstrom), eps (kcal/mol)
_LJ = {"H": (0.6000, 0.0157), "HC": (1.4870, 0.0157), "H1": (1.3870, 0.0157), "HP": (1.1000, 0.0157),
       "CT": (1.9080, 0.1094), "C": (1.9080, 0.0860), "N": (1.8240, 0.1700), "N3": (1.8240, 0.1700),
       "O": (1.6612, 0.2100), "O2": (1.6612, 0.2100)}
# parm99 bonds: k (kcal/mol/A^2, E = k (r-r0)^2), r0 (A)
_BOND = {("CT", "HC"): (340.0, 1.090), ("CT", "H1"): (340.0, 1.090), ("CT", "HP"): (340.0, 1.090),
         ("C", "CT"): (317.0, 1.522), ("C", "O"): (570.0, 1.229), ("C", "N"): (490.0, 1.335),
         ("H", "N"): (434.0, 1.010), ("CT", "N"): (337.0, 1.449), ("CT", "CT"): (310.0, 1.526),
         ("H", "N3"): (434.0, 1.010), ("CT", "N3"): (367.0, 1.471), ("C", "O2"): (656.0, 1.250)}
# parm99 angles: k (kcal/mol/rad^2, E = k (t-t0)^2), theta0 (deg); keyed (a, centre, c) with a <= c
_ANGLE = {("HC", "CT", "HC"): (35.0, 109.50), ("H1", "CT", "H1"): (35.0, 109.50), ("HP", "CT", "HP"): (35.0, 109.50),
          ("CT", "CT", "HC"): (50.0, 109.50), ("CT", "CT", "H1"): (50.0, 109.50), ("CT", "CT", "HP"): (50.0, 109.50),
          ("C", "CT", "HC"): (50.0, 109.50), ("C", "CT", "H1"): (50.0, 109.50), ("C", "CT", "HP"): (50.0, 109.50),
          ("CT", "CT", "CT"): (40.0, 109.50), ("C", "CT", "CT"): (63.0, 111.10),
          ("CT", "C", "O"): (80.0, 120.40), ("CT", "C", "N"): (70.0, 116.60), ("N", "C", "O"): (80.0, 122.90),
          ("CT", "C", "O2"): (70.0, 117.00), ("O2", "C", "O2"): (80.0, 126.00),
          ("C", "N", "H"): (50.0, 120.00), ("C", "N", "CT"): (50.0, 121.90), ("CT", "N", "H"): (50.0, 118.04),
          ("H", "N", "H"): (35.0, 120.00),
          ("H1", "CT", "N"): (50.0, 109.50), ("CT", "CT", "N"): (80.0, 109.70), ("C", "CT", "N"): (63.0, 110.10),
          ("H", "N3", "H"): (35.0, 109.50), ("CT", "N3", "H"): (50.0, 109.50), ("CT", "CT", "N3"): (80.0, 111.20),
          ("C", "CT", "N3"): (80.0, 111.20), ("HP", "CT", "N3"): (50.0, 109.50)}
# parm99 + ff99SB propers: list of (k kcal/mol, phase deg, n); generic entries are keyed by the central pair
_TORSION_SPECIFIC = {
    ("C", "N", "CT", "C"): [(0.42, 0.0, 3), (0.27, 0.0, 2)],                        # phi
    ("N", "CT", "C", "N"): [(0.55, 180.0, 3), (1.58, 180.0, 2), (0.45, 180.0, 1)],  # psi
    ("C", "N", "CT", "CT"): [(0.40, 0.0, 3), (2.00, 0.0, 2), (2.00, 0.0, 1)],       # phi'
    ("CT", "CT", "C", "N"): [(0.40, 0.0, 3), (0.20, 0.0, 2), (0.20, 0.0, 1)],       # psi'
    ("H", "N", "C", "O"): [(2.50, 180.0, 2), (2.00, 0.0, 1)],
    ("HC", "CT", "C", "O"): [(0.80, 0.0, 1), (0.08, 180.0, 3)],
    ("H1", "CT", "C", "O"): [(0.80, 0.0, 1), (0.08, 180.0, 3)],
    ("CT", "CT", "CT", "CT"): [(0.18, 0.0, 3), (0.25, 180.0, 2), (0.20, 180.0, 1)],
    ("HC", "CT", "CT", "HC"): [(0.15, 0.0, 3)],
    ("CT", "CT", "CT", "HC"): [(0.16, 0.0, 3)],
}
_TORSION_GENERIC = {
    ("C", "N"): [(2.50, 180.0, 2)],            # X-C-N-X   10.0 / 4 paths
    ("CT", "CT"): [(1.40 / 9.0, 0.0, 3)],      # X-CT-CT-X
    ("C", "CT"): [],                           # X-C-CT-X  0.0
    ("CT", "N"): [],                           # X-CT-N-X  0.0
    ("CT", "N3"): [(1.40 / 9.0, 0.0, 3)],      # X-CT-N3-X
}
# Side-chain torsions of asparagine that ff99SB-ILDN replaces, by atom names.  FITTED to the reference's OpenMM data (the
# published ILDN series are not available offline).  r02 fitted them to the 40-frame known-answer file alone, where
# C-CA-CB-CG only visits 180 +- 35 degrees.  r04 (tools/pin_energy/refit_asn.py): the reference holds three more files of the
# same peptide written by OpenMM - 200 + 140 + 2 frames of positions / energies / forces, other trajectories that visit the
# other chi1 rotamers - and the fit now runs over all 382 frames: twelve cosine + twelve sine coefficients on the forces of
# every second frame.  All phases come out within 0.04 degrees of 0 / 180 (kept as fitted: rounding them costs a factor of
# seven on the held-out energies - the carrier dihedral is presumably not the one ILDN uses), CA-CB-CG-ND2 reproduces the r02
# numbers, and the HELD-OUT frames of all four files are met to 0.0002 kJ/mol (spread of the energy differences) and 0.0017
# kJ/mol/nm rms (forces) - the float32 noise of the files.  No additive constant is needed any more: with the terms in
# PeriodicTorsionForce's form k (1 + cos(n phi - phase)) the ABSOLUTE energies of all files come out to 1e-3 kJ/mol (r02's
# local form needed one).
_ASN_FITTED_TORSIONS = {
    ("C", "CA", "CB", "CG"): [(0.57089, -0.04, 1), (0.59589, -179.975, 2), (0.11831, 0.015, 3), (0.41723, 179.99, 4),
                              (0.10421, 0.01, 5), (0.10072, 179.999, 6)],
    ("CA", "CB", "CG", "ND2"): [(1.04630, 180.0, 1), (0.18100, 180.0, 2), (0.03540, 179.997, 3), (0.10030, 0.0, 4),
                                (0.12980, 0.0, 5), (0.10605, 180.0, 6)],
}
# GBSAOBCForce parameters of amber99_obc.xml: radius (nm) by element and number of bonded atoms, scale by element
_GB_SCALE = {"H": 0.85, "C": 0.72, "N": 0.79, "O": 0.85}
ELEMENT_MASSES = {"C": 12.01, "H": 1.008, "N": 14.01, "O": 16.0}
AD_MASSES = ELEMENT_MASSES


def _gb_radius(element: str, n_bonds: int, partner_element: str) -> float:
    if element == "H":
        return 0.115 if partner_element == "N" else 0.125
    table = {("C", 4): 0.190, ("C", 3): 0.1875, ("N", 3): 0.1706, ("N", 4): 0.1625, ("O", 1): 0.148}
    if (element, n_bonds) not in table:
        raise NotImplementedError(f"GBSA-OBC radius of {element} with {n_bonds} bonds is not in the verified table")
    return table[(element, n_bonds)]


def _res(types: str, charges: Sequence[float], names: str, bonds: str):
    nm = names.split()
    return dict(names=nm, types=dict(zip(nm, types.split())), charges=dict(zip(nm, charges)),
                bonds=[tuple(b.split("-")) for b in bonds.split()])


_BB = "N-H N-CA CA-HA CA-C C-O CA-CB "
_ASN_SIDE = "CB-HB2 CB-HB3 CB-CG CG-OD1 CG-ND2 ND2-HD21 ND2-HD22"
_GLN_SIDE = "CB-HB2 CB-HB3 CB-CG CG-HG2 CG-HG3 CG-CD CD-OE1 CD-NE2 NE2-HE21 NE2-HE22"
# ff94 residue libraries (atom types, charges); "N..." / "C..." are the charged-terminus variants
RESIDUES = {
    "ACE": _res("HC CT HC HC C O", [0.1123, -0.3662, 0.1123, 0.1123, 0.5972, -0.5679],
                "HH31 CH3 HH32 HH33 C O", "CH3-HH31 CH3-HH32 CH3-HH33 CH3-C C-O"),
    "ALA": _res("N H CT H1 CT HC HC HC C O",
                [-0.4157, 0.2719, 0.0337, 0.0823, -0.1825, 0.0603, 0.0603, 0.0603, 0.5973, -0.5679],
                "N H CA HA CB HB1 HB2 HB3 C O", _BB + "CB-HB1 CB-HB2 CB-HB3"),
    "NME": _res("N H CT H1 H1 H1", [-0.4157, 0.2719, -0.1490, 0.0976, 0.0976, 0.0976],
                "N H CH3 HH31 HH32 HH33", "N-H N-CH3 CH3-HH31 CH3-HH32 CH3-HH33"),
    "ASN": _res("N H CT H1 CT HC HC C O N H H C O",
                [-0.4157, 0.2719, 0.0143, 0.1048, -0.2041, 0.0797, 0.0797, 0.7130, -0.5931, -0.9191, 0.4196, 0.4196,
                 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 CG OD1 ND2 HD21 HD22 C O", _BB + _ASN_SIDE),
    "GLN": _res("N H CT H1 CT HC HC CT HC HC C O N H H C O",
                [-0.4157, 0.2719, -0.0031, 0.0850, -0.0036, 0.0171, 0.0171, -0.0645, 0.0352, 0.0352, 0.6951, -0.6086,
                 -0.9407, 0.4251, 0.4251, 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 CG HG2 HG3 CD OE1 NE2 HE21 HE22 C O", _BB + _GLN_SIDE),
    "NASN": _res("N3 H H H CT HP CT HC HC C O N H H C O",
                 [0.1801, 0.1921, 0.1921, 0.1921, 0.0368, 0.1231, -0.0283, 0.0515, 0.0515, 0.5833, -0.5744, -0.8634,
                  0.4097, 0.4097, 0.6163, -0.5722],
                 "N H H2 H3 CA HA CB HB2 HB3 CG OD1 ND2 HD21 HD22 C O", _BB + "N-H2 N-H3 " + _ASN_SIDE),
    "CGLN": _res("N H CT H1 CT HC HC CT HC HC C O N H H C O2 O2",
                 [-0.3821, 0.2681, -0.2248, 0.1232, -0.0664, 0.0452, 0.0452, -0.0210, 0.0203, 0.0203, 0.7093, -0.6098,
                  -0.9574, 0.4304, 0.4304, 0.7775, -0.8042, -0.8042],
                 "N H CA HA CB HB2 HB3 CG HG2 HG3 CD OE1 NE2 HE21 HE22 C O OXT", _BB + "C-OXT " + _GLN_SIDE),
}

AD_ATOM_NAMES = "HH31 CH3 HH32 HH33 C O N H CA HA CB HB1 HB2 HB3 C O N H CH3 HH31 HH32 HH33".split()
AD_RESIDUES = ["ACE"] * 6 + ["ALA"] * 10 + ["NME"] * 6


def _neighbours(n: int, bonds: Sequence[Tuple[int, int]]) -> List[List[int]]:
    nb: List[List[int]] = [[] for _ in range(n)]
    for i, j in bonds:
        nb[i].append(j)
        nb[j].append(i)
    return nb


def _improper(centre: int, nb: List[int], ty: List[str], el: List[str]):
    """OpenMM's placement of the AMBER wildcard impropers (X-X-C-O, X-O2-C-O2, X-X-N-H) for an sp2 centre with three
    neighbours: (a1, a2, centre, a4) and k in kcal/mol, or None.  a4 is the atom the pattern names last; the other two
    go carbon first, else heavier element first, same element by index (openmm/app/forcefield.py `_matchImproper`,
    confirmed per class against the known-answer forces)."""
    t = ty[centre]
    if t == "C":
        o2 = [a for a in nb if ty[a] == "O2"]
        if len(o2) == 2:
            other = [a for a in nb if ty[a] != "O2"][0]
            return (other, min(o2), centre, max(o2)), 10.5
        last = [a for a in nb if ty[a] == "O"]
        k = 10.5
    elif t == "N":
        last = [a for a in nb if el[a] == "H"]
        # 1.1 for the backbone pattern (C, CT, H); 1.0 for the generic X-X-N-H (amide NH2)
        k = 1.1 if sorted(ty[a] for a in nb) == ["C", "CT", "H"] else 1.0
    else:
        return None
    if not last:
        return None
    a4 = max(last)
    a1, a2 = [a for a in nb if a != a4]
    if el[a1] == el[a2]:
        if a1 > a2:
            a1, a2 = a2, a1
    elif el[a1] != "C" and (el[a2] == "C" or ELEMENT_MASSES[el[a1]] < ELEMENT_MASSES[el[a2]]):
        a1, a2 = a2, a1
    return (a1, a2, centre, a4), k


def amber99sbildn_obc_tables(atom_names: Sequence[str], residue_names: Sequence[str],
                             residue_ids: Sequence[int], family: str = "amber99") -> ForceFieldTables:
    """Tables of `ForceField("amber99sbildn.xml", "amber99_obc.xml").createSystem(topology, CutoffNonPeriodic, 2 nm,
    constraints=None)` (simulation/md.py:150-173) for a single chain made of the residues in `RESIDUES`, atoms in any
    order.  A first residue carrying H2/H3 selects the NH3+ variant, a last residue carrying OXT the COO- variant.
    `family="amber14"`: the amber14-all + implicit/obc1 preset instead (amber14_obc1_tables below: parity UNPINNED)."""
    fam = _FAMILIES[family]
    RESIDUES, _TORSION_SPECIFIC, _TORSION_GENERIC = fam["residues"], fam["torsion_specific"], fam["torsion_generic"]
    parent = fam["parent_type"]  # ff14SB's renamed carbon types keep their parm99 parents' bond / angle / LJ numbers
    n = len(atom_names)
    rids = list(dict.fromkeys(residue_ids))
    index = {(r, a): i for i, (a, r) in enumerate(zip(atom_names, residue_ids))}
    ty, q, el, local = [""] * n, [0.0] * n, [""] * n, [""] * n
    bonds: List[Tuple[int, int]] = []
    prev_c = None
    for pos, rid in enumerate(rids):
        members = [i for i in range(n) if residue_ids[i] == rid]
        res = residue_names[members[0]]
        have = {atom_names[i] for i in members}
        key = res
        if pos == 0 and "H2" in have:
            key = "N" + res
        if pos == len(rids) - 1 and "OXT" in have:
            key = "C" + res
        if key not in RESIDUES:
            raise NotImplementedError(f"no {fam['label']} template for residue {key!r} (have {sorted(RESIDUES)})")
        tpl = RESIDUES[key]
        if have != set(tpl["names"]):
            raise ValueError(f"residue {key} {rid}: atoms {sorted(have)} do not match the template {sorted(tpl['names'])}")
        for i in members:
            nm = atom_names[i]
            ty[i], q[i], local[i] = tpl["types"][nm], tpl["charges"][nm], res
            el[i] = nm[0]
        for a, b in tpl["bonds"]:
            bonds.append((index[(rid, a)], index[(rid, b)]))
        if prev_c is not None:
            bonds.append((prev_c, index[(rid, "N")]))
        prev_c = index.get((rid, "C"))
    nb = _neighbours(n, bonds)
    bond_par = []
    for i, j in bonds:
        k, r0 = _BOND[tuple(sorted((parent(ty[i]), parent(ty[j]))))]
        bond_par.append((r0 * 0.1, 2.0 * k * KCAL * 100.0))
    angle_idx, angle_par = [], []
    for j in range(n):
        for i, k in combinations(sorted(nb[j]), 2):
            a, c = sorted((parent(ty[i]), parent(ty[k])))
            kk, t0 = _ANGLE[(a, parent(ty[j]), c)]
            angle_idx.append((i, j, k))
            angle_par.append((math.radians(t0), 2.0 * kk * KCAL))
    torsion_idx, torsion_par, pairs14 = [], [], set()
    for b, c in bonds:
        for a in nb[b]:
            if a == c:
                continue
            for d in nb[c]:
                if d == b or d == a:
                    continue
                pairs14.add((min(a, d), max(a, d)))
                terms = None
                if fam["asn_fitted"] and local[b] == "ASN" and residue_ids[a] == residue_ids[b] == residue_ids[c] == residue_ids[d]:
                    nm4 = (atom_names[a], atom_names[b], atom_names[c], atom_names[d])
                    terms = _ASN_FITTED_TORSIONS.get(nm4) or _ASN_FITTED_TORSIONS.get(nm4[::-1])
                if terms is None:
                    key = (ty[a], ty[b], ty[c], ty[d])
                    terms = _TORSION_SPECIFIC.get(key) or _TORSION_SPECIFIC.get(key[::-1])
                if terms is None:
                    terms = _TORSION_GENERIC[tuple(sorted((ty[b], ty[c])))]
                for kk, phase, per in terms:
                    torsion_idx.append((a, b, c, d))
                    torsion_par.append((float(per), math.radians(phase), kk * KCAL))
    for c in range(n):
        if len(nb[c]) == 3:
            imp = _improper(c, nb[c], [parent(t) for t in ty], el)
            if imp is not None:
                torsion_idx.append(imp[0])
                torsion_par.append((2.0, math.pi, imp[1] * KCAL))
    sigma = [_LJ[parent(t)][0] * 2.0 / 2.0 ** (1.0 / 6.0) * 0.1 for t in ty]
    eps = [_LJ[parent(t)][1] * KCAL for t in ty]
    atom_par = []
    for i in range(n):
        rad = fam["gb_radius"](el[i], len(nb[i]), el[nb[i][0]])
        atom_par.append((q[i], sigma[i], eps[i], rad, _GB_SCALE[el[i]]))
    # exceptions: 1-2 and 1-3 fully excluded, 1-4 scaled (Coulomb 1/1.2, LJ 1/2)
    excl = set()
    for i, j in bonds:
        excl.add((min(i, j), max(i, j)))
    for j in range(n):
        for i, k in combinations(sorted(nb[j]), 2):
            excl.add((i, k))
    exc_idx, exc_par = [], []
    for i, j in sorted(excl):
        exc_idx.append((i, j))
        exc_par.append((0.0, 1.0, 0.0))
    for i, j in sorted(pairs14 - excl):
        exc_idx.append((i, j))
        exc_par.append((q[i] * q[j] / 1.2, 0.5 * (sigma[i] + sigma[j]), math.sqrt(eps[i] * eps[j]) / 2.0))
    f = lambda a, w: np.asarray(a, dtype=np.float64).reshape(-1, w)
    g = lambda a, w: np.asarray(a, dtype=np.int32).reshape(-1, w)
    # GBSAOBCForce's own default solvent dielectric (78.3; createSystem does not override it), surface term 2.25936
    return ForceFieldTables(g(bonds, 2), f(bond_par, 2), g(angle_idx, 3), f(angle_par, 2), g(torsion_idx, 4),
                            f(torsion_par, 3), g(exc_idx, 2), f(exc_par, 3), f(atom_par, 5), has_gbsa=fam["has_gbsa"],
                            solvent_dielectric=fam["solvent_dielectric"])


# ---------------------------------------------------------------------------------------------------
# amber14-all (ff14SB) + implicit/obc1 (GBSA-OBC I): the `T1B-peptides` / 4AA / 2AA preset, simulation/md.py:31-32, 153-159
#
# PARITY UNPINNED.  The reference holds no known-answer data for this preset and OpenMM's XML files are not available
# offline, so nothing below could be checked against OpenMM.  What is here is ff14SB as published, for the residues whose
# parameters could be written down with confidence, and nothing else:
#   * residues ACE, NME, ALA, GLY and the charged-terminus forms NALA / CALA / NGLY / CGLY.  ff14SB keeps the ff94 charges
#     and the parm99 bond / angle / van der Waals numbers; it renames the alpha carbon CX (same bond, angle, LJ parameters
#     as CT) and refits side-chain torsions PER RESIDUE - which is why no residue with a rotatable side chain is offered:
#     ASN / GLN (the 99SB-ILDN entries above) and every other residue raise NotImplementedError under this family.
#   * backbone torsions of frcmod.ff14SB: phi and psi as in ff99SB; phi' (C-N-CX-CT) 0.8 / 1.8 / 2.0 kcal/mol for n = 3 / 2 / 1
#     and psi' (CT-CX-C-N) 0.4 / 0.2 / 0.2, all with phase 0 - RECALLED values, the least certain numbers in this file.
#   * GBSA-OBC I: alpha = 0.8, beta = 0, gamma = 2.909125 in the kernel and the C oracle (has_gbsa = 2, equal at 1e-10); the
#     mbondi2 radii AMBER prescribes for igb = 2 (H 0.12 nm, 0.13 on nitrogen; C 0.17; N 0.155; O 0.15), the OBC scale
#     factors by element (the ones the known-answer file pins for OBC II), solvent dielectric 78.5.  Which radius set
#     OpenMM's implicit/obc1.xml actually assigns is NOT known here.
# What can be checked is checked in tests/test_host_logic.py: every template is neutral / +-1, every type resolves, the
# amber14 tables of alanine dipeptide differ from the pinned amber99 ones ONLY in the phi' / psi' series, the GB mode and the
# GB radii.  With OpenMM present, `tables_from_openmm_system` is the route that needs none of this.
# ---------------------------------------------------------------------------------------------------
def _gb_radius_mbondi2(element: str, n_bonds: int, partner_element: str) -> float:
    if element == "H":
        return 0.13 if partner_element == "N" else 0.12
    return {"C": 0.17, "N": 0.155, "O": 0.15}[element]


_GLY_BONDS = "N-H N-CA CA-HA2 CA-HA3 CA-C C-O"
_RESIDUES_FF14SB = {
    "ACE": RESIDUES["ACE"],
    "NME": RESIDUES["NME"],
    "ALA": _res("N H CX H1 CT HC HC HC C O",
                [-0.4157, 0.2719, 0.0337, 0.0823, -0.1825, 0.0603, 0.0603, 0.0603, 0.5973, -0.5679],
                "N H CA HA CB HB1 HB2 HB3 C O", _BB + "CB-HB1 CB-HB2 CB-HB3"),
    "GLY": _res("N H CX H1 H1 C O", [-0.4157, 0.2719, -0.0252, 0.0698, 0.0698, 0.5973, -0.5679],
                "N H CA HA2 HA3 C O", _GLY_BONDS),
    "NALA": _res("N3 H H H CX HP CT HC HC HC C O",
                 [0.1414, 0.1997, 0.1997, 0.1997, 0.0962, 0.0889, -0.0597, 0.0300, 0.0300, 0.0300, 0.6163, -0.5722],
                 "N H H2 H3 CA HA CB HB1 HB2 HB3 C O", _BB + "N-H2 N-H3 CB-HB1 CB-HB2 CB-HB3"),
    "CALA": _res("N H CX H1 CT HC HC HC C O2 O2",
                 [-0.3821, 0.2681, -0.1747, 0.1067, -0.2093, 0.0764, 0.0764, 0.0764, 0.7731, -0.8055, -0.8055],
                 "N H CA HA CB HB1 HB2 HB3 C O OXT", _BB + "C-OXT CB-HB1 CB-HB2 CB-HB3"),
    "NGLY": _res("N3 H H H CX HP HP C O", [0.2943, 0.1642, 0.1642, 0.1642, -0.0100, 0.0895, 0.0895, 0.6163, -0.5722],
                 "N H H2 H3 CA HA2 HA3 C O", _GLY_BONDS + " N-H2 N-H3"),
    "CGLY": _res("N H CX H1 H1 C O2 O2", [-0.3821, 0.2681, -0.2493, 0.1056, 0.1056, 0.7231, -0.7855, -0.7855],
                 "N H CA HA2 HA3 C O OXT", _GLY_BONDS + " C-OXT"),
}
_TORSION_SPECIFIC_FF14SB = {
    ("C", "N", "CX", "C"): [(0.42, 0.0, 3), (0.27, 0.0, 2)],                        # phi  (as ff99SB)
    ("N", "CX", "C", "N"): [(0.55, 180.0, 3), (1.58, 180.0, 2), (0.45, 180.0, 1)],  # psi  (as ff99SB)
    ("C", "N", "CX", "CT"): [(0.80, 0.0, 3), (1.80, 0.0, 2), (2.00, 0.0, 1)],       # phi' (ff14SB; recalled)
    ("CT", "CX", "C", "N"): [(0.40, 0.0, 3), (0.20, 0.0, 2), (0.20, 0.0, 1)],       # psi' (ff14SB; recalled)
    ("H", "N", "C", "O"): [(2.50, 180.0, 2), (2.00, 0.0, 1)],
    ("HC", "CT", "C", "O"): [(0.80, 0.0, 1), (0.08, 180.0, 3)],
    ("H1", "CX", "C", "O"): [(0.80, 0.0, 1), (0.08, 180.0, 3)],
    ("HC", "CT", "CX", "H1"): [(1.40 / 9.0, 0.0, 3)],
}
_TORSION_GENERIC_FF14SB = {
    ("C", "N"): [(2.50, 180.0, 2)],
    ("CT", "CX"): [(1.40 / 9.0, 0.0, 3)],      # X-CT-CX-X as X-CT-CT-X
    ("C", "CT"): [], ("C", "CX"): [],
    ("CT", "N"): [], ("CX", "N"): [],
    ("CX", "N3"): [(1.40 / 9.0, 0.0, 3)],
}
_FAMILIES = {
    "amber99": dict(label="amber99sb-ildn", residues=RESIDUES, torsion_specific=_TORSION_SPECIFIC,
                    torsion_generic=_TORSION_GENERIC, parent_type=lambda t: t, asn_fitted=True, gb_radius=_gb_radius,
                    has_gbsa=1, solvent_dielectric=78.3),  # GBSAOBCForce's own default (createSystem does not override it)
    "amber14": dict(label="amber14 (ff14SB, parity unpinned)", residues=_RESIDUES_FF14SB,
                    torsion_specific=_TORSION_SPECIFIC_FF14SB, torsion_generic=_TORSION_GENERIC_FF14SB,
                    parent_type=lambda t: "CT" if t == "CX" else t, asn_fitted=False, gb_radius=_gb_radius_mbondi2,
                    has_gbsa=2, solvent_dielectric=78.5),
}
PRESET_FAMILY = {  # simulation/md.py:31-37 (dataset -> preset) and :153-159 (preset -> force-field files)
    "T1B-peptides": "amber14", "amber14-implicit": "amber14",
    "T1-peptides": "amber99", "HP-1400": "amber99", "HP-4000": "amber99", "alanine-dipeptide": "amber99",
    "amber99-implicit-old": "amber99", "amber99-implicit": "amber99",
}


def amber14_obc1_tables(atom_names: Sequence[str], residue_names: Sequence[str], residue_ids: Sequence[int]) -> ForceFieldTables:
    """amber14-all + implicit/obc1 tables (the T1B-peptides / 4AA preset) for chains of ACE / NME / ALA / GLY (+ charged
    termini).  PARITY UNPINNED - see the section comment; other residues raise NotImplementedError."""
    return amber99sbildn_obc_tables(atom_names, residue_names, residue_ids, family="amber14")


def tables_for_preset(preset_or_dataset: str, atom_names: Sequence[str], residue_names: Sequence[str],
                      residue_ids: Sequence[int]) -> ForceFieldTables:
    """The reference's `get_system(model, preset)` (simulation/md.py:128-187) without OpenMM: tables for a dataset or
    preset name and a topology given as per-atom (name, residue name, residue id)."""
    if preset_or_dataset not in PRESET_FAMILY:
        raise ValueError(f"unknown dataset / preset {preset_or_dataset!r} (known: {sorted(PRESET_FAMILY)}); explicit-solvent "
                         "presets (amber14-explicit) are periodic and not supported")
    return amber99sbildn_obc_tables(atom_names, residue_names, residue_ids, family=PRESET_FAMILY[preset_or_dataset])


def alanine_dipeptide_amber99sb() -> ForceFieldTables:
    """The 22-atom ACE-ALA-NME topology of `simulation/testdata/alanine-dipeptide.pdb` (atom order of that file)."""
    rid = {"ACE": 1, "ALA": 2, "NME": 3}
    return amber99sbildn_obc_tables(AD_ATOM_NAMES, AD_RESIDUES, [rid[r] for r in AD_RESIDUES])


def alanine_dipeptide_masses() -> np.ndarray:
    return np.asarray([ELEMENT_MASSES[nm[0]] for nm in AD_ATOM_NAMES], dtype=np.float32)


def tables_from_pdb(path: str) -> ForceFieldTables:
    """amber99sb-ildn + OBC tables for the ATOM records of a PDB file (residues limited to `RESIDUES`)."""
    names, res, rid = [], [], []
    for line in open(path):
        if line.startswith(("ATOM", "HETATM")):
            names.append(line[12:16].strip())
            res.append(line[17:20].strip())
            rid.append(int(line[22:26]))
    return amber99sbildn_obc_tables(names, res, rid)


def _md(x) -> float:
    """Value of an OpenMM `Quantity` in OpenMM's MD unit system (nm, kJ/mol, elementary charge, radian, kelvin) - the
    units of `tw_forcefield` - or the number itself (getters of unitless values and of CustomGBForce return floats)."""
    if hasattr(x, "value_in_unit_system"):
        import openmm.unit as u

        return float(x.value_in_unit_system(u.md_unit_system))
    return float(x)


# Forces that contribute no potential energy (simulation/md.py:160-187 adds none of these itself; OpenMM's
# createSystem adds CMMotionRemover); anything else that is not handled below is an error, not a silent skip.
_ENERGY_FREE_FORCES = ("CMMotionRemover", "AndersenThermostat", "MonteCarloBarostat", "MonteCarloAnisotropicBarostat",
                       "MonteCarloMembraneBarostat", "MonteCarloFlexibleBarostat")
_NO_CUTOFF, _CUTOFF_NON_PERIODIC = 0, 1  # openmm.NonbondedForce.NoCutoff / CutoffNonPeriodic


def tables_from_openmm_system(system, allow_custom_gb_obc1: bool = True) -> ForceFieldTables:
    """Extract the tables from an `openmm.System` (what evaluate.py:290-301 / sample_trajectory.py:190-202 build with
    simulation/md.py:128-187).  Mirrors the Force getters of OpenMM 7.7; forces are recognised by class name, so any
    object with those getters works (tests/test_host_logic.py drives it with a stand-in System).  A force that
    carries energy and is not one of HarmonicBond / HarmonicAngle / PeriodicTorsion / Nonbonded (NoCutoff or
    CutoffNonPeriodic) / GBSAOBC / CustomGB-as-OBC-I raises NotImplementedError: dropping it would bias the MH
    acceptance silently."""
    out: Dict[str, list] = {k: [] for k in ("bi", "bp", "ai", "ap", "ti", "tp", "ei", "ep")}
    atom_par = np.zeros((system.getNumParticles(), 5))
    kw = dict(has_gbsa=0, cutoff=0.0, rf_dielectric=78.3)
    seen = set()
    for force in system.getForces():
        kind = type(force).__name__
        if kind in _ENERGY_FREE_FORCES:
            continue
        if kind in seen and kind in ("NonbondedForce", "GBSAOBCForce", "CustomGBForce"):
            raise NotImplementedError(f"two {kind} objects in one System")
        seen.add(kind)
        if kind == "HarmonicBondForce":
            for b in range(force.getNumBonds()):
                i, j, r0, k = force.getBondParameters(b)
                out["bi"].append((i, j)); out["bp"].append((_md(r0), _md(k)))
        elif kind == "HarmonicAngleForce":
            for a in range(force.getNumAngles()):
                i, j, k_, t0, k = force.getAngleParameters(a)
                out["ai"].append((i, j, k_)); out["ap"].append((_md(t0), _md(k)))
        elif kind == "PeriodicTorsionForce":
            for t in range(force.getNumTorsions()):
                a, b, c, d, per, phase, k = force.getTorsionParameters(t)
                out["ti"].append((a, b, c, d)); out["tp"].append((float(per), _md(phase), _md(k)))
        elif kind == "NonbondedForce":
            method = int(force.getNonbondedMethod())
            if method not in (_NO_CUTOFF, _CUTOFF_NON_PERIODIC):
                raise NotImplementedError("periodic nonbonded methods (CutoffPeriodic / Ewald / PME) are not supported")
            for i in range(force.getNumParticles()):
                q, sig, eps = force.getParticleParameters(i)
                atom_par[i, 0:3] = (_md(q), _md(sig), _md(eps))
            for e in range(force.getNumExceptions()):
                i, j, qq, sig, eps = force.getExceptionParameters(e)
                out["ei"].append((i, j))
                out["ep"].append((_md(qq), _md(sig), _md(eps)))
            if method == _CUTOFF_NON_PERIODIC:
                kw["cutoff"] = _md(force.getCutoffDistance())
            kw["rf_dielectric"] = float(force.getReactionFieldDielectric())
        elif kind == "GBSAOBCForce":
            kw["has_gbsa"] = 1
            for i in range(force.getNumParticles()):
                _, radius, scale = force.getParticleParameters(i)
                atom_par[i, 3:5] = (_md(radius), _md(scale))
            kw["solute_dielectric"] = float(force.getSoluteDielectric())
            kw["solvent_dielectric"] = float(force.getSolventDielectric())
            kw["surface_area_energy"] = _md(force.getSurfaceAreaEnergy())
        elif kind in ("CustomGBForce", "GBSAOBC1Force") and allow_custom_gb_obc1:
            # amber14's implicit/obc1.xml (T1B-peptides preset, simulation/md.py:153-159) builds GBSA-OBC I as a
            # CustomGBForce (openmm.app.internal.customgbforces.GBSAOBC1Force): per-particle parameters
            # (charge, or, sr) with or = radius - 0.009 nm and sr = scale * or; recognised by its tanh coefficients.
            # No known-answer data exists for this mode (parity unpinned).
            exprs = " ".join(force.getComputedValueParameters(i)[1] for i in range(force.getNumComputedValues()))
            if "2.909125" not in exprs:
                raise NotImplementedError("CustomGBForce other than GBSA-OBC I (implicit/obc1.xml) is not supported")
            kw["has_gbsa"] = 2
            names = [force.getPerParticleParameterName(i) for i in range(force.getNumPerParticleParameters())]
            for i in range(force.getNumParticles()):
                par = dict(zip(names, (_md(v) for v in force.getParticleParameters(i))))
                o_r = par.get("or", par.get("radius"))
                s_r = par.get("sr", par.get("scale"))
                radius = o_r + 0.009 if "or" in par else o_r
                scale = s_r / o_r if "sr" in par else s_r
                atom_par[i, 3:5] = (radius, scale)
            gp = {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i) for i in range(force.getNumGlobalParameters())}
            kw["solute_dielectric"] = float(gp.get("soluteDielectric", 1.0))
            kw["solvent_dielectric"] = float(gp.get("solventDielectric", 78.5))
            # the ACE term's coefficient is written into the energy expression, 28.3919551 = 4 pi * 2.25936 kJ/mol/nm^2
            energy_terms = " ".join(force.getEnergyTermParameters(i)[0] for i in range(force.getNumEnergyTerms()))
            if "28.3919551" not in energy_terms:
                raise NotImplementedError("GBSA-OBC I CustomGBForce without the standard ACE surface term (28.3919551)")
            kw["surface_area_energy"] = 2.25936
        else:
            raise NotImplementedError(
                f"openmm force {kind!r} is not evaluated by the HIP energy kernel (supported: HarmonicBondForce, "
                "HarmonicAngleForce, PeriodicTorsionForce, NonbondedForce without periodic boundary, GBSAOBCForce, "
                "GBSA-OBC I as CustomGBForce); its energy would be missing from the MH acceptance")
    f = lambda a, w: np.asarray(a, dtype=np.float64).reshape(-1, w)
    g = lambda a, w: np.asarray(a, dtype=np.int32).reshape(-1, w)
    return ForceFieldTables(g(out["bi"], 2), f(out["bp"], 2), g(out["ai"], 3), f(out["ap"], 2), g(out["ti"], 4),
                            f(out["tp"], 3), g(out["ei"], 2), f(out["ep"], 3), atom_par, **kw)
