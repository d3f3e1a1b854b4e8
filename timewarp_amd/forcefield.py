"""Force-field parameter tables for the HIP energy kernel (`tw_forcefield`, include/timewarp_hip.h).

The reference never holds these numbers itself: `simulation/md.py:150-173` asks OpenMM's
`ForceField("amber99sbildn.xml", "amber99_obc.xml").createSystem(...)` for them.  Two sources here:

* `tables_from_openmm_system(system)` -- reads the tables out of an `openmm.System` exactly as the
  scripts build it (the drop-in route; needs OpenMM importable, which it is not in this image);
* `alanine_dipeptide_amber99sb()` -- a hand-authored table for the 22-atom ACE-ALA-NME topology of
  `simulation/testdata/alanine-dipeptide.pdb` so the whole MH path can run without OpenMM.
  PARITY UNPINNED: the numbers are the published parm99 / ff99SB / OBC values written from general
  knowledge; they have not been checked against OpenMM output (none is available offline).

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
# hand-authored alanine dipeptide (ACE-ALA-NME) table
# ---------------------------------------------------------------------------------------------------
AD_ATOM_NAMES = "HH31 CH3 HH32 HH33 C O N H CA HA CB HB1 HB2 HB3 C O N H CH3 HH31 HH32 HH33".split()
AD_TYPES = "HC CT HC HC C O N H CT H1 CT HC HC HC C O N H CT H1 H1 H1".split()
AD_CHARGES = [0.1123, -0.3662, 0.1123, 0.1123, 0.5972, -0.5679, -0.4157, 0.2719, 0.0337, 0.0823, -0.1825,
              0.0603, 0.0603, 0.0603, 0.5973, -0.5679, -0.4157, 0.2719, -0.1490, 0.0976, 0.0976, 0.0976]
AD_BONDS = [(0, 1), (1, 2), (1, 3), (1, 4), (4, 5), (4, 6), (6, 7), (6, 8), (8, 9), (8, 10), (10, 11), (10, 12),
            (10, 13), (8, 14), (14, 15), (14, 16), (16, 17), (16, 18), (18, 19), (18, 20), (18, 21)]
AD_MASSES = {"C": 12.01, "H": 1.008, "N": 14.01, "O": 16.0}

# parm99: Rmin/2 (Angstrom), eps (kcal/mol)
_LJ = {"H": (0.6000, 0.0157), "HC": (1.4870, 0.0157), "H1": (1.3870, 0.0157), "CT": (1.9080, 0.1094),
       "C": (1.9080, 0.0860), "N": (1.8240, 0.1700), "O": (1.6612, 0.2100)}
# parm99 bonds: k (kcal/mol/A^2, E = k (r-r0)^2), r0 (A)
_BOND = {("CT", "HC"): (340.0, 1.090), ("CT", "H1"): (340.0, 1.090), ("C", "CT"): (317.0, 1.522),
         ("C", "O"): (570.0, 1.229), ("C", "N"): (490.0, 1.335), ("H", "N"): (434.0, 1.010),
         ("CT", "N"): (337.0, 1.449), ("CT", "CT"): (310.0, 1.526)}
# parm99 angles: k (kcal/mol/rad^2, E = k (t-t0)^2), theta0 (deg); keyed (a, centre, c) with a <= c
_ANGLE = {("HC", "CT", "HC"): (35.0, 109.50), ("C", "CT", "HC"): (50.0, 109.50), ("CT", "C", "O"): (80.0, 120.40),
          ("CT", "C", "N"): (70.0, 116.60), ("N", "C", "O"): (80.0, 122.90), ("C", "N", "H"): (50.0, 120.00),
          ("C", "N", "CT"): (50.0, 121.90), ("CT", "N", "H"): (50.0, 118.04), ("H1", "CT", "N"): (50.0, 109.50),
          ("CT", "CT", "N"): (80.0, 109.70), ("C", "CT", "N"): (63.0, 110.10), ("CT", "CT", "H1"): (50.0, 109.50),
          ("C", "CT", "H1"): (50.0, 109.50), ("C", "CT", "CT"): (63.0, 111.10), ("CT", "CT", "HC"): (50.0, 109.50),
          ("H1", "CT", "H1"): (35.0, 109.50)}
# ff99SB propers: list of (k kcal/mol, phase deg, n); generic entries use "X"
_TORSION_SPECIFIC = {
    ("C", "N", "CT", "C"): [(0.42, 0.0, 3), (0.27, 0.0, 2)],                        # phi
    ("N", "CT", "C", "N"): [(0.55, 180.0, 3), (1.58, 180.0, 2), (0.45, 180.0, 1)],  # psi
    ("C", "N", "CT", "CT"): [(0.40, 0.0, 3), (2.00, 0.0, 2), (2.00, 0.0, 1)],       # phi'
    ("CT", "CT", "C", "N"): [(0.40, 0.0, 3), (0.20, 0.0, 2), (0.20, 0.0, 1)],       # psi'
    ("H", "N", "C", "O"): [(2.50, 180.0, 2), (2.00, 0.0, 1)],
    ("HC", "CT", "C", "O"): [(0.80, 0.0, 1), (0.08, 180.0, 3)],
    ("H1", "CT", "C", "O"): [(0.80, 0.0, 1), (0.08, 180.0, 3)],
}
_TORSION_GENERIC = {  # keyed by the two central types (sorted): per-path k, phase, n
    ("C", "N"): [(2.50, 180.0, 2)],            # X-C-N-X   10.0 / 4 paths
    ("CT", "CT"): [(1.40 / 9.0, 0.0, 3)],      # X-CT-CT-X
    ("C", "CT"): [],                           # X-C-CT-X  0.0
    ("CT", "N"): [],                           # X-CT-N-X  0.0
}
_IMPROPERS = [((1, 6, 4, 5), 10.5), ((4, 8, 6, 7), 1.0), ((8, 16, 14, 15), 10.5), ((14, 18, 16, 17), 1.0)]
# amber99_obc.xml (mbondi2-style): radius (nm), scale
_GB = {"H": (0.12, 0.85), "C": (0.17, 0.72), "N": (0.155, 0.79), "O": (0.15, 0.85)}


def _neighbours(n: int, bonds: Sequence[Tuple[int, int]]) -> List[List[int]]:
    nb: List[List[int]] = [[] for _ in range(n)]
    for i, j in bonds:
        nb[i].append(j)
        nb[j].append(i)
    return nb


def alanine_dipeptide_amber99sb() -> ForceFieldTables:
    n = len(AD_TYPES)
    ty, nb = AD_TYPES, _neighbours(len(AD_TYPES), AD_BONDS)
    bond_par = []
    for i, j in AD_BONDS:
        k, r0 = _BOND[tuple(sorted((ty[i], ty[j])))]
        bond_par.append((r0 * 0.1, 2.0 * k * KCAL * 100.0))
    angle_idx, angle_par = [], []
    for j in range(n):
        for i, k in combinations(sorted(nb[j]), 2):
            a, c = sorted((ty[i], ty[k]))
            kk, t0 = _ANGLE[(a, ty[j], c)]
            angle_idx.append((i, j, k))
            angle_par.append((math.radians(t0), 2.0 * kk * KCAL))
    torsion_idx, torsion_par, pairs14 = [], [], set()
    for b, c in AD_BONDS:
        for a in nb[b]:
            if a == c:
                continue
            for d in nb[c]:
                if d == b or d == a:
                    continue
                pairs14.add((min(a, d), max(a, d)))
                key = (ty[a], ty[b], ty[c], ty[d])
                terms = _TORSION_SPECIFIC.get(key) or _TORSION_SPECIFIC.get(key[::-1])
                if terms is None:
                    terms = _TORSION_GENERIC[tuple(sorted((ty[b], ty[c])))]
                for kk, phase, per in terms:
                    torsion_idx.append((a, b, c, d))
                    torsion_par.append((float(per), math.radians(phase), kk * KCAL))
    for (a, b, c, d), kk in _IMPROPERS:
        torsion_idx.append((a, b, c, d))
        torsion_par.append((2.0, math.pi, kk * KCAL))
    sigma = [_LJ[t][0] * 2.0 / 2.0 ** (1.0 / 6.0) * 0.1 for t in ty]
    eps = [_LJ[t][1] * KCAL for t in ty]
    atom_par = []
    for i in range(n):
        el = AD_ATOM_NAMES[i][0]
        rad, sc = _GB[el]
        if el == "H" and ty[nb[i][0]] == "N":
            rad = 0.13  # hydrogens bound to nitrogen
        atom_par.append((AD_CHARGES[i], sigma[i], eps[i], rad, sc))
    # exceptions: 1-2 and 1-3 fully excluded, 1-4 scaled (Coulomb 1/1.2, LJ 1/2)
    excl = set()
    for i, j in AD_BONDS:
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
        exc_par.append((AD_CHARGES[i] * AD_CHARGES[j] / 1.2, 0.5 * (sigma[i] + sigma[j]), math.sqrt(eps[i] * eps[j]) / 2.0))
    f = lambda a, w: np.asarray(a, dtype=np.float64).reshape(-1, w)
    g = lambda a, w: np.asarray(a, dtype=np.int32).reshape(-1, w)
    return ForceFieldTables(g(AD_BONDS, 2), f(bond_par, 2), g(angle_idx, 3), f(angle_par, 2), g(torsion_idx, 4),
                            f(torsion_par, 3), g(exc_idx, 2), f(exc_par, 3), f(atom_par, 5))


def alanine_dipeptide_masses() -> np.ndarray:
    return np.asarray([AD_MASSES[nm[0]] for nm in AD_ATOM_NAMES], dtype=np.float32)


def tables_from_openmm_system(system) -> ForceFieldTables:  # pragma: no cover - needs OpenMM
    """Extract the tables from an `openmm.System` (what evaluate.py:290-301 /
    sample_trajectory.py:190-202 build).  Mirrors the Force API of OpenMM 7.7."""
    import openmm
    import openmm.unit as u

    nm, kj, rad = u.nanometer, u.kilojoule_per_mole, u.radian
    out: Dict[str, list] = {k: [] for k in ("bi", "bp", "ai", "ap", "ti", "tp", "ei", "ep")}
    atom_par = np.zeros((system.getNumParticles(), 5))
    kw = dict(has_gbsa=0, cutoff=0.0, rf_dielectric=78.3)
    for force in system.getForces():
        if isinstance(force, openmm.HarmonicBondForce):
            for b in range(force.getNumBonds()):
                i, j, r0, k = force.getBondParameters(b)
                out["bi"].append((i, j)); out["bp"].append((r0.value_in_unit(nm), k.value_in_unit(kj / nm**2)))
        elif isinstance(force, openmm.HarmonicAngleForce):
            for a in range(force.getNumAngles()):
                i, j, k_, t0, k = force.getAngleParameters(a)
                out["ai"].append((i, j, k_)); out["ap"].append((t0.value_in_unit(rad), k.value_in_unit(kj / rad**2)))
        elif isinstance(force, openmm.PeriodicTorsionForce):
            for t in range(force.getNumTorsions()):
                a, b, c, d, per, phase, k = force.getTorsionParameters(t)
                out["ti"].append((a, b, c, d)); out["tp"].append((float(per), phase.value_in_unit(rad), k.value_in_unit(kj)))
        elif isinstance(force, openmm.NonbondedForce):
            for i in range(force.getNumParticles()):
                q, sig, eps = force.getParticleParameters(i)
                atom_par[i, 0:3] = (q.value_in_unit(u.elementary_charge), sig.value_in_unit(nm), eps.value_in_unit(kj))
            for e in range(force.getNumExceptions()):
                i, j, qq, sig, eps = force.getExceptionParameters(e)
                out["ei"].append((i, j))
                out["ep"].append((qq.value_in_unit(u.elementary_charge**2), sig.value_in_unit(nm), eps.value_in_unit(kj)))
            if force.getNonbondedMethod() != openmm.NonbondedForce.NoCutoff:
                kw["cutoff"] = force.getCutoffDistance().value_in_unit(nm)
            kw["rf_dielectric"] = force.getReactionFieldDielectric()
        elif isinstance(force, openmm.GBSAOBCForce):
            kw["has_gbsa"] = 1
            for i in range(force.getNumParticles()):
                _, radius, scale = force.getParticleParameters(i)
                atom_par[i, 3:5] = (radius.value_in_unit(nm), scale)
            kw["solute_dielectric"] = force.getSoluteDielectric()
            kw["solvent_dielectric"] = force.getSolventDielectric()
            kw["surface_area_energy"] = force.getSurfaceAreaEnergy().value_in_unit(kj / nm**2)
        elif isinstance(force, openmm.CustomGBForce):
            # amber14's implicit/obc1.xml (T1B-peptides preset, simulation/md.py) builds GBSA-OBC I as a CustomGBForce
            # (openmm.app.internal.customgbforces.GBSAOBC1Force): per-particle parameters (charge, or, sr) with
            # or = radius - 0.009 nm and sr = scale * or; recognised by its tanh coefficients.  Untested here (no OpenMM).
            exprs = " ".join(force.getComputedValueParameters(i)[1] for i in range(force.getNumComputedValues()))
            if "2.909125" not in exprs:
                raise NotImplementedError("CustomGBForce other than GBSA-OBC I (implicit/obc1.xml) is not supported")
            kw["has_gbsa"] = 2
            names = [force.getPerParticleParameterName(i) for i in range(force.getNumPerParticleParameters())]
            for i in range(force.getNumParticles()):
                par = dict(zip(names, force.getParticleParameters(i)))
                o_r = par.get("or", par.get("radius"))
                s_r = par.get("sr", par.get("scale"))
                radius = o_r + 0.009 if "or" in par else o_r
                scale = s_r / o_r if "sr" in par else s_r
                atom_par[i, 3:5] = (radius, scale)
            gp = {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i) for i in range(force.getNumGlobalParameters())}
            kw["solute_dielectric"] = gp.get("soluteDielectric", 1.0)
            kw["solvent_dielectric"] = gp.get("solventDielectric", 78.5)
            kw["surface_area_energy"] = 2.25936
        elif isinstance(force, openmm.CustomGBForce):
            # amber14's implicit/obc1.xml (T1B-peptides preset, simulation/md.py) builds GBSA-OBC I as a CustomGBForce
            # (openmm.app.internal.customgbforces.GBSAOBC1Force): per-particle parameters (charge, or, sr) with
            # or = radius - 0.009 nm and sr = scale * or; recognised by its tanh coefficients.  Untested here (no OpenMM).
            exprs = " ".join(force.getComputedValueParameters(i)[1] for i in range(force.getNumComputedValues()))
            if "2.909125" not in exprs:
                raise NotImplementedError("CustomGBForce other than GBSA-OBC I (implicit/obc1.xml) is not supported")
            kw["has_gbsa"] = 2
            names = [force.getPerParticleParameterName(i) for i in range(force.getNumPerParticleParameters())]
            for i in range(force.getNumParticles()):
                par = dict(zip(names, force.getParticleParameters(i)))
                o_r = par.get("or", par.get("radius"))
                s_r = par.get("sr", par.get("scale"))
                radius = o_r + 0.009 if "or" in par else o_r
                scale = s_r / o_r if "sr" in par else s_r
                atom_par[i, 3:5] = (radius, scale)
            gp = {force.getGlobalParameterName(i): force.getGlobalParameterDefaultValue(i) for i in range(force.getNumGlobalParameters())}
            kw["solute_dielectric"] = gp.get("soluteDielectric", 1.0)
            kw["solvent_dielectric"] = gp.get("solventDielectric", 78.5)
            kw["surface_area_energy"] = 2.25936
    f = lambda a, w: np.asarray(a, dtype=np.float64).reshape(-1, w)
    g = lambda a, w: np.asarray(a, dtype=np.int32).reshape(-1, w)
    return ForceFieldTables(g(out["bi"], 2), f(out["bp"], 2), g(out["ai"], 3), f(out["ap"], 2), g(out["ti"], 4),
                            f(out["tp"], 3), g(out["ei"], 2), f(out["ep"], 3), atom_par, **kw)
