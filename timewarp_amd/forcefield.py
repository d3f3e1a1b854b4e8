"""Force-field parameter tables for the HIP energy kernel (`tw_forcefield`, include/timewarp_hip.h).

The reference never holds these numbers itself: `simulation/md.py:150-173` asks OpenMM's
`ForceField("amber99sbildn.xml", "amber99_obc.xml").createSystem(...)` for them.  Two sources here:

* `tables_from_openmm_system(system)` -- reads the tables out of an `openmm.System` exactly as the
  scripts build it (the drop-in route; needs OpenMM importable, which it is not in this image);
* `amber99sbildn_obc_tables(...)` / `tables_from_pdb(path)` / `alanine_dipeptide_amber99sb()` -- the published
  parm99 / ff99SB / ff94-charge / OBC numbers for ACE, NME and 18 of the 20 amino acids (not HIS, CYS), written out here so
  the whole MH path runs without OpenMM.  Pinned against the reference's own OpenMM known-answer files: 40 (+ 342) frames
  of the peptide NNQQ (simulation/tests/test_md.py:35-83) and 140 frames of a 691-atom protein, energies and forces; see
  the section comments below for what they cover and for the side-chain torsion series of ASN / ILE / LEU / ASP
  (ff99SB-ILDN's) that had to be fitted.

Units: nm, kJ/mol, elementary charge, radians.
"""
from __future__ import annotations

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
       "O": (1.6612, 0.2100), "O2": (1.6612, 0.2100),
       # r04: the types of the other residues (pinned by the 691-atom protein file, see RESIDUES below)
       "HO": (0.0000, 0.0000), "HA": (1.4590, 0.0150), "H4": (1.4090, 0.0150), "CA": (1.9080, 0.0860), "C*": (1.9080, 0.0860),
       "CW": (1.9080, 0.0860), "CB": (1.9080, 0.0860), "CN": (1.9080, 0.0860), "N2": (1.8240, 0.1700), "NA": (1.8240, 0.1700),
       "OH": (1.7210, 0.2104), "S": (2.0000, 0.2500)}
# parm99 bonds: k (kcal/mol/A^2, E = k (r-r0)^2), r0 (A)
_BOND = {("CT", "HC"): (340.0, 1.090), ("CT", "H1"): (340.0, 1.090), ("CT", "HP"): (340.0, 1.090),
         ("C", "CT"): (317.0, 1.522), ("C", "O"): (570.0, 1.229), ("C", "N"): (490.0, 1.335),
         ("H", "N"): (434.0, 1.010), ("CT", "N"): (337.0, 1.449), ("CT", "CT"): (310.0, 1.526),
         ("H", "N3"): (434.0, 1.010), ("CT", "N3"): (367.0, 1.471), ("C", "O2"): (656.0, 1.250),
         ("C", "CA"): (469.0, 1.409), ("C", "OH"): (450.0, 1.364), ("CA", "CA"): (469.0, 1.400), ("CA", "CB"): (469.0, 1.404),
         ("CA", "CN"): (469.0, 1.400), ("CA", "CT"): (317.0, 1.510), ("CA", "HA"): (367.0, 1.080), ("CA", "N2"): (481.0, 1.340),
         ("C*", "CB"): (388.0, 1.459), ("CB", "CN"): (447.0, 1.419), ("C*", "CW"): (546.0, 1.352), ("C*", "CT"): (317.0, 1.495),
         ("CN", "NA"): (428.0, 1.380), ("CW", "H4"): (367.0, 1.080), ("CW", "NA"): (427.0, 1.381), ("CT", "N2"): (337.0, 1.463),
         ("CT", "OH"): (320.0, 1.410), ("CT", "S"): (227.0, 1.810), ("H", "N2"): (434.0, 1.010), ("H", "NA"): (434.0, 1.010),
         ("HO", "OH"): (553.0, 0.960)}
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
          ("C", "CT", "N3"): (80.0, 111.20), ("HP", "CT", "N3"): (50.0, 109.50),
          # aromatic side chains
          ("CA", "CT", "CT"): (63.0, 114.00), ("CA", "CT", "HC"): (50.0, 109.50), ("CA", "CA", "CT"): (70.0, 120.00),
          ("CA", "CA", "CA"): (63.0, 120.00), ("CA", "CA", "HA"): (50.0, 120.00), ("C", "CA", "CA"): (63.0, 120.00),
          ("CA", "C", "CA"): (63.0, 120.00), ("CA", "C", "OH"): (70.0, 120.00), ("C", "OH", "HO"): (50.0, 113.00),
          ("C", "CA", "HA"): (50.0, 120.00),
          ("C*", "CT", "CT"): (63.0, 115.60), ("C*", "CT", "HC"): (50.0, 109.50), ("CT", "C*", "CW"): (70.0, 125.00),
          ("CB", "C*", "CT"): (70.0, 128.60), ("CB", "C*", "CW"): (63.0, 106.40), ("C*", "CW", "H4"): (50.0, 120.00),
          ("C*", "CW", "NA"): (70.0, 108.70), ("H4", "CW", "NA"): (50.0, 120.00), ("CW", "NA", "H"): (50.0, 120.00),
          ("CN", "NA", "CW"): (70.0, 111.60), ("CN", "NA", "H"): (50.0, 123.10), ("CA", "CN", "NA"): (70.0, 132.80),
          ("CB", "CN", "NA"): (70.0, 104.40), ("CA", "CN", "CB"): (63.0, 122.70), ("CA", "CA", "CN"): (63.0, 120.00),
          ("CN", "CA", "HA"): (50.0, 120.00), ("CA", "CA", "CB"): (63.0, 120.00), ("CB", "CA", "HA"): (50.0, 120.00),
          ("CA", "CB", "CN"): (63.0, 116.20), ("C*", "CB", "CA"): (63.0, 134.90), ("C*", "CB", "CN"): (63.0, 108.80),
          # arginine, lysine, methionine, serine / threonine, proline
          ("CT", "CT", "N2"): (80.0, 111.20), ("H1", "CT", "N2"): (50.0, 109.50), ("CT", "N2", "H"): (50.0, 118.40),
          ("CA", "N2", "CT"): (50.0, 123.20), ("CA", "N2", "H"): (50.0, 120.00), ("H", "N2", "H"): (35.0, 120.00),
          ("N2", "CA", "N2"): (70.0, 120.00),
          ("CT", "CT", "S"): (50.0, 114.70), ("H1", "CT", "S"): (50.0, 109.50), ("CT", "S", "CT"): (62.0, 98.90),
          ("CT", "CT", "OH"): (50.0, 109.50), ("H1", "CT", "OH"): (50.0, 109.50), ("CT", "OH", "HO"): (55.0, 108.50),
          ("CT", "N", "CT"): (50.0, 118.00)}
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
    ("HO", "OH", "CT", "CT"): [(0.16, 0.0, 3), (0.25, 0.0, 1)],
    ("H1", "CT", "CT", "OH"): [(0.25, 0.0, 1)],
    ("HC", "CT", "CT", "OH"): [(0.25, 0.0, 1)],
}
_TORSION_GENERIC = {
    ("C", "N"): [(2.50, 180.0, 2)],            # X-C-N-X   10.0 / 4 paths
    ("CT", "CT"): [(1.40 / 9.0, 0.0, 3)],      # X-CT-CT-X
    ("C", "CT"): [],                           # X-C-CT-X  0.0
    ("CT", "N"): [],                           # X-CT-N-X  0.0
    ("CT", "N3"): [(1.40 / 9.0, 0.0, 3)],      # X-CT-N3-X
    ("C", "CA"): [(14.5 / 4.0, 180.0, 2)], ("CA", "CA"): [(14.5 / 4.0, 180.0, 2)], ("CA", "CB"): [(14.0 / 4.0, 180.0, 2)],
    ("CA", "CN"): [(14.5 / 4.0, 180.0, 2)], ("CA", "CT"): [], ("CA", "N2"): [(9.6 / 4.0, 180.0, 2)],
    ("CB", "CN"): [(12.0 / 4.0, 180.0, 2)], ("C*", "CB"): [(6.7 / 4.0, 180.0, 2)], ("C*", "CT"): [],
    ("C*", "CW"): [(26.1 / 4.0, 180.0, 2)], ("CN", "NA"): [(6.1 / 4.0, 180.0, 2)], ("CW", "NA"): [(6.0 / 4.0, 180.0, 2)],
    ("CT", "N2"): [], ("CT", "OH"): [(0.5 / 3.0, 0.0, 3)], ("CT", "S"): [(1.0 / 3.0, 0.0, 3)],
    ("C", "OH"): [(4.6 / 2.0, 180.0, 2)],
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
# The other three residues ff99SB-ILDN touches, FITTED the same way (r04, tools/pin_energy/fit_ildn_1hgv.py) to the forces of
# the reference's second OpenMM file - 140 frames of a 46-residue protein (3 ILE, 4 LEU, 1 ASP; testdata/output/1hgv-traj-
# arrays.npz).  Every other entry of this file meets that data at its float32 noise without any fitting, so the residual
# force there IS the difference between ILDN's series and the parm99 terms on those bonds.  Least squares over cos + sin
# coefficients up to n = 6 on ten candidate carrier dihedrals, every second frame: one carrier per bond comes out non-zero, all
# phases within 0.004 degrees of 0 / 180 (written as such), the generic X-CT-CT-X term on the carrier cancels exactly (ILE) -
# and the held-out frames are met to 0.0016 kJ/mol/nm rms (forces) and 0.0011 kJ/mol (energies, ABSOLUTE: in
# PeriodicTorsionForce's form k (1 + cos(n phi - phase)) no additive constant is needed).  The two asparagines of that protein
# meet the asparagine series above at the same level: an independent check of the r02 / r04 fit on another molecule.
# What the data cannot say: terms whose dihedral the protein's frames never move (none seen: the held-out residual is noise).
_ILDN_FITTED_TORSIONS = {
    "ASN": _ASN_FITTED_TORSIONS,
    "ILE": {("N", "CA", "CB", "CG2"): [(0.19510, 0.0, 1), (0.84590, 180.0, 2)]},
    "LEU": {("C", "CA", "CB", "CG"): [(0.57143, 0.0, 1), (0.35831, 180.0, 2), (0.13480, 0.0, 3)]},
    "ASP": {("N", "CA", "CB", "CG"): [(2.63493, 180.0, 1), (1.19037, 180.0, 2), (0.00687, 180.0, 3), (0.42274, 0.0, 4),
                                      (0.23170, 0.0, 5), (0.21276, 180.0, 6)],
            ("CA", "CB", "CG", "OD1"): [(0.44320, 180.0, 2), (0.13760, 180.0, 4), (0.01325, 180.0, 6)],
            ("CA", "CB", "CG", "OD2"): [(0.44320, 180.0, 2), (0.13760, 180.0, 4), (0.01325, 180.0, 6)]},
}
# GBSAOBCForce parameters of amber99_obc.xml: radius (nm) by element and number of bonded atoms, scale by element
_GB_SCALE = {"H": 0.85, "C": 0.72, "N": 0.79, "O": 0.85, "S": 0.96}
ELEMENT_MASSES = {"C": 12.01, "H": 1.008, "N": 14.01, "O": 16.0, "S": 32.06}
AD_MASSES = ELEMENT_MASSES


def _gb_radius(element: str, n_bonds: int, partner_element: str) -> float:
    """GBSAOBCForce radius (nm) as amber99_obc.xml assigns it: the rule of TINKER's OBC set by element, number of bonded atoms
    and - for hydrogen - the element it sits on.  Pinned by the two known-answer files for every case a complete standard
    residue produces (H on C / N / O, sp3 / sp2 carbon, 3- and 4-bonded nitrogen, carbonyl / hydroxyl oxygen, sulfur)."""
    if element == "H":
        return {"N": 0.115, "O": 0.105}.get(partner_element, 0.125)
    if element == "C":
        return {3: 0.1875, 2: 0.1825}.get(n_bonds, 0.190)
    if element == "N":
        return {4: 0.1625, 1: 0.160}.get(n_bonds, 0.1706)
    if element == "O":
        return 0.148 if n_bonds == 1 else 0.1535
    if element == "S":
        return 0.1775
    raise NotImplementedError(f"GBSA-OBC radius of element {element}")


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

# r04: the other residues of the ff94 library (charges as published; every template sums to its formal charge to 1e-4).
# Pinned by a second OpenMM known-answer file the reference holds: testdata/output/1hgv-traj-arrays.npz, 140 frames of a
# 46-residue, 691-atom protein written by the same run as the NNQQ file next to it (energies and forces;
# tools/pin_energy/pin_1hgv.py, tests/test_energy_kat.py).  HIS and CYS do not occur in it and are not offered.
_PHE_RING = "CB-HB2 CB-HB3 CB-CG CG-CD1 CD1-HD1 CD1-CE1 CE1-HE1 CE1-CZ CZ-CE2 CE2-HE2 CE2-CD2 CD2-HD2 CD2-CG "
_MET_SIDE = "CB-HB2 CB-HB3 CB-CG CG-HG2 CG-HG3 CG-SD SD-CE CE-HE1 CE-HE2 CE-HE3"
RESIDUES.update({
    "GLY": _res("N H CT H1 H1 C O", [-0.4157, 0.2719, -0.0252, 0.0698, 0.0698, 0.5973, -0.5679],
                "N H CA HA2 HA3 C O", "N-H N-CA CA-HA2 CA-HA3 CA-C C-O"),
    "CGLY": _res("N H CT H1 H1 C O2 O2", [-0.3821, 0.2681, -0.2493, 0.1056, 0.1056, 0.7231, -0.7855, -0.7855],
                 "N H CA HA2 HA3 C O OXT", "N-H N-CA CA-HA2 CA-HA3 CA-C C-O C-OXT"),
    "SER": _res("N H CT H1 CT H1 H1 OH HO C O",
                [-0.4157, 0.2719, -0.0249, 0.0843, 0.2117, 0.0352, 0.0352, -0.6546, 0.4275, 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 OG HG C O", _BB + "CB-HB2 CB-HB3 CB-OG OG-HG"),
    "THR": _res("N H CT H1 CT H1 CT HC HC HC OH HO C O",
                [-0.4157, 0.2719, -0.0389, 0.1007, 0.3654, 0.0043, -0.2438, 0.0642, 0.0642, 0.0642, -0.6761, 0.4102, 0.5973, -0.5679],
                "N H CA HA CB HB CG2 HG21 HG22 HG23 OG1 HG1 C O", _BB + "CB-HB CB-CG2 CG2-HG21 CG2-HG22 CG2-HG23 CB-OG1 OG1-HG1"),
    "VAL": _res("N H CT H1 CT HC CT HC HC HC CT HC HC HC C O",
                [-0.4157, 0.2719, -0.0875, 0.0969, 0.2985, -0.0297, -0.3192, 0.0791, 0.0791, 0.0791, -0.3192, 0.0791, 0.0791, 0.0791,
                 0.5973, -0.5679],
                "N H CA HA CB HB CG1 HG11 HG12 HG13 CG2 HG21 HG22 HG23 C O",
                _BB + "CB-HB CB-CG1 CG1-HG11 CG1-HG12 CG1-HG13 CB-CG2 CG2-HG21 CG2-HG22 CG2-HG23"),
    "LEU": _res("N H CT H1 CT HC HC CT HC CT HC HC HC CT HC HC HC C O",
                [-0.4157, 0.2719, -0.0518, 0.0922, -0.1102, 0.0457, 0.0457, 0.3531, -0.0361, -0.4121, 0.1000, 0.1000, 0.1000,
                 -0.4121, 0.1000, 0.1000, 0.1000, 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 CG HG CD1 HD11 HD12 HD13 CD2 HD21 HD22 HD23 C O",
                _BB + "CB-HB2 CB-HB3 CB-CG CG-HG CG-CD1 CD1-HD11 CD1-HD12 CD1-HD13 CG-CD2 CD2-HD21 CD2-HD22 CD2-HD23"),
    "ILE": _res("N H CT H1 CT HC CT HC HC HC CT HC HC CT HC HC HC C O",
                [-0.4157, 0.2719, -0.0597, 0.0869, 0.1303, 0.0187, -0.3204, 0.0882, 0.0882, 0.0882, -0.0430, 0.0236, 0.0236,
                 -0.0660, 0.0186, 0.0186, 0.0186, 0.5973, -0.5679],
                "N H CA HA CB HB CG2 HG21 HG22 HG23 CG1 HG12 HG13 CD1 HD11 HD12 HD13 C O",
                _BB + "CB-HB CB-CG2 CG2-HG21 CG2-HG22 CG2-HG23 CB-CG1 CG1-HG12 CG1-HG13 CG1-CD1 CD1-HD11 CD1-HD12 CD1-HD13"),
    "MET": _res("N H CT H1 CT HC HC CT H1 H1 S CT H1 H1 H1 C O",
                [-0.4157, 0.2719, -0.0237, 0.0880, 0.0342, 0.0241, 0.0241, 0.0018, 0.0440, 0.0440, -0.2737, -0.0536, 0.0684, 0.0684,
                 0.0684, 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 CG HG2 HG3 SD CE HE1 HE2 HE3 C O", _BB + _MET_SIDE),
    "NMET": _res("N3 H H H CT HP CT HC HC CT H1 H1 S CT H1 H1 H1 C O",
                 [0.1592, 0.1984, 0.1984, 0.1984, 0.0221, 0.1116, 0.0865, 0.0125, 0.0125, 0.0334, 0.0292, 0.0292, -0.2774, -0.0341,
                  0.0597, 0.0597, 0.0597, 0.6123, -0.5713],
                 "N H H2 H3 CA HA CB HB2 HB3 CG HG2 HG3 SD CE HE1 HE2 HE3 C O", _BB + "N-H2 N-H3 " + _MET_SIDE),
    "PRO": _res("N CT H1 H1 CT HC HC CT HC HC CT H1 C O",
                [-0.2548, 0.0192, 0.0391, 0.0391, 0.0189, 0.0213, 0.0213, -0.0070, 0.0253, 0.0253, -0.0266, 0.0641, 0.5896, -0.5748],
                "N CD HD2 HD3 CG HG2 HG3 CB HB2 HB3 CA HA C O",
                "N-CD CD-HD2 CD-HD3 CD-CG CG-HG2 CG-HG3 CG-CB CB-HB2 CB-HB3 CB-CA N-CA CA-HA CA-C C-O"),
    "PHE": _res("N H CT H1 CT HC HC CA CA HA CA HA CA HA CA HA CA HA C O",
                [-0.4157, 0.2719, -0.0024, 0.0978, -0.0343, 0.0295, 0.0295, 0.0118, -0.1256, 0.1330, -0.1704, 0.1430, -0.1072, 0.1297,
                 -0.1704, 0.1430, -0.1256, 0.1330, 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 CG CD1 HD1 CE1 HE1 CZ HZ CE2 HE2 CD2 HD2 C O", _BB + _PHE_RING + "CZ-HZ"),
    "TYR": _res("N H CT H1 CT HC HC CA CA HA CA HA C OH HO CA HA CA HA C O",
                [-0.4157, 0.2719, -0.0014, 0.0876, -0.0152, 0.0295, 0.0295, -0.0011, -0.1906, 0.1699, -0.2341, 0.1656, 0.3226, -0.5579,
                 0.3992, -0.2341, 0.1656, -0.1906, 0.1699, 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 CG CD1 HD1 CE1 HE1 CZ OH HH CE2 HE2 CD2 HD2 C O", _BB + _PHE_RING + "CZ-OH OH-HH"),
    "TRP": _res("N H CT H1 CT HC HC C* CW H4 NA H CN CA HA CA HA CA HA CA HA CB C O",
                [-0.4157, 0.2719, -0.0275, 0.1123, -0.0050, 0.0339, 0.0339, -0.1415, -0.1638, 0.2062, -0.3418, 0.3412, 0.1380, -0.2601,
                 0.1572, -0.1134, 0.1417, -0.1972, 0.1447, -0.2387, 0.1700, 0.1243, 0.5973, -0.5679],
                "N H CA HA CB HB2 HB3 CG CD1 HD1 NE1 HE1 CE2 CZ2 HZ2 CH2 HH2 CZ3 HZ3 CE3 HE3 CD2 C O",
                _BB + "CB-HB2 CB-HB3 CB-CG CG-CD1 CD1-HD1 CD1-NE1 NE1-HE1 NE1-CE2 CE2-CZ2 CZ2-HZ2 CZ2-CH2 CH2-HH2 CH2-CZ3 CZ3-HZ3 "
                      "CZ3-CE3 CE3-HE3 CE3-CD2 CD2-CE2 CD2-CG"),
    "ASP": _res("N H CT H1 CT HC HC C O2 O2 C O",
                [-0.5163, 0.2936, 0.0381, 0.0880, -0.0303, -0.0122, -0.0122, 0.7994, -0.8014, -0.8014, 0.5366, -0.5819],
                "N H CA HA CB HB2 HB3 CG OD1 OD2 C O", _BB + "CB-HB2 CB-HB3 CB-CG CG-OD1 CG-OD2"),
    "GLU": _res("N H CT H1 CT HC HC CT HC HC C O2 O2 C O",
                [-0.5163, 0.2936, 0.0397, 0.1105, 0.0560, -0.0173, -0.0173, 0.0136, -0.0425, -0.0425, 0.8054, -0.8188, -0.8188,
                 0.5366, -0.5819],
                "N H CA HA CB HB2 HB3 CG HG2 HG3 CD OE1 OE2 C O", _BB + "CB-HB2 CB-HB3 CB-CG CG-HG2 CG-HG3 CG-CD CD-OE1 CD-OE2"),
    "LYS": _res("N H CT H1 CT HC HC CT HC HC CT HC HC CT HP HP N3 H H H C O",
                [-0.3479, 0.2747, -0.2400, 0.1426, -0.0094, 0.0362, 0.0362, 0.0187, 0.0103, 0.0103, -0.0479, 0.0621, 0.0621, -0.0143,
                 0.1135, 0.1135, -0.3854, 0.3400, 0.3400, 0.3400, 0.7341, -0.5894],
                "N H CA HA CB HB2 HB3 CG HG2 HG3 CD HD2 HD3 CE HE2 HE3 NZ HZ1 HZ2 HZ3 C O",
                _BB + "CB-HB2 CB-HB3 CB-CG CG-HG2 CG-HG3 CG-CD CD-HD2 CD-HD3 CD-CE CE-HE2 CE-HE3 CE-NZ NZ-HZ1 NZ-HZ2 NZ-HZ3"),
    "ARG": _res("N H CT H1 CT HC HC CT HC HC CT H1 H1 N2 H CA N2 H H N2 H H C O",
                [-0.3479, 0.2747, -0.2637, 0.1560, -0.0007, 0.0327, 0.0327, 0.0390, 0.0285, 0.0285, 0.0486, 0.0687, 0.0687, -0.5295,
                 0.3456, 0.8076, -0.8627, 0.4478, 0.4478, -0.8627, 0.4478, 0.4478, 0.7341, -0.5894],
                "N H CA HA CB HB2 HB3 CG HG2 HG3 CD HD2 HD3 NE HE CZ NH1 HH11 HH12 NH2 HH21 HH22 C O",
                _BB + "CB-HB2 CB-HB3 CB-CG CG-HG2 CG-HG3 CG-CD CD-HD2 CD-HD3 CD-NE NE-HE NE-CZ CZ-NH1 NH1-HH11 NH1-HH12 CZ-NH2 "
                      "NH2-HH21 NH2-HH22"),
})

AD_ATOM_NAMES = "HH31 CH3 HH32 HH33 C O N H CA HA CB HB1 HB2 HB3 C O N H CH3 HH31 HH32 HH33".split()
AD_RESIDUES = ["ACE"] * 6 + ["ALA"] * 10 + ["NME"] * 6


def _neighbours(n: int, bonds: Sequence[Tuple[int, int]]) -> List[List[int]]:
    nb: List[List[int]] = [[] for _ in range(n)]
    for i, j in bonds:
        nb[i].append(j)
        nb[j].append(i)
    return nb


# parm99 impropers as OpenMM's amber99sb.xml holds them: centre type -> [(type2, type3, type4, k kcal/mol)], None = wildcard.
# (AMBER writes the centre third: X-X-C-O, X-O2-C-O2, CA-CA-C-OH, X-X-N-H, C-CT-N-H (ff99SB backbone, 1.1), X-CT-N-CT,
#  X-X-N2-H, X-X-NA-H, X-X-CA-HA, X-N2-CA-N2, CA-CA-CA-CT, X-X-CW-H4, CW-CB-C*-CT.)
_IMPROPER_PATTERNS = {
    "C": [(None, None, "O", 10.5), (None, "O2", "O2", 10.5), ("CA", "CA", "OH", 1.1)],
    "N": [(None, None, "H", 1.0), ("C", "CT", "H", 1.1), (None, "CT", "CT", 1.0)],
    "N2": [(None, None, "H", 1.0)], "NA": [(None, None, "H", 1.0)],
    "CA": [(None, None, "HA", 1.1), (None, "N2", "N2", 10.5), ("CA", "CA", "CT", 1.1)],
    "CW": [(None, None, "H4", 1.1)], "C*": [("CW", "CB", "CT", 1.1)],
}
_PERMUTATIONS3 = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))


def _improper(centre: int, nb: List[int], ty: List[str], el: List[str], neighbour_order: str = "sorted"):
    """OpenMM's placement of the AMBER impropers for an sp2 centre with three neighbours: (a1, a2, centre, a4) and k in
    kcal/mol, or None - a restatement of openmm/app/forcefield.py `_matchImproper`: the first permutation of the neighbours
    that fits a pattern gives a4 (the atom in the pattern's last slot); the other two go carbon first, else heavier element
    first, same element by index; a specific pattern beats a wildcard one.  Confirmed per class against the known-answer
    forces of both files.  `neighbour_order`: the order OpenMM walks the neighbours in - "sorted" (by atom index: OpenMM 7.6
    and later, what the 40-frame file pins) or "pyset" (iteration order of a Python set of the indices, i.e. by index mod 8
    for three neighbours: OpenMM 7.4, what the reference's 2021 files were written with; tests only)."""
    pats = _IMPROPER_PATTERNS.get(ty[centre])
    if pats is None:
        return None
    order = sorted(nb) if neighbour_order == "sorted" else list(set(nb))
    match = None
    for t2, t3, t4, k in pats:
        wild = t2 is None or t3 is None
        if match is not None and wild:
            continue
        for perm in _PERMUTATIONS3:
            x2, x3, x4 = (order[i] for i in perm)
            if (t2 is None or ty[x2] == t2) and (t3 is None or ty[x3] == t3) and ty[x4] == t4:
                a1, a2 = x2, x3
                if el[a1] == el[a2]:
                    if a1 > a2:
                        a1, a2 = a2, a1
                elif el[a1] != "C" and (el[a2] == "C" or ELEMENT_MASSES[el[a1]] < ELEMENT_MASSES[el[a2]]):
                    a1, a2 = a2, a1
                match = ((a1, a2, centre, x4), k)
                break
    return match


def amber99sbildn_obc_tables(atom_names: Sequence[str], residue_names: Sequence[str],
                             residue_ids: Sequence[int], family: str = "amber99",
                             improper_neighbour_order: str = "sorted") -> ForceFieldTables:
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
                if fam["asn_fitted"] and local[b] in _ILDN_FITTED_TORSIONS and \
                        residue_ids[a] == residue_ids[b] == residue_ids[c] == residue_ids[d]:
                    nm4 = (atom_names[a], atom_names[b], atom_names[c], atom_names[d])
                    fitted = _ILDN_FITTED_TORSIONS[local[b]]
                    terms = fitted.get(nm4) or fitted.get(nm4[::-1])
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
            imp = _improper(c, nb[c], [parent(t) for t in ty], el, improper_neighbour_order)
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
