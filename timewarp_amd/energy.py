"""Potential-energy callable with the contract of `OpenmmPotentialEnergyTorch`
(utils/openmm/openmm_bridge.py:252-307): `energy(coords[..., V, 3]) -> [N, 1]` in kJ/mol on the
input's device/dtype, attribute `kbT` (kJ/mol).  The arithmetic is the HIP kernel behind
`tw_amber_energy`; there is no CPU path."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .forcefield import ForceFieldTables, alanine_dipeptide_amber99sb, tables_from_openmm_system

GAS_CONSTANT = 8.314462618e-3  # kJ/(mol K), as openmm.unit.MOLAR_GAS_CONSTANT_R


_UNPINNED_WARNED = [False]


class AmberPotentialEnergyTorch:
    # "pinned": the tables behind this object are held to OpenMM known-answer files of the reference (amber99sb-ildn + OBC-II, all
    # 18 residue types of its two test systems) or come from the caller's own OpenMM System; "unpinned": see from_preset
    parity = "pinned"

    def __init__(self, tables: ForceFieldTables, temperature: float = 310.0, integrator=None, md_preset: str = None):
        self.tables = tables
        # simulation preset the topology's dataset was made with (simulation/md.py:31-37): decides the integrator scheme of
        # sample_with_model(sim="device").  Unknown origin (from_openmm without a usable integrator): "unknown", which runs
        # LangevinMiddleIntegrator - what every preset of the reference uses except the oldest amber99 one (md.py:213-231); an
        # integrator handed to from_openmm() overrides the preset with its own scheme / step / friction (`md_integrator`).
        self.md_preset = md_preset or ("amber14-implicit" if tables.has_gbsa == 2 else "unknown")
        self.md_integrator = self._integrator_parameters(integrator)
        self.temperature = float(temperature)
        self.num_particles = tables.n_atoms  # openmm_bridge.py:279
        self._integrator = integrator
        self._dev = {}

    @staticmethod
    def _integrator_parameters(integrator):
        """(class name, step size ps, friction 1/ps) of an OpenMM Langevin integrator object, or None when it is absent or not
        one of the two schemes the device integrator implements (the device route then falls back to the preset)."""
        if integrator is None:
            return None
        from .forcefield import _md

        name = type(integrator).__name__
        if name not in ("LangevinMiddleIntegrator", "LangevinIntegrator") or not (hasattr(integrator, "getStepSize") and hasattr(integrator, "getFriction")):
            return None
        return name, _md(integrator.getStepSize()), _md(integrator.getFriction())

    @classmethod
    def alanine_dipeptide(cls, temperature: float = 310.0) -> "AmberPotentialEnergyTorch":
        """Preset `alanine-dipeptide` of simulation/md.py:31-37,75-82 (310 K, 2 nm cutoff, OBC)."""
        return cls(alanine_dipeptide_amber99sb(), temperature, md_preset="amber99-implicit-old")

    @classmethod
    def from_preset(cls, preset_or_dataset: str, atom_names, residue_names, residue_ids, temperature: float = 310.0):
        """The energy of `get_system(model, preset)` (simulation/md.py:128-187) without OpenMM: `preset_or_dataset` as in
        md.py:31-37 ("alanine-dipeptide", "T1-peptides", "T1B-peptides", ...), the topology as per-atom names, residue
        names and residue ids.  amber99sb-ildn + OBC II is pinned by the reference's known-answer file; amber14 + OBC I
        ("T1B-peptides", the 4AA preset) is PARITY UNPINNED and limited to ACE / NME / ALA / GLY (forcefield.py)."""
        from .forcefield import tables_for_preset

        from .forcefield import PRESET_FAMILY

        md_preset = {"T1B-peptides": "amber14-implicit", "T1-peptides": "amber99-implicit-old", "HP-1400": "amber99-implicit-old",
                     "HP-4000": "amber99-implicit-old", "alanine-dipeptide": "amber99-implicit-old"}.get(preset_or_dataset, preset_or_dataset)
        energy = cls(tables_for_preset(preset_or_dataset, atom_names, residue_names, residue_ids), temperature, md_preset=md_preset)
        if PRESET_FAMILY.get(preset_or_dataset) == "amber14":
            # no vector anywhere in the reference can pin these tables (no OpenMM here, no known-answer file for this preset):
            # say so on the object and once per process, instead of letting the number pass for a checked one
            energy.parity = "unpinned"
            if not _UNPINNED_WARNED[0]:
                _UNPINNED_WARNED[0] = True
                import warnings

                warnings.warn("timewarp_amd: the amber14-all + implicit/obc1 tables (preset of T1B-peptides / 4AA datasets) are PARITY "
                              "UNPINNED - written from the published ff14SB / OBC-I parameters for ACE / NME / ALA / GLY only, never "
                              "compared with OpenMM.  Pass your own System through AmberPotentialEnergyTorch.from_openmm for checked "
                              "energies.", RuntimeWarning, stacklevel=2)
        return energy

    @classmethod
    def from_openmm(cls, system, integrator=None, platform_name=None, platform_properties=None, **_):
        """Same positional arguments as OpenmmPotentialEnergyTorch(system, integrator, platform_name=...)
        (evaluate.py:296-301); the platform arguments are irrelevant here and ignored."""
        from .forcefield import _md

        temperature = 310.0
        if integrator is not None and hasattr(integrator, "getTemperature"):
            temperature = _md(integrator.getTemperature())  # kelvin in OpenMM's MD unit system
        return cls(tables_from_openmm_system(system), temperature, integrator)

    def get_integrator(self):
        """openmm_bridge.py:296-297: the integrator the energy object was built with (None for the built-in presets,
        which carry their temperature themselves)."""
        return self._integrator

    @property
    def kbT(self) -> float:
        """openmm_bridge.py:299-307: R * T in kJ/mol."""
        return GAS_CONSTANT * self.temperature

    def _device_ff(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = self.tables.to_device(device)
        return self._dev[key]

    @torch.no_grad()
    def energy_and_terms(self, coords: torch.Tensor, want_terms: bool = False):
        V = self.tables.n_atoms
        # the reference's two assertions (openmm_bridge.py:286-291)
        assert coords.size(-1) == 3, f"last dimension is expected to be of size 3 but it is {coords.size(-1)}"
        assert coords.size(-2) == V, f"size {coords.size()} does not align with expected number of particles {V}"
        x = _lib.require_gpu_tensor(coords.reshape(-1, V, 3), torch.float32, "coords")
        n = x.shape[0]
        dev = x.device
        ff = self._device_ff(dev)
        out = torch.empty(n, dtype=torch.float64, device=dev)
        terms = torch.empty((n, 5), dtype=torch.float64, device=dev) if want_terms else None
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.tw_amber_energy(C.byref(ff.struct), x.data_ptr(), out.data_ptr(), _lib.ptr(terms), n,
                                           _lib.stream_ptr(dev)), "tw_amber_energy")
        return out, terms

    @torch.no_grad()
    def energy_and_forces(self, coords: torch.Tensor):
        """(E [N] kJ/mol, F [N, V, 3] kJ/mol/nm), both float64: what `Context.getState(getEnergy=True, getForces=True)` gives
        for the System (simulation/md.py:292-298 sums the same forces per term).  Analytic forces of all five terms from the
        HIP kernel behind `tw_amber_energy_forces`."""
        V = self.tables.n_atoms
        assert coords.size(-1) == 3 and coords.size(-2) == V, f"size {coords.size()} does not align with {V} particles"
        x = _lib.require_gpu_tensor(coords.reshape(-1, V, 3), torch.float32, "coords")
        n, dev = x.shape[0], x.device
        ff = self._device_ff(dev)
        e = torch.empty(n, dtype=torch.float64, device=dev)
        f = torch.empty((n, V, 3), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().tw_amber_energy_forces(C.byref(ff.struct), x.data_ptr(), e.data_ptr(), f.data_ptr(), n,
                                                          _lib.stream_ptr(dev)), "tw_amber_energy_forces")
        return e, f

    def __call__(self, coords: torch.Tensor) -> torch.Tensor:
        e, _ = self.energy_and_terms(coords)
        return e.to(coords.dtype)[:, None]

    forward = __call__
    energy = __call__
