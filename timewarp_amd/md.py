"""Langevin dynamics on the HIP force kernel: the device-side stand-in for the `openmm.app.Simulation` the reference's
hybrid moves drive (`openmm_step`, utils/evaluation_utils.py:439-466; integrators of simulation/md.py:213-231 with the
preset parameters of md.py:75-93: 310 K, friction 0.3 / ps, time step 0.5 fs, LangevinMiddleIntegrator - LangevinIntegrator
for the oldest datasets).

`sample_with_model(..., openmm_on_current / openmm_on_proposal, num_openmm_steps=n, sim=LangevinDynamics(...))` - or
`sim="device"` with an `AmberPotentialEnergyTorch` energy, for which the chain builds one (the preset's integrator, a seed
drawn from the chain's noise; `sim=None` turns the options off, as in the reference) - advances states on the GPU without
the host round trip an OpenMM Simulation costs per iteration.  The integration schemes are OpenMM's; the Gaussian noise is
this library's own counter-based stream, so trajectories agree with OpenMM's statistically (temperature, energy
conservation without friction), not step for step.  There is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from .energy import AmberPotentialEnergyTorch

SCHEMES = {"LangevinMiddleIntegrator": 0, "LangevinIntegrator": 1}


class LangevinDynamics:
    def __init__(self, energy: AmberPotentialEnergyTorch, masses: torch.Tensor, timestep_ps: float = 0.0005,
                 friction_per_ps: float = 0.3, integrator: str = "LangevinMiddleIntegrator", seed: int = 0,
                 temperature: Optional[float] = None):
        if integrator not in SCHEMES:
            raise ValueError(f"integrator {integrator!r}: expected one of {sorted(SCHEMES)}")
        self.energy = energy
        self.masses = masses.detach().to(torch.float32).reshape(-1).contiguous()
        if self.masses.numel() != energy.tables.n_atoms:
            raise ValueError("one mass per atom")
        self.dt, self.friction, self.scheme = float(timestep_ps), float(friction_per_ps), SCHEMES[integrator]
        self.integrator = integrator
        self.kbT = energy.kbT if temperature is None else 8.314462618e-3 * float(temperature)
        self.seed = int(seed) & (2 ** 64 - 1)
        self.steps_done = 0
        self._m = {}

    def _masses_on(self, device):
        key = str(device)
        if key not in self._m:
            self._m[key] = self.masses.to(device)
        return self._m[key]

    @torch.no_grad()
    def step(self, coords: torch.Tensor, velocs: torch.Tensor, num_steps: int = 1, want_energy: bool = False):
        """`num_steps` steps of every conformation: coords (nm), velocs (nm/ps) [..., V, 3] -> new (coords, velocs) like the
        inputs (and, with `want_energy`, the potential energy [N] at the last force evaluation)."""
        V = self.energy.tables.n_atoms
        x = _lib.require_gpu_tensor(coords.reshape(-1, V, 3), torch.float32, "coords").clone()
        v = _lib.require_gpu_tensor(velocs.reshape(-1, V, 3), torch.float32, "velocs").clone()
        n, dev = x.shape[0], x.device
        ff = self.energy._device_ff(dev)
        e = torch.empty(n, dtype=torch.float64, device=dev) if want_energy else None
        with torch.cuda.device(dev):
            _lib.check(_lib.load().tw_langevin_steps(
                C.byref(ff.struct), self._masses_on(dev).data_ptr(), x.data_ptr(), v.data_ptr(), int(num_steps), self.dt,
                self.friction, self.kbT, self.scheme, self.seed, self.steps_done, _lib.ptr(e), n, _lib.stream_ptr(dev)),
                "tw_langevin_steps")
        self.steps_done += int(num_steps)
        out = (x.reshape(coords.shape).to(coords.dtype), v.reshape(velocs.shape).to(velocs.dtype))
        return out + (e,) if want_energy else out

    @classmethod
    def from_preset(cls, energy: AmberPotentialEnergyTorch, masses: torch.Tensor, preset: str = "amber14-implicit", seed: int = 0):
        """simulation/md.py:75-93: both presets run 310 K, 0.3 / ps, 0.5 fs; "amber99-implicit-old" uses LangevinIntegrator,
        everything else (the other presets, energies of unknown origin) LangevinMiddleIntegrator (md.py:213-231)."""
        integ = "LangevinIntegrator" if preset == "amber99-implicit-old" else "LangevinMiddleIntegrator"
        return cls(energy, masses, 0.0005, 0.3, integ, seed)

    @classmethod
    def for_energy(cls, energy: AmberPotentialEnergyTorch, masses: torch.Tensor, seed: int = 0):
        """The integrator of the energy object: the scheme, step size and friction of the OpenMM integrator it was built with
        (`AmberPotentialEnergyTorch.from_openmm(system, integrator)`), else its dataset preset's."""
        if getattr(energy, "md_integrator", None) is not None:
            name, dt, friction = energy.md_integrator
            return cls(energy, masses, dt, friction, name, seed)
        return cls.from_preset(energy, masses, preset=energy.md_preset, seed=seed)
