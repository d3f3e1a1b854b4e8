"""Batched Metropolis-Hastings sampling with the flow as proposal: the sampling half of the
reference's utils/evaluation_utils.py (`sample_with_model` :468-745, `compute_kinetic_energy`
:416-436, `compute_num_proposal_steps` :32-64, `ChainStats` :67-114), same names, arguments and
return values.  Every numeric step of an iteration runs on the GPU through libtimewarp_hip.so:

    proposals + log p(y|x)      tw_flow_sample_with_logp      (model.conditional_sample_with_logp)
    potential energies          tw_amber_energy               (the energy callable)
    kinetic energies            tw_kinetic_energy
    chirality guard             tw_chirality_changed
    log p(x|y) of the reverse   tw_flow_log_likelihood        (model.log_likelihood)
    exponent, p_acc, u < p_acc, first accepted index          tw_mh_accept

When the proposal is the HIP flow and the energy the AMBER kernel, the whole list above is ONE call, tw_mh_iteration
(csrc/tw_mh_step.hip; same bits as the op-by-op route, which serves every other combination).
The host reads back 8 bytes per iteration (first accepted index, any-accepted flag) - batched over
`sync_every` iterations, the accept kernel moving the chain state on the device meanwhile; chain states
and ChainStats stay on the device until the loop ends (the reference does >= 10 D2H copies per
iteration).  Two redundancies of the reference are not reproduced because they cannot change the
result: E_pot and E_kin of the current state are evaluated once instead of on S identical copies
(evaluation_utils.py:620-629)."""
from __future__ import annotations

import ctypes as C
import pickle
from dataclasses import astuple, dataclass
from typing import Optional

import numpy as np
import torch
from tqdm.auto import tqdm

from .. import _lib


def compute_num_proposal_steps(current_acceptance_probability: float, target_acceptance_per_step: float = 0.9,
                               max_num_proposal_steps: int = 100) -> int:
    """Number of parallel proposals so that at least one is accepted with the target probability:
    ceil(log(1-target) / log(1-p)), p clipped to [1e-3, 1-1e-3], result in [1, max]."""
    p_rej = min(max(1 - current_acceptance_probability, 1e-3), 1 - 1e-3)
    with np.errstate(all="ignore"):
        wanted = np.nan_to_num(np.log(1 - target_acceptance_per_step) / np.log(p_rej), nan=np.inf)
    return max(int(np.ceil(min(wanted, max_num_proposal_steps))), 1)


@dataclass
class ChainStats:
    """Per-emitted-state statistics of one chain (same nine fields as the reference)."""

    acceptance_indicator: np.ndarray
    acceptance: np.ndarray
    p_xy: np.ndarray
    p_yx: np.ndarray
    exponent: np.ndarray
    energies_pot: np.ndarray
    energies_kin: np.ndarray
    energies_pot_delta: np.ndarray
    energies_kin_delta: np.ndarray

    def __len__(self):
        return len(self.acceptance)

    def __getitem__(self, key):
        return ChainStats(*map(lambda x: x[key], astuple(self)))

    def thin(self, step):
        return ChainStats(*map(lambda x: x[0: x.shape[0]: step], astuple(self)))

    def save(self, path):
        with open(path, "wb") as f:
            pickle.dump(self, f)

    @staticmethod
    def load(path):
        with open(path, "rb") as f:
            return pickle.load(f)


def compute_kinetic_energy(velocs: torch.Tensor, masses: torch.Tensor, random_velocs: bool = False,
                           kbT: Optional[float] = None) -> torch.Tensor:
    """0.5 * sum v^2 (isotropic-Gaussian velocities) or 0.5 * sum m v^2 / kbT; [batch]."""
    if not random_velocs:
        assert kbT, "Requires kbT to compute energy"
    v = _lib.require_gpu_tensor(velocs, torch.float32, "velocs")
    n, V = v.shape[0], v.shape[1]
    m = _lib.require_gpu_tensor(masses.to(v.device).reshape(-1), torch.float32, "masses")
    out = torch.empty(n, dtype=torch.float32, device=v.device)
    lib = _lib.load()
    with torch.cuda.device(v.device):
        _lib.check(lib.tw_kinetic_energy(v.data_ptr(), m.data_ptr(), int(random_velocs), float(kbT or 0.0),
                                         out.data_ptr(), n, V, _lib.stream_ptr(v.device)), "tw_kinetic_energy")
    return out


def check_symmetry_change(coords: torch.Tensor, chirality_centers: torch.Tensor, reference_signs: torch.Tensor) -> torch.Tensor:
    """True where the sign of the triple product at any chirality centre differs from the
    reference (utils/chirality.py:40-80); [batch] bool."""
    x = _lib.require_gpu_tensor(coords, torch.float32, "coords")
    n, V = x.shape[0], x.shape[1]
    cen = _lib.require_gpu_tensor(chirality_centers.to(x.device), torch.int32, "chirality_centers")
    ref = _lib.require_gpu_tensor(reference_signs.to(x.device).reshape(-1), torch.float32, "reference_signs")
    out = torch.empty(n, dtype=torch.uint8, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.tw_chirality_changed(x.data_ptr(), cen.data_ptr(), ref.data_ptr(), cen.shape[0], out.data_ptr(),
                                            n, V, _lib.stream_ptr(x.device)), "tw_chirality_changed")
    return out.bool()


class DeviceNoise:
    """Random draws on the device, in the reference's order and shapes: from the default generator, or - with
    `seed` - from a private generator (one independent, reproducible stream per chain)."""

    def __init__(self, device, seed: Optional[int] = None):
        self.device = device
        self.gen = None
        self.host_gen = None  # the rotation matrix is drawn on the host (3 x 3, fp64)
        if seed is not None:
            self.gen = torch.Generator(device=device)
            self.gen.manual_seed(int(seed))
            self.host_gen = torch.Generator()
            self.host_gen.manual_seed(int(seed))

    def randn_like(self, t):
        return torch.randn(t.shape, device=t.device, dtype=t.dtype, generator=self.gen)

    def latents(self, S, B, V, scale_c, scale_v):
        zc = torch.randn((S, B, V, 3), device=self.device, generator=self.gen) * scale_c
        zv = torch.randn((S, B, V, 3), device=self.device, generator=self.gen) * scale_v
        return zc, zv

    def latents_into(self, zc, zv, S, std_c: float, std_v: float):
        """The same draws as `latents` (same generator, same order, same values: randn * std), written into the first S
        rows of caller-owned [S + 1, V, 3] buffers - what tw_mh_iteration takes."""
        zc[:S].normal_(0.0, std_c, generator=self.gen)
        zv[:S].normal_(0.0, std_v, generator=self.gen)

    def uniform(self, S):
        return torch.rand(S, device=self.device, generator=self.gen)

    def rotation(self):
        # uniform SO(3) via QR of a Gaussian matrix (the reference uses scipy's Rotation.random())
        q, r = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64, generator=self.host_gen))
        q = q * torch.sign(torch.diagonal(r))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        return q.to(torch.float32).to(self.device)


class RecordingNoise:
    """A noise source that keeps what it hands out, in draw order, so that the iterations since the last read-back can
    be redone on other kernels with the SAME random numbers (the split-fp16 range guard, modules/flow.py).  Wraps any
    object with the DeviceNoise protocol; `mark()` forgets everything drawn so far."""

    def __init__(self, inner):
        self.inner = inner
        self.log = []

    def mark(self):
        self.log = []

    def randn_like(self, t):
        r = self.inner.randn_like(t)
        self.log.append(r)
        return r

    def latents(self, S, B, V, scale_c, scale_v):
        zc, zv = self.inner.latents(S, B, V, scale_c, scale_v)
        self.log.append((zc, zv))
        return zc, zv

    def latents_into(self, zc, zv, S, std_c, std_v, scale_c=None, scale_v=None):
        if isinstance(self.inner, DeviceNoise):
            # Two device copies per iteration would be the price of keeping the values (the caller's buffers become the
            # proposals): keep the generator state in front of the draw instead and draw again on replay.
            gen = self.inner.gen if self.inner.gen is not None else _default_generator(zc.device)
            self.log.append(_RegenLatents(gen.get_state(), zc.device, S, tuple(zc.shape[1:]), float(std_c), float(std_v)))
            self.inner.latents_into(zc, zv, S, std_c, std_v)
            return
        if hasattr(self.inner, "latents_into"):
            self.inner.latents_into(zc, zv, S, std_c, std_v)
        else:  # replayed / host-drawn noise
            a, b = self.inner.latents(S, 1, zc.shape[1], scale_c, scale_v)
            zc[:S].copy_(a.reshape(S, -1, 3))
            zv[:S].copy_(b.reshape(S, -1, 3))
        self.log.append((zc[:S].clone(), zv[:S].clone()))  # the caller's buffers become the proposals

    def uniform(self, S):
        u = self.inner.uniform(S)
        self.log.append(u)
        return u

    def rotation(self):
        q = self.inner.rotation()
        self.log.append(q)
        return q


def _default_generator(device):
    device = torch.device(device)
    if device.type == "cuda":
        return torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    return torch.default_generator


class _RegenLatents:
    """A `DeviceNoise.latents_into` draw, kept as the generator state in front of it."""

    def __init__(self, state, device, S, row_shape, std_c, std_v):
        self.state, self.device, self.S, self.row_shape, self.std_c, self.std_v = state, device, S, row_shape, std_c, std_v

    def draw(self):
        g = torch.Generator(device=self.device)
        g.set_state(self.state)
        zc = torch.empty((self.S, *self.row_shape), device=self.device).normal_(0.0, self.std_c, generator=g)
        zv = torch.empty((self.S, *self.row_shape), device=self.device).normal_(0.0, self.std_v, generator=g)
        return zc, zv


class ReplayDraws:
    """Hands a RecordingNoise log back, draw by draw."""

    def __init__(self, log):
        self.log = list(log)

    def _pop(self):
        e = self.log.pop(0)
        return e.draw() if isinstance(e, _RegenLatents) else e

    def randn_like(self, t):
        return self._pop()

    def latents(self, S, B, V, scale_c, scale_v):
        zc, zv = self._pop()
        return zc.reshape(S, B, V, 3), zv.reshape(S, B, V, 3)

    def latents_into(self, zc, zv, S, std_c, std_v, scale_c=None, scale_v=None):
        a, b = self._pop()
        zc[:S].copy_(a.reshape(S, -1, 3))
        zv[:S].copy_(b.reshape(S, -1, 3))

    def uniform(self, S):
        return self._pop()

    def rotation(self):
        return self._pop()


def _range_guarded(model) -> bool:
    """The model may run on the split-fp16 kernel (and so needs the range guard's recorded draws)."""
    return hasattr(model, "split_fp16_overflowed") and not getattr(model, "demoted", False) and \
        getattr(model, "execution_path", 0) in (-1, -2, _lib.TW_PATH_FUSED_H3, _lib.TW_PATH_FUSED_H1, _lib.TW_PATH_SIMPLE_H3)


class _no_defer:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _deferred(model):
    return model.deferred_range_check() if hasattr(model, "deferred_range_check") else _no_defer()


def _mh_accept(energy, p_xy, p_yx, u, y_c=None, y_v=None, x_c=None, x_v=None):
    """tw_mh_accept.  With y_c/y_v [S,V,3] and x_c/x_v [1,V,3] the kernel also moves the chain state on the device
    (x <- y[k] if any proposal was accepted), so the host does not have to look at the result before it queues the
    next iteration."""
    S = energy.shape[0]
    dev = energy.device
    ex = torch.empty(S, dtype=torch.float32, device=dev)
    p_acc = torch.empty(S, dtype=torch.float32, device=dev)
    acc = torch.empty(S, dtype=torch.uint8, device=dev)
    res = torch.empty(4, dtype=torch.int32, device=dev)
    lib = _lib.load()
    ptr = lambda t: None if t is None else t.data_ptr()
    V = 0 if x_c is None else x_c.shape[-2]
    with torch.cuda.device(dev):
        _lib.check(lib.tw_mh_accept(energy.data_ptr(), p_xy.data_ptr(), p_yx.data_ptr(), u.data_ptr(), ptr(y_c), ptr(y_v),
                                    ptr(x_c), ptr(x_v), ex.data_ptr(), p_acc.data_ptr(), acc.data_ptr(), res.data_ptr(), S, V,
                                    _lib.stream_ptr(dev)), "tw_mh_accept")
    return ex, p_acc, acc, res


def openmm_step(sim, coords: torch.Tensor, velocs: Optional[torch.Tensor] = None, num_steps: int = 1, integrator=None):
    """`num_steps` integration steps of the caller's `openmm.app.Simulation` from (coords, velocs); returns the new
    (coords, velocs) as tensors like `coords`.  Same calls, in the same order, as the reference's function
    (evaluation_utils.py:439-466) - nothing here imports OpenMM, the Simulation object is the caller's.
    r04: a `timewarp_amd.md.LangevinDynamics` in the Simulation's place integrates on the device (analytic AMBER forces,
    OpenMM's integration schemes, its own noise stream): no host round trip."""
    from ..md import LangevinDynamics

    if isinstance(sim, LangevinDynamics):
        if velocs is None:
            if integrator is None:
                raise ValueError("either `velocs` or `integrator` needs to be specified")
            m = sim.masses.to(coords.device)
            velocs = torch.randn_like(coords) * (sim.kbT / m).sqrt()[None, :, None]   # setVelocitiesToTemperature
        return sim.step(coords, velocs, num_steps)
    sim.context.setPositions(coords.detach().cpu().numpy().squeeze(0))
    if velocs is not None:
        sim.context.setVelocities(velocs.detach().cpu().numpy().squeeze(0))
    elif integrator is not None:
        sim.context.setVelocitiesToTemperature(integrator.getTemperature())
    else:
        raise ValueError("either `velocs` or `integrator` needs to be specified")
    sim.step(num_steps)
    state = sim.context.getState(getPositions=True, getVelocities=True)
    coords_new = torch.from_numpy(np.asarray(state.getPositions(asNumpy=True)._value)).reshape(coords.shape).to(coords)
    velocs_new = torch.from_numpy(np.asarray(state.getVelocities(asNumpy=True)._value)).reshape(coords.shape).to(coords)
    return coords_new, velocs_new


_HYBRID_WARNED = [False]


def _warn_hybrid_moves_off(energy_fn) -> None:
    """openmm_on_current / openmm_on_proposal with num_openmm_steps > 0 but sim=None: the reference silently runs without the
    hybrid moves (`... and sim is not None`, evaluation_utils.py:558, 594, 623) and so does this loop - but with the HIP
    energy there IS an integrator to be had, so say so once per process."""
    from ..energy import AmberPotentialEnergyTorch

    if _HYBRID_WARNED[0] or not isinstance(energy_fn, AmberPotentialEnergyTorch):
        return
    _HYBRID_WARNED[0] = True
    import warnings

    warnings.warn("timewarp_amd: openmm_on_current / openmm_on_proposal were requested with sim=None: the hybrid moves are OFF "
                  "(as in the reference).  Pass sim='device' to integrate on the HIP force kernel, or your own Simulation.",
                  RuntimeWarning, stacklevel=4)


class MetropolisHastingsChain:
    """State and one-iteration `step()` of the loop in `sample_with_model` (reference
    evaluation_utils.py:517-745).  `sample_with_model` drives it until enough states are emitted;
    bench.py drives it for an exact number of iterations."""

    KEYS = ("ind", "acc", "pxy", "pyx", "exp", "epot", "ekin", "dpot", "dkin")

    def __init__(self, batch, model, device, energy_fn, masses, accept=False, random_velocs=False,
                 resample_velocs=False, initialize_randomly=False, num_proposal_steps=1, adaptive_parallelism=False,
                 acceptance_rate_smoothing_factor=0.01, rotate=False, reference_signs=None, chirality_centers=None,
                 noise=None, sim=None, num_openmm_steps=0, openmm_on_proposal=False, openmm_on_current=False):
        assert batch.atom_coords.size(0) == 1, "only batch-size of 1 is supported"
        self.device = device = torch.device(device)
        self.model, self.energy_fn = model, energy_fn
        self.noise = noise or DeviceNoise(device)
        # split-fp16 range guard: keep the draws since the last read-back so that those iterations can be redone on the
        # exact-f32 kernels if the model's activations turn out to leave the fp16 range (_redo_on_f32)
        self._guard = _range_guarded(model)
        self._unwrapped_noise = self.noise
        if self._guard:
            self.noise = RecordingNoise(self.noise)
        f32 = torch.float32
        self.accept, self.random_velocs, self.resample_velocs, self.rotate = accept, random_velocs, resample_velocs, rotate
        self.adaptive, self.smoothing = adaptive_parallelism, acceptance_rate_smoothing_factor
        self.x_coords = batch.atom_coords.to(device, f32).contiguous()
        self.x_velocs = (self.noise.randn_like(self.x_coords) if random_velocs
                         else batch.atom_velocs.to(device, f32).contiguous())
        self.masked = batch.masked_elements.to(device)
        self.adj_list = batch.adj_list.to(device) if batch.adj_list is not None else None
        self.ebi = batch.edge_batch_idx.to(device) if batch.edge_batch_idx is not None else None
        self.atom_types = batch.atom_types.to(device)
        self.masses = masses.to(device, f32)
        self.V = self.x_coords.shape[1]
        self.use_chirality = chirality_centers is not None and reference_signs is not None
        self.chirality_centers, self.reference_signs = chirality_centers, reference_signs
        self._const = None
        if initialize_randomly:
            print("Initializaing chain at a random point rather than a data sample.")
            yc, yv, _ = self._propose(self.noise.randn_like(self.x_coords), self.noise.randn_like(self.x_velocs), 1)
            self.x_coords, self.x_velocs = yc.squeeze(0).contiguous(), yv.squeeze(0).contiguous()
        self.kbT = energy_fn.kbT
        # OpenMM steps on the current state / on the proposal (evaluation_utils.py:556-565, 594-602, 623-626): a host
        # round trip through the caller's Simulation per iteration, so these chains take the op-by-op route
        self.sim, self.n_omm = sim, int(num_openmm_steps)
        self._sim_steps_mark = None
        if isinstance(sim, str):
            # sim="device" (opt-in; the reference has no such value): the chain integrates by itself on the HIP force kernel
            # with the reference's preset integrator for the energy's force-field family (simulation/md.py:75-93, 213-231).
            # sim=None stays what it is in the reference - the hybrid-move options are silently off
            # (`... and sim is not None`, evaluation_utils.py:556, 594, 623).
            from ..energy import AmberPotentialEnergyTorch
            from ..md import LangevinDynamics
            if sim != "device":
                raise ValueError(f"sim={sim!r}: pass an openmm.app.Simulation-like object, a LangevinDynamics, or 'device'")
            if not isinstance(energy_fn, AmberPotentialEnergyTorch):
                raise ValueError("sim='device' needs the HIP energy (AmberPotentialEnergyTorch) as energy_fn")
            # the thermostat's Gaussian stream is keyed on (seed, step, atom): every chain draws its own seed from its own
            # noise source, so independently seeded chains (distributed.chain_seed) stay independent.  NOTE: these are TWO
            # uniform draws from the chain's noise - a chain built with sim="device" is two draws ahead of one built without.
            u = self.noise.uniform(2).double().cpu()
            seed = (int(float(u[0]) * (1 << 26)) << 26) | int(float(u[1]) * (1 << 26))
            self.sim = sim = LangevinDynamics.for_energy(energy_fn, masses, seed=seed)
        if sim is None and (openmm_on_current or openmm_on_proposal) and self.n_omm > 0:
            _warn_hybrid_moves_off(energy_fn)
        self.omm_current = bool(openmm_on_current) and self.n_omm > 0 and sim is not None
        self.omm_proposal = bool(openmm_on_proposal) and self.n_omm > 0 and sim is not None
        self.velocs_std = (self.kbT / self.masses.unsqueeze(0).unsqueeze(-1)).sqrt()
        if self.omm_current:
            self.x_coords, self.x_velocs = self._openmm_on_current(self.x_coords, self.x_velocs)
        self.chain_c, self.chain_v = [self.x_coords.clone()], [self.x_velocs.clone()]
        self.rec = {k: [] for k in self.KEYS}
        self._pending = []
        self.accepted = 0
        self.proposals = 0
        self.p_bar = 1e-3  # start by proposing as many as possible
        self.s_max = num_proposal_steps
        self.S = self.s_max if not adaptive_parallelism else compute_num_proposal_steps(self.p_bar, max_num_proposal_steps=self.s_max)
        self.sgn = 1.0 if random_velocs else -1.0
        self._fused = self._fused_iteration_available()
        # r06: without a caller's noise source the fused iteration draws its latents, resampled velocities and accept uniforms
        # itself (tw_mh_iteration_chains with one chain: Philox4x32-10 keyed (seed, 0, iteration) inside its first glue kernel)
        # instead of through four ATen generator launches; the seed comes from the device's default generator, so
        # torch.cuda.manual_seed still decides the chain.  TW_MH_KERNEL_DRAWS=0: the ATen draws of r05.
        import os as _os
        self._kernel_draws = (self._fused and noise is None and not rotate and _os.environ.get("TW_MH_KERNEL_DRAWS", "1") != "0")
        self._kd_seed = int(torch.randint(0, 2 ** 62, (1,), device=device).item()) if self._kernel_draws else 0
        self._kd_iter = 0
        self._inflight = None   # flush(lag=True): the window whose read-back is under way

    # ---- whole iteration in one C-ABI call (tw_mh_iteration) ----------------------------------------------------
    def _fused_iteration_available(self) -> bool:
        """tw_mh_iteration serves the case the reference's scripts run - the HIP flow as proposal, the AMBER energy
        kernel, the accept test on - unless TW_MH_FUSED=0.  Anything else (another energy callable, accept=False,
        an ignore_conditional_velocity model) takes the op-by-op route below; both give the same numbers."""
        import os
        from ..energy import AmberPotentialEnergyTorch
        from ..modules.flow import ConditionalFlowDensityModel

        if os.environ.get("TW_MH_FUSED", "1") == "0" or not self.accept or self.omm_current or self.omm_proposal:
            return False
        return (isinstance(self.model, ConditionalFlowDensityModel) and isinstance(self.energy_fn, AmberPotentialEnergyTorch)
                and not self.model.dims.ignore_cond_velocity and self.x_coords.is_cuda
                and self.energy_fn.tables.n_atoms == self.V)

    def _fused_constants(self, S):
        if getattr(self, "_fconst", None) is None or self._fconst["S"] != S:
            dev = self.device
            lib = _lib.load()
            desc = self.model._desc(dev)   # carries the model's own range-guard word (ABI 7)
            opt = _lib.MHOptions()
            opt.random_velocs = int(self.random_velocs)
            keep = {"masses": self.masses.contiguous()}
            opt.masses = keep["masses"].data_ptr()
            opt.kbT = float(self.kbT)
            opt.n_centres = 0
            if self.use_chirality:
                keep["centres"] = self.chirality_centers.to(dev, torch.int32).contiguous()
                keep["signs"] = self.reference_signs.to(dev, torch.float32).reshape(-1).contiguous()
                opt.n_centres = int(keep["centres"].shape[0])
                opt.centres, opt.reference_signs = keep["centres"].data_ptr(), keep["signs"].data_ptr()
            need = max(lib.tw_mh_iteration_workspace_bytes(C.byref(desc), S, self.V),
                       lib.tw_mh_iteration_chains_workspace_bytes(C.byref(desc), S, 1, self.V) if self._kernel_draws else 0)
            if need < 0:
                raise RuntimeError("tw_mh_iteration_workspace_bytes failed: " + lib.tw_last_error().decode())
            self._fconst = dict(S=S, desc=desc, opt=opt, keep=keep, ws=torch.empty(int(need), dtype=torch.uint8, device=dev),
                                std_c=float(torch.exp(self.model.coords_prior_log_scale.detach())),
                                std_v=float(torch.exp(self.model.velocs_prior_log_scale.detach())))
        return self._fconst

    def _iteration_fused(self):
        """Draws + tw_mh_iteration.  Returns what `_evaluate` + `_mh_accept` return together."""
        S, V, dev = self.S, self.V, self.device
        noise, model = self.noise, self.model
        x_coords, x_velocs = self.x_coords, self.x_velocs
        if self._kernel_draws:
            return self._iteration_fused_kernel_draws()
        if self.random_velocs and self.resample_velocs:
            x_velocs = noise.randn_like(x_velocs)
        if self.rotate:
            Q = noise.rotation().to(x_coords)
            x_coords = (x_coords @ Q.T).contiguous()
            x_velocs = (x_velocs @ Q.T).contiguous()
        fc = self._fused_constants(S)
        at, mk, _, _, sc, sv = self._constants(S)
        zc = torch.empty((S + 1, V, 3), dtype=torch.float32, device=dev)
        zv = torch.empty((S + 1, V, 3), dtype=torch.float32, device=dev)
        if isinstance(noise, (RecordingNoise, ReplayDraws)):
            noise.latents_into(zc, zv, S, fc["std_c"], fc["std_v"], sc, sv)
        elif hasattr(noise, "latents_into"):
            noise.latents_into(zc, zv, S, fc["std_c"], fc["std_v"])
        else:  # replayed / host-drawn noise
            a, b = noise.latents(S, 1, V, sc, sv)
            zc[:S].copy_(a.reshape(S, V, 3))
            zv[:S].copy_(b.reshape(S, V, 3))
        u = noise.uniform(S).to(dev, torch.float32).contiguous()
        path = model._path_for(V)
        raw, packed = model._weights(dev, path)
        ff = self.energy_fn._device_ff(dev)
        new_c = torch.empty_like(x_coords)
        new_v = torch.empty_like(x_velocs)
        stats = torch.empty((8, S), dtype=torch.float32, device=dev)
        acc = torch.empty(S, dtype=torch.uint8, device=dev)
        res = torch.empty(4, dtype=torch.int32, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(lib.tw_mh_iteration(
                C.byref(fc["desc"]), raw.data_ptr(), _lib.ptr(packed), path, C.byref(ff.struct), C.byref(fc["opt"]),
                at.data_ptr(), mk.data_ptr(), V, x_coords.data_ptr(), x_velocs.data_ptr(), zc.data_ptr(), zv.data_ptr(),
                u.data_ptr(), new_c.data_ptr(), new_v.data_ptr(), stats.data_ptr(), acc.data_ptr(), res.data_ptr(), S,
                fc["ws"].data_ptr(), fc["ws"].numel(), _lib.stream_ptr(dev)), "tw_mh_iteration")
        self.proposals += S
        per_proposal = tuple(zip(("acc", "pxy", "pyx", "exp", "epot", "ekin", "dpot", "dkin"), stats.unbind(0)))
        return x_coords, x_velocs, zc[:S], zv[:S], new_c, new_v, acc, res, per_proposal

    def _iteration_fused_kernel_draws(self):
        """tw_mh_iteration_chains with ONE chain and its own draws: nothing is drawn through ATen, the call's first glue
        kernel generates (and the read-back never sees) the latents, the resampled velocities and the uniforms."""
        S, V, dev = self.S, self.V, self.device
        model = self.model
        x_coords, x_velocs = self.x_coords, self.x_velocs
        fc = self._fused_constants(S)
        at, mk, _, _, _, _ = self._constants(S)
        f32 = torch.float32
        zc = torch.empty((S + 1, V, 3), dtype=f32, device=dev)
        zv = torch.empty((S, V, 3), dtype=f32, device=dev)
        u = torch.empty(S, dtype=f32, device=dev)
        cur_v = torch.empty_like(x_velocs)
        new_c, new_v = torch.empty_like(x_coords), torch.empty_like(x_velocs)
        stats = torch.empty((8, S), dtype=f32, device=dev)
        acc = torch.empty(S, dtype=torch.uint8, device=dev)
        res = torch.empty(4, dtype=torch.int32, device=dev)
        draws = _lib.MHDraws(self._kd_seed, self._kd_iter, 0, int(self.random_velocs and self.resample_velocs))
        self._kd_iter += 1
        path = model._path_for(V)
        raw, packed = model._weights(dev, path)
        ff = self.energy_fn._device_ff(dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().tw_mh_iteration_chains(
                C.byref(fc["desc"]), raw.data_ptr(), _lib.ptr(packed), path, C.byref(ff.struct), C.byref(fc["opt"]), C.byref(draws),
                at.data_ptr(), mk.data_ptr(), V, x_coords.data_ptr(), x_velocs.data_ptr(), cur_v.data_ptr(), zc.data_ptr(),
                zv.data_ptr(), u.data_ptr(), new_c.data_ptr(), new_v.data_ptr(), stats.data_ptr(), acc.data_ptr(), res.data_ptr(),
                S, 1, fc["ws"].data_ptr(), fc["ws"].numel(), _lib.stream_ptr(dev)), "tw_mh_iteration_chains")
        self.proposals += S
        per_proposal = tuple(zip(("acc", "pxy", "pyx", "exp", "epot", "ekin", "dpot", "dkin"), stats.unbind(0)))
        return x_coords, cur_v, zc[:S], zv, new_c, new_v, acc, res, per_proposal

    def _constants(self, S):
        """Per-chain constants in the dtypes / shapes the C ABI takes (int32 atom types, uint8 mask, both also
        repeated S times for the reverse-move likelihood) and the prior scales: built once instead of per iteration."""
        if self._const is None or self._const[0] != S:
            at = self.atom_types.to(torch.int32).contiguous()
            mk = self.masked.to(torch.uint8).contiguous()
            sc = torch.exp(self.model.coords_prior_log_scale.detach()).to(self.device)
            sv = torch.exp(self.model.velocs_prior_log_scale.detach()).to(self.device)
            self._const = (S, at, mk, at.expand(S, self.V).contiguous(), mk.expand(S, self.V).contiguous(), sc, sv)
        return self._const[1:]

    def _propose(self, xc, xv, S):
        at, mk, _, _, sc, sv = self._constants(S)
        zc, zv = self.noise.latents(S, 1, self.V, sc, sv)
        return self.model.conditional_sample_with_logp(
            atom_types=at, x_coords=xc, x_velocs=xv, adj_list=self.adj_list, edge_batch_idx=self.ebi,
            masked_elements=mk, num_samples=S, z_coords=zc, z_velocs=zv)

    def _openmm_on_current(self, x_coords, x_velocs):
        if self.random_velocs:  # the velocities are standard-normal draws: scaled for the integrator, kept as they are
            c, _ = openmm_step(self.sim, x_coords, x_velocs * self.velocs_std, num_steps=self.n_omm)
            return c.contiguous(), x_velocs
        c, v = openmm_step(self.sim, x_coords, x_velocs, num_steps=self.n_omm)
        return c.contiguous(), v.contiguous()

    def _evaluate(self):
        """Everything of one iteration up to the accept test: proposals, energies, both log-likelihoods.
        Returns the (possibly resampled / rotated) current state and the per-proposal quantities."""
        S, V = self.S, self.V
        model, noise, kbT = self.model, self.noise, self.kbT
        x_coords, x_velocs = self.x_coords, self.x_velocs
        if self.random_velocs and self.resample_velocs:
            x_velocs = noise.randn_like(x_velocs)
        if self.omm_current:
            x_coords, x_velocs = self._openmm_on_current(x_coords, x_velocs)
        if self.rotate:
            # the reference's (Q @ x.T).T raises for a [1,V,3] tensor; this applies the intended rotation
            Q = noise.rotation().to(x_coords)
            x_coords = (x_coords @ Q.T).contiguous()
            x_velocs = (x_velocs @ Q.T).contiguous()

        y_c, y_v, p_xy = self._propose(x_coords, x_velocs, S)
        y_c, y_v = y_c.squeeze(1), y_v.squeeze(1)
        if self.omm_proposal:  # one proposal per iteration (openmm_step squeezes dimension 0), as in the reference
            y_c, _ = openmm_step(self.sim, y_c, y_v * self.velocs_std, num_steps=self.n_omm)
            y_c = y_c.contiguous()
        # current state: one evaluation broadcast over the S proposals
        e_pot_x = (self.energy_fn(x_coords) / kbT).squeeze(-1)
        e_kin_x = compute_kinetic_energy(x_velocs, self.masses, random_velocs=self.random_velocs, kbT=kbT)
        e_kin_y = compute_kinetic_energy(y_v, self.masses, random_velocs=self.random_velocs, kbT=kbT)
        e_pot_y = (self.energy_fn(y_c) / kbT).squeeze(-1)
        if self.use_chirality:
            changed = check_symmetry_change(y_c, self.chirality_centers, self.reference_signs)
            e_pot_y = torch.where(changed, e_pot_y + 2000, e_pot_y)
        e_kin = e_kin_y - e_kin_x
        e_pot = e_pot_y - e_pot_x
        energy = (e_pot + e_kin).contiguous()

        _, _, at_s, mk_s, _, _ = self._constants(S)
        # reverse move: velocities negated unless they are treated as resampled Gaussians (evaluation_utils.py:648-657)
        rx_v, ry_v = (x_velocs, y_v) if self.sgn == 1.0 else (-x_velocs, -y_v)
        p_yx = model.log_likelihood(
            atom_types=at_s, y_coords=x_coords.expand(S, V, 3), y_velocs=rx_v.expand(S, V, 3), x_coords=y_c, x_velocs=ry_v,
            adj_list=self.adj_list, edge_batch_idx=self.ebi, masked_elements=mk_s)
        p_xy = p_xy.reshape(S).contiguous()
        self.proposals += S
        return x_coords, x_velocs, y_c, y_v, energy, p_xy, p_yx, e_pot_y, e_kin_y, e_pot, e_kin

    def _emit(self, k, old_c, old_v, new_c, new_v, ind, per_proposal):
        """Chain rows of one iteration: k copies of the old state, then the new one; statistics rows 0..k."""
        V = self.V
        if k > 0:
            self.chain_c.append(old_c.expand(k, V, 3))
            self.chain_v.append(old_v.expand(k, V, 3))
        self.chain_c.append(new_c)
        self.chain_v.append(new_v)
        self.rec["ind"].append(ind[: k + 1].bool())
        for name, t in per_proposal:
            self.rec[name].append(t[: k + 1])

    # ---- one iteration = _compute (device work, no host read) + bookkeeping once the accept result is on the host
    def _compute(self):
        """Everything of one MH iteration up to and including the accept kernel, which also writes the next chain state
        (x <- y[k] if a proposal was accepted) into buffers of its own.  No host synchronisation.  Returns
        (res, x_coords, x_velocs, new_c, new_v, acc, per_proposal); `res` int32[4] = {first accepted index or S-1, any}."""
        S, device = self.S, self.device
        if self._fused:
            x_coords, x_velocs, _, _, new_c, new_v, acc, res, per_proposal = self._iteration_fused()
            return res, x_coords, x_velocs, new_c, new_v, acc, per_proposal
        with _deferred(self.model):  # the range flag is looked at where the accept result is read back
            x_coords, x_velocs, y_c, y_v, energy, p_xy, p_yx, e_pot_y, e_kin_y, e_pot, e_kin = self._evaluate()
        u = self.noise.uniform(S).to(device, torch.float32).contiguous()
        new_c, new_v = x_coords.clone(), x_velocs.clone()  # becomes y[k] inside the kernel if a proposal is accepted
        ex, p_acc, acc, res = _mh_accept(energy, p_xy, p_yx, u, y_c.contiguous(), y_v.contiguous(), new_c, new_v)
        return res, x_coords, x_velocs, new_c, new_v, acc, (
            ("acc", p_acc), ("pxy", p_xy), ("pyx", p_yx), ("exp", ex), ("epot", e_pot_y), ("ekin", e_kin_y),
            ("dpot", e_pot), ("dkin", e_kin))

    def _overflowed(self) -> bool:
        """Split-fp16 range guard, asked right after a host read-back (which has synchronised).  The flag word is the MODEL's
        own (tw_flow_desc.range_flag, ABI 7), read and cleared by whoever of its users looks first: if another user of the
        same model (another chain's flush, a public call) has seen it and demoted the model while this chain still has
        iterations recorded, those iterations may be the ones that overflowed - `model.demoted` therefore counts as an
        overflow for an open window.  Another model's overflow on the same device no longer shows up here."""
        return self._guard and (bool(getattr(self.model, "demoted", False)) or self.model.split_fp16_overflowed(self.device))

    def _mark(self):
        """Start of a range-guard window: forget the draws recorded so far, remember where the device integrator stands."""
        self.noise.mark()
        self._sim_steps_mark = getattr(self.sim, "steps_done", None)

    def _redo_on_f32(self, start_c, start_v, n_iterations: int, kd_iter: Optional[int] = None):
        """The model's activations left the fp16 range somewhere in the last `n_iterations` iterations: demote the model
        to the exact-f32 kernels and run those iterations again from their starting state with the recorded draws.
        Returns their `_compute` tuples; the chain continues on the f32 kernels with its ordinary noise source."""
        self.model.demote_to_f32()
        recorder = self.noise
        self.noise = ReplayDraws(recorder.log)
        self.x_coords, self.x_velocs = start_c, start_v
        if kd_iter is not None:
            self._kd_iter = kd_iter   # the kernel's own draws are keyed on the iteration counter: the same draws again
        if hasattr(self.sim, "steps_done") and self._sim_steps_mark is not None:
            self.sim.steps_done = self._sim_steps_mark   # the device integrator's draws are keyed on the step count: same draws
        self.proposals -= n_iterations * self.S
        outs = []
        try:
            for _ in range(n_iterations):
                out = self._compute()
                outs.append(out)
                self.x_coords, self.x_velocs = out[3], out[4]
        finally:
            self.noise = recorder.inner
            self._guard = False
        return outs

    def step(self, remaining: Optional[int] = None) -> int:
        """One MH iteration; returns the number of chain states emitted (k + 1).  `remaining`
        = num_samples - i applies the reference's clip `k = min(k, N - i)` (:680)."""
        self.flush()
        S, device = self.S, self.device
        start_c, start_v, start_kd = self.x_coords, self.x_velocs, self._kd_iter
        if self._guard:
            self._mark()
        if self.accept:
            out = self._compute()
            k_true, any_acc = (int(v) for v in out[0][:2].tolist())  # the one host sync of the iteration
            if self._overflowed():
                out, = self._redo_on_f32(start_c, start_v, 1, start_kd)
                k_true, any_acc = (int(v) for v in out[0][:2].tolist())
            _, x_coords, x_velocs, new_c, new_v, acc, per_proposal = out
            self.accepted += int(any_acc)
            k = k_true if remaining is None else min(k_true, remaining)  # NB: N - i, not N - i - 1
            moved = bool(any_acc) and k == k_true
            self.p_bar = self.smoothing * (1 - (not any_acc)) + (1 - self.smoothing) ** k * self.p_bar
            if self.adaptive:
                self.S = compute_num_proposal_steps(self.p_bar, max_num_proposal_steps=self.s_max)
            if not moved:  # nothing accepted, or the clip cut the emitted rows short of the accepted proposal
                new_c, new_v = x_coords.clone(), x_velocs.clone()
            self._emit(k, x_coords, x_velocs, new_c, new_v, acc, per_proposal)
            self.x_coords, self.x_velocs = new_c.contiguous(), new_v.contiguous()
            return k + 1
        if S != 1:
            raise ValueError("Number of proposals has to be one if everything is accepted!")
        with _deferred(self.model):
            ev = self._evaluate()
        if self._overflowed():  # synchronises
            self.model.demote_to_f32()
            self.noise, self._guard = ReplayDraws(self.noise.log), False
            self.x_coords, self.x_velocs = start_c, start_v
            self.proposals -= S
            try:
                ev = self._evaluate()
            finally:
                self.noise = self._unwrapped_noise
        x_coords, x_velocs, y_c, y_v, energy, p_xy, p_yx, e_pot_y, e_kin_y, e_pot, e_kin = ev
        ex = energy + p_xy - p_yx
        p_acc = torch.clamp(torch.exp(-ex), max=1.0)
        self.accepted += 1
        acc = torch.ones(1, dtype=torch.bool, device=device)
        new_c, new_v = y_c[0:1].clone(), y_v[0:1].clone()
        self._emit(0, x_coords, x_velocs, new_c, new_v, acc,
                   (("acc", p_acc), ("pxy", p_xy), ("pyx", p_yx), ("exp", ex), ("epot", e_pot_y), ("ekin", e_kin_y),
                    ("dpot", e_pot), ("dkin", e_kin)))
        self.x_coords, self.x_velocs = new_c.contiguous(), new_v.contiguous()
        return 1

    # ---- deferred bookkeeping: the accept kernel moves the state on the device, the host reads the results later
    def can_defer(self) -> bool:
        """Deferred iterations need the accept test, a fixed proposal count and no per-iteration host decision."""
        return self.accept and not self.adaptive and not self.rotate and not (self.omm_current or self.omm_proposal)

    def step_deferred(self) -> None:
        """One MH iteration without a host synchronisation: the accept kernel writes x <- y[k] itself and the
        iteration's device tensors are parked until `flush()`.  Identical results to `step()` as long as the
        clip `k = min(k, N - i)` cannot bind (the caller keeps N - i > S)."""
        assert self.can_defer()
        if not self._pending:
            self._pending_start = (self.x_coords, self.x_velocs, self._kd_iter)
            if self._guard:
                self._mark()
        out = self._compute()
        self._pending.append(out)
        self.x_coords, self.x_velocs = out[3], out[4]

    def can_lag(self) -> bool:
        """flush(lag=True) is honoured when a replay needs no recorded draws: the kernel's own generator (a replay runs the
        same counters again) or a model the range guard does not watch."""
        return self._kernel_draws or not self._guard

    def inflight_states_max(self) -> int:
        """The most chain states the window whose read-back is under way can still emit (for the caller's clip arithmetic)."""
        return len(self._inflight["pending"]) * self.S if self._inflight is not None else 0

    def _snapshot(self):
        """Start the read-back of the parked window - 4 ints per iteration and the range-guard word, into pinned memory
        behind the window's kernels - and hand the window over."""
        res = torch.stack([p[0] for p in self._pending])
        host = torch.empty(res.shape, dtype=res.dtype, pin_memory=True)
        host.copy_(res, non_blocking=True)
        flag_host = None
        if self._guard:
            key = self.device.index if self.device.index is not None else torch.cuda.current_device()
            flag = getattr(self.model, "_range_flags", {}).get(key)
            if flag is not None:
                flag_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                flag_host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        win = dict(pending=self._pending, start=self._pending_start, host=host, flag=flag_host, ev=ev)
        self._pending = []
        return win

    def flush(self, lag: bool = False) -> int:
        """Read back the parked iterations (one D2H copy for all of them) and do their bookkeeping; returns the
        number of chain states booked by this call.
        lag=True (r06): only START the read-back of the window just queued and book the window BEFORE it, whose results
        arrived long ago: the host never waits for the device, the device never waits for the host's bookkeeping.  Counters and
        the trajectory then trail the queue by one window until a plain flush()."""
        if lag and self.can_lag():
            newer = self._snapshot() if self._pending else None
            older, self._inflight = self._inflight, newer
            return self._finish(older) if older is not None else 0
        older, self._inflight = self._inflight, None
        n = self._finish(older) if older is not None else 0
        if self._pending:
            n += self._finish(self._snapshot())
        return n

    def _finish(self, win) -> int:
        """Book one window whose read-back was started; on a range-guard trip redo it and everything queued behind it."""
        win["ev"].synchronize()
        pending, results = win["pending"], win["host"].tolist()
        tripped = self._guard and (bool(getattr(self.model, "demoted", False)) or (win["flag"] is not None and int(win["flag"][0]) != 0))
        if tripped:
            torch.cuda.synchronize(self.device)
            self.model.split_fp16_overflowed(self.device)   # clears the model's word
            behind = (len(self._inflight["pending"]) if self._inflight is not None else 0) + len(self._pending)
            self._inflight, self._pending = None, []
            start_c, start_v, start_kd = win["start"]
            pending = self._redo_on_f32(start_c, start_v, len(pending) + behind, start_kd)
            results = torch.stack([p[0] for p in pending]).cpu().tolist()
        emitted = 0
        for (k_true, any_acc, _, _), (_, old_c, old_v, new_c, new_v, acc, per_proposal) in zip(results, pending):
            self.accepted += int(any_acc)
            self.p_bar = self.smoothing * (1 - (not any_acc)) + (1 - self.smoothing) ** k_true * self.p_bar
            self._emit(k_true, old_c, old_v, new_c, new_v, acc, per_proposal)
            emitted += k_true + 1
        return emitted

    def trajectory(self):
        """Device tensors: coords [1+n,V,3], velocs [1+n,V,3]."""
        self.flush()
        return torch.cat(self.chain_c, dim=0), torch.cat(self.chain_v, dim=0)

    def result(self):
        self.flush()
        c, v = self.trajectory()
        stats = ChainStats(*[torch.cat(self.rec[k], dim=0).cpu().numpy() for k in self.KEYS])
        return c.cpu().numpy(), v.cpu().numpy(), self.accepted, stats


def sample_with_model(
    batch,
    model,
    device: torch.device,
    openmm_potential_energy_torch,
    masses: torch.Tensor,
    num_samples: int,
    accept: bool = False,
    random_velocs: bool = False,
    resample_velocs: bool = False,
    initialize_randomly: bool = False,
    num_openmm_steps: int = 0,
    sim=None,
    openmm_on_proposal: bool = False,
    openmm_on_current: bool = False,
    num_proposal_steps: int = 1,
    adaptive_parallelism: bool = False,
    acceptance_rate_smoothing_factor: float = 0.01,
    rotate: bool = False,
    reference_signs: Optional[torch.Tensor] = None,
    chirality_centers: Optional[torch.Tensor] = None,
    disable_tqdm: Optional[bool] = False,
    noise=None,
    sync_every: int = 8,
):
    """Run one Markov chain of (at least) `num_samples` states.

    Arguments, semantics and returns follow the reference function; `noise` (extension) supplies
    the random draws (default: the device generator); `sync_every` (extension) is the number of iterations
    queued per host synchronisation (1 = read the accept result every iteration).  `sim` is the caller's
    `openmm.app.Simulation` (only its context / step / getState calls are used: `openmm_step`).  Returns
    (sampled_coords [1+n,V,3] float32 numpy, sampled_velocs, accepted:int, ChainStats)."""
    chain = MetropolisHastingsChain(
        batch, model, device, openmm_potential_energy_torch, masses, accept=accept, random_velocs=random_velocs,
        resample_velocs=resample_velocs, initialize_randomly=initialize_randomly, num_proposal_steps=num_proposal_steps,
        adaptive_parallelism=adaptive_parallelism, acceptance_rate_smoothing_factor=acceptance_rate_smoothing_factor,
        rotate=rotate, reference_signs=reference_signs, chirality_centers=chirality_centers, noise=noise, sim=sim,
        num_openmm_steps=num_openmm_steps, openmm_on_proposal=openmm_on_proposal, openmm_on_current=openmm_on_current)
    print("Sample with the model using Metropolis Hastings" if accept else "Sample with the model by accepting every setp")
    i = 0
    pbar = tqdm(total=num_samples, disable=disable_tqdm)
    # Iterations are queued `sync_every` at a time without a host synchronisation while the clip k = min(k, N - i)
    # cannot bind (every iteration emits at most S states); the tail runs one synchronous iteration at a time.
    # Replayed noise (tests) is consumed exactly as recorded, so it always takes the synchronous path.
    defer = sync_every > 1 and noise is None and chain.can_defer()
    lag = defer and chain.can_lag()   # r06: book window k - 1 while the device runs window k
    with torch.no_grad():
        while i < num_samples:
            # (a window in flight can still emit up to S states per iteration: they count against the room the clip needs)
            if defer and num_samples - i - chain.inflight_states_max() > sync_every * chain.S:
                for _ in range(sync_every):
                    chain.step_deferred()
                n = chain.flush(lag=lag)
            else:
                n = chain.flush()       # settle what is in flight: i is exact again
                if n == 0:
                    n = chain.step(num_samples - i)
            i += n
            pbar.update(n)
    pbar.close()
    return chain.result()


def sample_on_batches(batches, model, device, openmm_potential_energy_torch, data_augmentation, masses,
                      random_velocs: bool = False, noise=None):
    """One-step acceptance statistics over dataset batches (reference utils/evaluation_utils.py:190-333):
    for every batch (one conditioning state each) draw one proposal y ~ p(.|x), evaluate log p(y|x),
    log p(x~|y~) (velocities negated unless `random_velocs`), the energy change, the MH acceptance
    probability, and the forward / reverse likelihood of the dataset's own target pair.  Same argument
    order and the same eleven return arrays as the reference; `noise` (extension) injects the random
    draws, as in `sample_with_model`."""
    from ..dataloader import DenseMolDynBatch, transform_batch

    device = torch.device(device)
    noise = noise or DeviceNoise(device)
    f32 = torch.float32
    masses = masses.to(device, f32)
    kbT = openmm_potential_energy_torch.kbT
    cols = {k: [] for k in ("y_c", "y_v", "t_c", "t_v", "c_c", "c_v", "p_xy", "p_yx", "p_xy_tr", "p_yx_tr", "acc")}
    sgn = 1.0 if random_velocs else -1.0
    with torch.no_grad():
        for batch in tqdm(batches):
            if data_augmentation:
                assert isinstance(batch, DenseMolDynBatch)
                batch = transform_batch(batch)
            x_c = batch.atom_coords.to(device, f32).contiguous()
            y_t = batch.atom_coord_targets.to(device, f32).contiguous()
            if random_velocs:
                x_v = noise.randn_like(x_c)
                w_t = noise.randn_like(y_t)
            else:
                x_v = batch.atom_velocs.to(device, f32).contiguous()
                w_t = batch.atom_veloc_targets.to(device, f32).contiguous()
            at = batch.atom_types.to(device)
            mk = batch.masked_elements.to(device)
            adj = batch.adj_list.to(device) if batch.adj_list is not None else None
            ebi = batch.edge_batch_idx.to(device) if batch.edge_batch_idx is not None else None
            B, V = x_c.shape[0], x_c.shape[1]
            sc = torch.exp(model.coords_prior_log_scale.detach()).to(device)
            sv = torch.exp(model.velocs_prior_log_scale.detach()).to(device)
            z_c, z_v = noise.latents(1, B, V, sc, sv)
            kw = dict(atom_types=at, adj_list=adj, edge_batch_idx=ebi, masked_elements=mk)

            def model_calls():
                y_c, y_v, _ = model.conditional_sample_with_logp(x_coords=x_c, x_velocs=x_v, num_samples=1, z_coords=z_c,
                                                                 z_velocs=z_v, **kw)
                y_c, y_v = y_c.squeeze(0).contiguous(), y_v.squeeze(0).contiguous()
                p_xy = model.log_likelihood(x_coords=x_c, x_velocs=x_v, y_coords=y_c, y_velocs=y_v, **kw)
                e_kin = (compute_kinetic_energy(y_v, masses, random_velocs=random_velocs, kbT=kbT)
                         - compute_kinetic_energy(x_v, masses, random_velocs=random_velocs, kbT=kbT))
                e_pot = ((openmm_potential_energy_torch(y_c) - openmm_potential_energy_torch(x_c)) / kbT).view(-1)
                assert e_kin.shape == e_pot.shape
                energy = e_pot + e_kin
                p_yx = model.log_likelihood(x_coords=y_c, x_velocs=(sgn * y_v).contiguous(), y_coords=x_c,
                                            y_velocs=(sgn * x_v).contiguous(), **kw)
                assert energy.shape == p_xy.shape and p_yx.shape == p_xy.shape
                p_acc = torch.clamp(torch.exp(-(energy + p_xy - p_yx)), max=1.0)
                p_xy_tr = model.log_likelihood(x_coords=x_c, x_velocs=x_v, y_coords=y_t, y_velocs=w_t, **kw)
                p_yx_tr = model.log_likelihood(x_coords=y_t, x_velocs=(sgn * w_t).contiguous(), y_coords=x_c,
                                               y_velocs=(sgn * x_v).contiguous(), **kw)
                return y_c, y_v, p_xy, p_yx, p_acc, p_xy_tr, p_yx_tr

            # five model calls per batch: no range-flag read-back (a device synchronisation) after each of them - one look
            # where the results are copied to the host anyway; on an fp16 range overflow the model is demoted and this
            # batch's calls run again on the exact-f32 kernels with the same draws
            with _deferred(model):
                out = model_calls()
            if hasattr(model, "split_fp16_overflowed") and model.split_fp16_overflowed(device):
                model.demote_to_f32()
                out = model_calls()
            y_c, y_v, p_xy, p_yx, p_acc, p_xy_tr, p_yx_tr = out
            for k, t in (("acc", p_acc), ("p_xy", p_xy), ("p_yx", p_yx), ("p_xy_tr", p_xy_tr), ("p_yx_tr", p_yx_tr),
                         ("y_c", y_c), ("y_v", y_v), ("c_c", x_c), ("c_v", x_v)):
                cols[k].append(t.cpu().numpy())
            cols["t_c"].append(batch.atom_coord_targets.cpu().numpy())
            cols["t_v"].append(batch.atom_veloc_targets.cpu().numpy())
    arr = {k: np.array(v) for k, v in cols.items()}
    sq = lambda a: a.squeeze(1)  # the reference assumes one conditioning state per batch here (:315-320)
    return (sq(arr["y_c"]), sq(arr["y_v"]), sq(arr["t_c"]), sq(arr["t_v"]), sq(arr["c_c"]), sq(arr["c_v"]),
            arr["p_yx"], arr["p_xy"], arr["p_yx_tr"], arr["p_xy_tr"], arr["acc"])


def sample_on_single_conditional(batch, model, num_samples, sim, step_width, random_velocs, device, noise=None):
    """`num_samples` model samples y ~ p(.|x) and `num_samples` OpenMM segments of `step_width` steps, all from the single
    conditioning state in `batch` (reference utils/evaluation_utils.py:356-413, called from evaluate.py:576).  Same
    arguments and the same five return arrays; `sim` is the caller's `openmm.app.Simulation` (its context / integrator /
    step calls, in the reference's order - nothing here imports OpenMM).  `noise` (extension) injects the random draws,
    as in `sample_with_model`; without it the velocity draw and the latents come from the device generator."""
    device = torch.device(device)
    positions, velocities, y_coords_model, y_velocs_model = [], [], [], []
    at = batch.atom_types.to(device)
    x_c = batch.atom_coords.to(device, torch.float32).contiguous()
    mk = batch.masked_elements.to(device)
    adj = batch.adj_list.to(device) if batch.adj_list is not None else None
    ebi = batch.edge_batch_idx.to(device) if batch.edge_batch_idx is not None else None
    B, V = x_c.shape[0], x_c.shape[1]
    with torch.no_grad():
        for _ in tqdm(range(num_samples)):
            sim.context.setPositions(batch.atom_coords.numpy().squeeze(0))
            if random_velocs:
                sim.context.setVelocitiesToTemperature(sim.integrator.getTemperature())
                sim.context.getState(getPositions=True, getVelocities=True)
                # drawn ON THE DEVICE (x_c has the shape and lives there): a seeded DeviceNoise owns a device generator, which
                # cannot fill the CPU tensor batch.atom_velocs.  The reference draws torch.randn_like on the
                # host here (evaluation_utils.py:384): seeded runs reproduce this library's stream, not the reference's
                x_v = (noise.randn_like(x_c) if noise is not None
                       else torch.randn(x_c.shape, device=device)).to(device, torch.float32)
            else:
                sim.context.setVelocities(batch.atom_velocs.numpy().squeeze(0))
                x_v = batch.atom_velocs.to(device, torch.float32)
            kw = dict(atom_types=at, x_coords=x_c, x_velocs=x_v.contiguous(), adj_list=adj, edge_batch_idx=ebi,
                      masked_elements=mk, num_samples=1)
            if noise is not None:
                sc = torch.exp(model.coords_prior_log_scale.detach()).to(device)
                sv = torch.exp(model.velocs_prior_log_scale.detach()).to(device)
                z_c, z_v = noise.latents(1, B, V, sc, sv)
                y_c, y_v, _ = model.conditional_sample_with_logp(z_coords=z_c, z_velocs=z_v, **kw)
            else:
                y_c, y_v = model.conditional_sample(**kw)
            sim.step(step_width)
            state = sim.context.getState(getPositions=True, getVelocities=True)
            positions.append(state.getPositions(asNumpy=True)._value)
            velocities.append(state.getVelocities(asNumpy=True)._value)
            y_coords_model.append(y_c.detach().cpu().numpy())
            y_velocs_model.append(y_v.detach().cpu().numpy())
    return (np.array(y_coords_model).squeeze(1).squeeze(1), np.array(y_velocs_model).squeeze(1).squeeze(1),
            np.array(positions), np.array(velocities), np.array(batch.atom_coords.numpy()))
