"""Several independent Markov chains evaluated in lock-step on one GPU (an extension -
the reference runs one chain, utils/evaluation_utils.py:517).

Each iteration draws S proposals for every one of the C chains and pushes all C*S rows through ONE flow reverse
pass, one energy evaluation and ONE flow forward pass (rows ordered sample-major, row = s*C + c, which is what the
reference's `[S, B]` reshape produces and what `tw_flow_sample_with_logp_multi` implements); the accept test runs
per chain (`tw_mh_accept_chains`, one workgroup per chain) and moves every chain's state on the device.  Per chain
the arithmetic, the order of its random draws and the emitted rows are exactly those of
`sample_with_model(..., num_proposal_steps=S)` on that chain alone, including the last-iteration clip
`k = min(k, N - i)`; tests/test_mh_gpu.py checks this bit for bit."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .. import _lib
from .evaluation_utils import (ChainStats, DeviceNoise, RecordingNoise, ReplayDraws, _deferred, _range_guarded,
                               check_symmetry_change, compute_kinetic_energy)


def _accept_chains(energy, p_xy, p_yx, u, y_c, y_v, x_c, x_v):
    S, Cn = energy.shape
    dev = energy.device
    ex = torch.empty((S, Cn), dtype=torch.float32, device=dev)
    p_acc = torch.empty((S, Cn), dtype=torch.float32, device=dev)
    acc = torch.empty((S, Cn), dtype=torch.uint8, device=dev)
    res = torch.empty((Cn, 4), dtype=torch.int32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.tw_mh_accept_chains(energy.data_ptr(), p_xy.data_ptr(), p_yx.data_ptr(), u.data_ptr(), y_c.data_ptr(),
                                           y_v.data_ptr(), x_c.data_ptr(), x_v.data_ptr(), ex.data_ptr(), p_acc.data_ptr(),
                                           acc.data_ptr(), res.data_ptr(), S, Cn, x_c.shape[-2], _lib.stream_ptr(dev)),
                   "tw_mh_accept_chains")
    return ex, p_acc, acc, res


class MetropolisHastingsChains:
    """C chains of one molecule type (same atom types, masses and energy function), each with its own state and
    its own noise source."""

    KEYS = ("ind", "acc", "pxy", "pyx", "exp", "epot", "ekin", "dpot", "dkin")

    def __init__(self, batches: Sequence, model, device, energy_fn, masses, num_proposal_steps: int,
                 random_velocs: bool = False, resample_velocs: bool = False, reference_signs=None,
                 chirality_centers=None, noises: Optional[Sequence] = None):
        self.device = device = torch.device(device)
        self.C = C = len(batches)
        assert C >= 1 and all(b.atom_coords.size(0) == 1 for b in batches)
        self.model, self.energy_fn, self.S = model, energy_fn, int(num_proposal_steps)
        self.noises = list(noises) if noises is not None else [DeviceNoise(device) for _ in range(C)]
        assert len(self.noises) == C
        # split-fp16 range guard (modules/flow.py): keep the draws since the last read-back, see flush()
        self._guard = _range_guarded(model)
        if self._guard:
            self.noises = [RecordingNoise(n) for n in self.noises]
        f32 = torch.float32
        self.random_velocs, self.resample_velocs = random_velocs, resample_velocs
        self.x_coords = torch.cat([b.atom_coords.to(device, f32) for b in batches], dim=0).contiguous()
        self.x_velocs = torch.cat([n.randn_like(b.atom_coords.to(device, f32)) if random_velocs else b.atom_velocs.to(device, f32)
                                   for n, b in zip(self.noises, batches)], dim=0).contiguous()
        self.atom_types = torch.cat([b.atom_types.to(device) for b in batches], dim=0)
        self.masked = torch.cat([b.masked_elements.to(device) for b in batches], dim=0)
        self.masses = masses.to(device, f32)
        self.V = self.x_coords.shape[1]
        self.use_chirality = chirality_centers is not None and reference_signs is not None
        self.chirality_centers, self.reference_signs = chirality_centers, reference_signs
        self.kbT = energy_fn.kbT
        self.sgn = 1.0 if random_velocs else -1.0
        self.chain_c = [[self.x_coords[c:c + 1].clone()] for c in range(C)]
        self.chain_v = [[self.x_velocs[c:c + 1].clone()] for c in range(C)]
        self.rec = [{k: [] for k in self.KEYS} for _ in range(C)]
        self.emitted = [0] * C
        self.accepted = [0] * C
        self.proposals = 0
        self._pending = []

    def step_deferred(self) -> None:
        """One iteration of every chain, no host synchronisation."""
        S, C, V, dev = self.S, self.C, self.V, self.device
        model, kbT = self.model, self.kbT
        if not self._pending:
            self._pending_start = (self.x_coords, self.x_velocs)
            if self._guard:
                for n in self.noises:
                    n.mark()
        x_c, x_v = self.x_coords, self.x_velocs
        if self.random_velocs and self.resample_velocs:
            x_v = torch.cat([n.randn_like(x_v[c:c + 1]) for c, n in enumerate(self.noises)], dim=0)
        sc = torch.exp(model.coords_prior_log_scale.detach()).to(dev)
        sv = torch.exp(model.velocs_prior_log_scale.detach()).to(dev)
        lat = [n.latents(S, 1, V, sc, sv) for n in self.noises]
        z_c = torch.cat([l[0] for l in lat], dim=1).contiguous()
        z_v = torch.cat([l[1] for l in lat], dim=1).contiguous()
        with _deferred(model):  # the range flag is looked at in flush(), where the results are read back
            y_c, y_v, p_xy = model.conditional_sample_with_logp(
                atom_types=self.atom_types, x_coords=x_c, x_velocs=x_v, adj_list=None, edge_batch_idx=None,
                masked_elements=self.masked, num_samples=S, z_coords=z_c, z_velocs=z_v, allow_multi=True)
        rows_c, rows_v = y_c.reshape(S * C, V, 3), y_v.reshape(S * C, V, 3)
        e_pot_x = (self.energy_fn(x_c) / kbT).reshape(C)
        e_kin_x = compute_kinetic_energy(x_v, self.masses, random_velocs=self.random_velocs, kbT=kbT)
        e_kin_y = compute_kinetic_energy(rows_v, self.masses, random_velocs=self.random_velocs, kbT=kbT).reshape(S, C)
        e_pot_y = (self.energy_fn(rows_c) / kbT).reshape(S, C)
        if self.use_chirality:
            changed = check_symmetry_change(rows_c, self.chirality_centers, self.reference_signs).reshape(S, C)
            e_pot_y = torch.where(changed, e_pot_y + 2000, e_pot_y)
        e_kin = e_kin_y - e_kin_x[None]
        e_pot = e_pot_y - e_pot_x[None]
        energy = (e_pot + e_kin).contiguous()
        sgn = self.sgn
        with _deferred(model):
            p_yx = model.log_likelihood(
                atom_types=self.atom_types.repeat(S, 1), y_coords=x_c.repeat(S, 1, 1), y_velocs=(sgn * x_v).repeat(S, 1, 1),
                x_coords=rows_c, x_velocs=sgn * rows_v, adj_list=None, edge_batch_idx=None,
                masked_elements=self.masked.repeat(S, 1)).reshape(S, C).contiguous()
        p_xy = p_xy.reshape(S, C).contiguous()
        self.proposals += S * C
        u = torch.stack([n.uniform(S).to(dev, torch.float32) for n in self.noises], dim=1).contiguous()
        new_c, new_v = x_c.clone(), x_v.clone()
        ex, p_acc, acc, res = _accept_chains(energy, p_xy, p_yx, u, y_c.contiguous(), y_v.contiguous(), new_c, new_v)
        self._pending.append((res, x_c, x_v, new_c, new_v, acc,
                              (("acc", p_acc), ("pxy", p_xy), ("pyx", p_yx), ("exp", ex), ("epot", e_pot_y),
                               ("ekin", e_kin_y), ("dpot", e_pot), ("dkin", e_kin))))
        self.x_coords, self.x_velocs = new_c, new_v

    def flush(self, num_samples: Optional[int] = None) -> None:
        """Bookkeeping of the parked iterations.  With `num_samples`, a chain that has emitted that many states
        ignores further iterations and its last counted iteration gets the reference's clip."""
        if not self._pending:
            return
        results = torch.stack([p[0] for p in self._pending]).cpu().tolist()  # [iterations][C][4]
        # (`demoted`: someone else read - and cleared - the device's sticky flag while these iterations were parked)
        if self._guard and (bool(getattr(self.model, "demoted", False)) or self.model.split_fp16_overflowed(self.device)):
            # the model's activations left the fp16 range: the parked iterations again, on the exact-f32 kernels, from
            # their starting states and with the recorded draws; the chains then continue there
            self.model.demote_to_f32()
            n_iter, recorders = len(self._pending), self.noises
            self.noises = [ReplayDraws(r.log) for r in recorders]
            self.x_coords, self.x_velocs = self._pending_start
            self.proposals -= n_iter * self.S * self.C
            self._pending, self._guard = [], False
            try:
                for _ in range(n_iter):
                    self.step_deferred()
            finally:
                self.noises = [r.inner for r in recorders]
            results = torch.stack([p[0] for p in self._pending]).cpu().tolist()
        V = self.V
        for per_chain, (_, old_c, old_v, new_c, new_v, acc, per_proposal) in zip(results, self._pending):
            for c, (k_true, any_acc, _, _) in enumerate(per_chain):
                if num_samples is not None and self.emitted[c] >= num_samples:
                    continue
                self.accepted[c] += int(any_acc)
                k = k_true if num_samples is None else min(k_true, num_samples - self.emitted[c])
                moved = bool(any_acc) and k == k_true
                oc, ov = old_c[c:c + 1], old_v[c:c + 1]
                if k > 0:
                    self.chain_c[c].append(oc.expand(k, V, 3))
                    self.chain_v[c].append(ov.expand(k, V, 3))
                self.chain_c[c].append(new_c[c:c + 1] if moved else oc)
                self.chain_v[c].append(new_v[c:c + 1] if moved else ov)
                self.rec[c]["ind"].append(acc[: k + 1, c].bool())
                for name, t in per_proposal:
                    self.rec[c][name].append(t[: k + 1, c])
                self.emitted[c] += k + 1
        self._pending = []

    def results(self) -> List:
        """Per chain: (coords [1+n,V,3] numpy, velocs, accepted, ChainStats) as `sample_with_model` returns them."""
        self.flush()
        out = []
        for c in range(self.C):
            stats = ChainStats(*[torch.cat(self.rec[c][k], dim=0).cpu().numpy() for k in self.KEYS])
            out.append((torch.cat(self.chain_c[c], dim=0).cpu().numpy(), torch.cat(self.chain_v[c], dim=0).cpu().numpy(),
                        self.accepted[c], stats))
        return out


def sample_with_model_chains(batches: Sequence, model, device, openmm_potential_energy_torch, masses, num_samples: int,
                             num_proposal_steps: int, random_velocs: bool = False, resample_velocs: bool = False,
                             reference_signs=None, chirality_centers=None, noises: Optional[Sequence] = None,
                             sync_every: int = 8) -> List:
    """Run len(batches) chains (Metropolis-Hastings with `num_proposal_steps` parallel proposals each) until every
    one has emitted `num_samples` states; per chain the result equals `sample_with_model(batch, ..., num_samples,
    accept=True, num_proposal_steps=...)` driven by the same noise source."""
    chains = MetropolisHastingsChains(batches, model, device, openmm_potential_energy_torch, masses, num_proposal_steps,
                                      random_velocs=random_velocs, resample_velocs=resample_velocs,
                                      reference_signs=reference_signs, chirality_centers=chirality_centers, noises=noises)
    with torch.no_grad():
        while min(chains.emitted) < num_samples:
            for _ in range(max(1, sync_every)):
                chains.step_deferred()
            chains.flush(num_samples)
    return chains.results()
