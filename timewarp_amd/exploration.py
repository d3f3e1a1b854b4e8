"""Exploration mode: `num_parallel_steps` independent explorers that always move to the model's proposal unless the potential
energy rises by more than a threshold or a chirality centre flips - no Metropolis-Hastings correction (reference
exploration.py:120-262, the loop at :236-252).

`explore` is that loop on the device: one `conditional_sample` launch for all explorers (each is its own conditioning state,
`num_samples = 1`), the AMBER energy kernel, the chirality kernel and two `torch.where` - nothing crosses to the host inside
the loop; the caller copies the trajectory once at the end (the script's `np.savez(positions=..., time=...)`, :259-260).
Like `sample_trajectory.py`, this is the loop without the script's argument parsing and checkpoint loading: the reference's
own script runs unchanged on this package's model and energy after `integration.install()` (its loop is written inline in
`main`, so it then drives the same kernels call by call).
"""
from __future__ import annotations

import torch

from .utils.chirality import compute_chirality_sign, find_chirality_centers
from .utils.evaluation_utils import (DeviceNoise, RecordingNoise, ReplayDraws, _deferred, _range_guarded,
                                     check_symmetry_change)

CHIRALITY_PENALTY = 10000.0  # exploration.py:243
RANGE_CHECK_WINDOW = 64   # exploration steps per look at the split-fp16 range flag (and per recorded window of draws)


def explore(batch, model, device, openmm_potential_energy_torch, num_steps: int, num_parallel_steps: int = 1,
            energy_threshold: float = 300.0, noise=None):
    """exploration.py:219-257.  `batch`: one conditioning state (B = 1).  Returns (positions [num_steps * P, V, 3],
    energies [num_steps * P, 1]) as device tensors: after every step the current state of all P explorers, in step order
    (`torch.cat(trajectory_exploration, axis=0)`).  `noise` (extension, as in `sample_with_model`) supplies the latent and
    velocity draws; by default they come from the device generator."""
    device = torch.device(device)
    noise = noise or DeviceNoise(device)
    P = int(num_parallel_steps)
    f32 = torch.float32
    centers = find_chirality_centers(batch.adj_list, batch.atom_types)                  # :229
    signs = compute_chirality_sign(batch.atom_coords, centers) if len(centers) else None  # :230
    y_coords = batch.atom_coords.to(device, f32)                                          # :232
    y_velocs = batch.atom_velocs.to(device, f32)
    energies = openmm_potential_energy_torch(y_coords).repeat(P, 1)                       # :236-237
    y_coords = y_coords.repeat(P, 1, 1).contiguous()
    y_velocs = y_velocs.repeat(P, 1, 1).contiguous()
    V = y_coords.shape[1]
    kw = dict(atom_types=batch.atom_types.repeat(P, 1).to(device), adj_list=batch.adj_list,
              edge_batch_idx=batch.edge_batch_idx.to(device) if batch.edge_batch_idx is not None else None,
              masked_elements=batch.masked_elements.repeat(P, 1).to(device))              # :122-131
    sc = torch.exp(model.coords_prior_log_scale.detach()).to(device)
    sv = torch.exp(model.velocs_prior_log_scale.detach()).to(device)
    def run_window(noise, state, n):
        """n exploration steps from `state` = (y_c, y_v, e); returns the new state and the steps' trajectory / energy lists."""
        y_c, y_v, e = state
        trajectory, energy_log = [], []
        for _ in range(n):
            z_c, z_v = noise.latents(1, P, V, sc, sv)
            y_new, _, _ = model.conditional_sample_with_logp(x_coords=y_c, x_velocs=y_v, num_samples=1, z_coords=z_c,
                                                             z_velocs=z_v, **kw)
            y_new = y_new.squeeze(0).contiguous()
            e_new = openmm_potential_energy_torch(y_new)                                  # :242
            if signs is not None:
                changes = check_symmetry_change(y_new, centers.to(device), signs.to(device))
                e_new = e_new + CHIRALITY_PENALTY * changes.to(e_new.dtype).unsqueeze(-1)  # :245 (reject if chirality changes)
            stay = e_new - e > energy_threshold                                           # [P,1]
            y_c = torch.where(stay.unsqueeze(-1), y_c, y_new)                             # :246-248
            e = torch.where(stay, e, e_new)                                               # :249
            trajectory.append(y_c)
            energy_log.append(e)
            y_v = noise.randn_like(y_c)                                                   # :253
        return (y_c, y_v, e), trajectory, energy_log

    with torch.no_grad():
        state = (y_coords, y_velocs, energies)
        trajectory, energy_log = [], []
        guarded = _range_guarded(model)
        # split-fp16 kernels: no range-flag read-back (a device synchronisation) per model call.  The run goes in WINDOWS of
        # RANGE_CHECK_WINDOW steps (r04 recorded the draws of the whole exploration - ~3x the trajectory's own memory
        # for num_steps of 1e5 - and looked at the flag once, at the end): only the open window's draws are kept; one look at the
        # flag per window; on an overflow the model is demoted and THAT window runs again from its starting state on the
        # exact-f32 kernels with the same draws, the rest of the exploration continues there with the ordinary noise source.
        done = 0
        while done < num_steps:
            n = min(RANGE_CHECK_WINDOW, num_steps - done)
            if guarded:
                rec = RecordingNoise(noise)
                with _deferred(model):
                    new_state, tr, el = run_window(rec, state, n)
                if model.demoted or model.split_fp16_overflowed(device):
                    model.demote_to_f32()
                    new_state, tr, el = run_window(ReplayDraws(rec.log), state, n)
                    guarded = False
            else:
                new_state, tr, el = run_window(noise, state, n)
            state = new_state
            trajectory += tr
            energy_log += el
            done += n
        return torch.cat(trajectory, dim=0), torch.cat(energy_log, dim=0)
