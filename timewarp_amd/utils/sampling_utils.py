"""One-at-a-time conditional sampling driver used by `sample.py`
(reference utils/sampling_utils.py:17-181): same function names, arguments and return types."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor
from tqdm.auto import tqdm


def get_sample(model, batch, num_samples: int, device: Optional[torch.device] = None, tb_logger=None) -> Tuple[Tensor, Tensor]:
    """model.conditional_sample on the batch; models that take forces (the Euler-Maruyama
    baseline, `ConditionalDensityModelWithForce` in the reference) also receive `x_forces`."""
    kw = dict(
        atom_types=batch.atom_types.to(device, non_blocking=True),
        x_coords=batch.atom_coords.to(device, non_blocking=True),
        x_velocs=batch.atom_velocs.to(device, non_blocking=True),
        adj_list=batch.adj_list.to(device, non_blocking=True),
        edge_batch_idx=batch.edge_batch_idx.to(device, non_blocking=True),
        masked_elements=batch.masked_elements.to(device, non_blocking=True),
        num_samples=num_samples,
        logger=tb_logger,
    )
    if getattr(model, "takes_forces", False):
        kw["x_forces"] = batch.atom_forces.to(device, non_blocking=True)
    return model.conditional_sample(**kw)


def get_decorrelated_sample(model, batch, num_samples: int, device: Optional[torch.device] = None) -> Tuple[Tensor, Tensor]:
    """Every atom taken from an independent joint sample (destroys inter-atom correlations)."""
    assert num_samples == 1
    coords = torch.zeros_like(batch.atom_coords)
    velocs = torch.zeros_like(batch.atom_velocs)
    for atom in range(batch.atom_coords.shape[-2]):
        with torch.no_grad():
            c, v = get_sample(model, batch, num_samples=num_samples, device=device)
        coords[:, atom, :] = c[0, :, atom, :].to(coords.device)
        velocs[:, atom, :] = v[0, :, atom, :].to(velocs.device)
    return coords[None, ...], velocs[None, ...]


def sample(model, batch, num_samples: int, decorrelated: bool = False, device: Optional[torch.device] = None) -> Tuple[np.ndarray, np.ndarray]:
    assert len(batch.atom_coords) == 1, f"Expected batchsize of one instead of {len(batch.atom_coords)}."
    y_coords = np.zeros((num_samples, batch.atom_coords.shape[-2], batch.atom_coords.shape[-1]))
    y_velocs = np.zeros((num_samples, batch.atom_velocs.shape[-2], batch.atom_velocs.shape[-1]))
    draw = get_decorrelated_sample if decorrelated else get_sample
    for i in range(num_samples):
        c, v = draw(model, batch, num_samples=1, device=device)
        y_coords[i] = c[0, 0].detach().cpu().numpy()
        y_velocs[i] = v[0, 0].detach().cpu().numpy()
    return y_coords, y_velocs


def sample_from_trajectory(model, batches: List, num_samples: int, decorrelated: bool = False,
                           device: Optional[torch.device] = None) -> Tuple[List, List]:
    """Length-B lists of [S, V, 3] float64 arrays, one per conditioning state."""
    assert len(batches[0].atom_coords) == 1, f"Expected batchsize of one instead of {len(batches[0].atom_coords)}."
    out_c: List[np.ndarray] = []
    out_v: List[np.ndarray] = []
    for batch in tqdm(batches, desc="Sampling", unit="initial state"):
        with torch.no_grad():
            c, v = sample(model=model, batch=batch, num_samples=num_samples, decorrelated=decorrelated, device=device)
        out_c.append(c)
        out_v.append(v)
    return out_c, out_v
