"""The slice of the reference's batch types the sampling path touches (dataloader.py:24-25,109-197)."""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import List, Optional, Sequence

import torch

KNOWN_ELEMENTS = ["C", "H", "N", "O", "S"]
ELEMENT_VOCAB = {e: i for i, e in enumerate(KNOWN_ELEMENTS)}


@dataclass
class DenseMolDynBatch:
    """Dense (padded) batch; every tensor is [B, ...] (dataloader.py:109-124)."""

    names: List[str]
    atom_types: torch.Tensor       # int64 [B, V]
    adj_list: torch.Tensor         # int64 [E, 2]
    edge_batch_idx: torch.Tensor   # int64 [E]
    atom_coords: torch.Tensor      # f32 [B, V, 3]
    atom_velocs: torch.Tensor
    atom_forces: torch.Tensor
    atom_coord_targets: torch.Tensor
    atom_veloc_targets: torch.Tensor
    atom_force_targets: torch.Tensor
    masked_elements: torch.Tensor  # bool [B, V]

    def todevice(self, device: torch.device) -> "DenseMolDynBatch":
        kw = {f.name: getattr(self, f.name) for f in fields(self)}
        return DenseMolDynBatch(**{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in kw.items()})


def single_state_batch(name: str, atom_types: torch.Tensor, coords: torch.Tensor, velocs: Optional[torch.Tensor] = None,
                       adj_list: Optional[torch.Tensor] = None) -> DenseMolDynBatch:
    """B = 1 batch for one conditioning state (what `moldyn_dense_collate_fn([dp])` yields,
    dataloader.py:328-400): no padding, targets = features."""
    coords = coords.reshape(1, -1, 3).to(torch.float32)
    V = coords.shape[1]
    velocs = torch.zeros_like(coords) if velocs is None else velocs.reshape(1, V, 3).to(torch.float32)
    adj = torch.zeros((0, 2), dtype=torch.int64) if adj_list is None else adj_list
    return DenseMolDynBatch(
        names=[name], atom_types=atom_types.reshape(1, V).to(torch.int64), adj_list=adj,
        edge_batch_idx=torch.zeros((adj.shape[0],), dtype=torch.int64), atom_coords=coords, atom_velocs=velocs,
        atom_forces=torch.zeros_like(coords), atom_coord_targets=coords.clone(), atom_veloc_targets=velocs.clone(),
        atom_force_targets=torch.zeros_like(coords), masked_elements=torch.zeros((1, V), dtype=torch.bool))


def elements_from_atom_names(names: Sequence[str]) -> torch.Tensor:
    """Element id from a PDB atom name = its first alphabetic character (1HH3 -> H, CA -> C)."""
    return torch.tensor([ELEMENT_VOCAB[next(ch for ch in n if ch.isalpha())] for n in names], dtype=torch.int64)


def random_rotation_matrix(dtype=torch.float32) -> torch.Tensor:
    """Uniform SO(3) rotation (QR of a Gaussian matrix, sign-fixed; the reference draws its matrix with
    scipy, equivariance/equivariance_utils.py)."""
    q, r = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.to(dtype)


def transform_batch(batch: DenseMolDynBatch, rotation: Optional[torch.Tensor] = None,
                    translation: Optional[torch.Tensor] = None, dtype=torch.float32) -> DenseMolDynBatch:
    """Data augmentation used by `sample_on_batches`: translate, then rotate (coordinates get both,
    velocities and forces only the rotation), equivariance/equivariance_transforms.py:153-177.  With no
    arguments a random translation ~ N(0, 1)^3 and a random rotation are drawn."""
    t = torch.randn(3, dtype=dtype) if translation is None else translation.to(dtype)
    R = random_rotation_matrix(dtype) if rotation is None else rotation.to(dtype)

    def coord(x):
        return ((R.to(x.device) @ (x + t.to(x.device)).transpose(-1, -2)).transpose(-1, -2)).contiguous()

    def veloc(v):
        return ((R.to(v.device) @ v.transpose(-1, -2)).transpose(-1, -2)).contiguous()

    return DenseMolDynBatch(
        names=batch.names, atom_types=batch.atom_types, adj_list=batch.adj_list, edge_batch_idx=batch.edge_batch_idx,
        atom_coords=coord(batch.atom_coords), atom_velocs=veloc(batch.atom_velocs), atom_forces=veloc(batch.atom_forces),
        atom_coord_targets=coord(batch.atom_coord_targets), atom_veloc_targets=veloc(batch.atom_veloc_targets),
        atom_force_targets=veloc(batch.atom_force_targets), masked_elements=batch.masked_elements)
