"""Chirality bookkeeping used to set up the MH chirality guard (reference utils/chirality.py:14-80).
Host-side, run once per chain; the per-proposal check runs on the GPU (tw_chirality_changed)."""
from __future__ import annotations

import torch


def find_chirality_centers(adj_list: torch.Tensor, atom_types: torch.Tensor, num_h_atoms: int = 2) -> torch.Tensor:
    """Rows (centre, n1, n2, n3) for every atom with exactly four bonds of which more than
    `num_h_atoms` go to non-hydrogen atoms (utils/chirality.py:14-38).  `adj_list` [E,2] holds each
    bond once; `atom_types` [1,V] uses the element vocabulary C0 H1 N2 O3 S4.

    As in the reference, a candidate is identified by its position in the sorted list of atoms that
    occur in `adj_list`, which equals the atom index whenever every atom has at least one bond."""
    occurring, counts = torch.unique(adj_list, return_counts=True)
    rows = []
    for centre in torch.where(counts == 4)[0]:
        bond, pos = torch.where(adj_list == centre)
        neighbours = adj_list[bond, (pos + 1) % 2]
        n_heavy = int(torch.count_nonzero(atom_types[0][neighbours] - 1))
        if n_heavy > num_h_atoms:
            rows.append([int(centre), *[int(n) for n in neighbours[:3]]])
    return torch.tensor(rows).to(adj_list)


def compute_chirality_sign(coords: torch.Tensor, chirality_centers: torch.Tensor) -> torch.Tensor:
    """Sign of the triple product (n1-c) . ((n2-c) x (n3-c)) per centre, [..., n_centres]
    (utils/chirality.py:41-60)."""
    assert coords.dim() == 3
    d = coords[:, chirality_centers[:, 1:], :] - coords[:, chirality_centers[:, [0]], :]
    return torch.sign(torch.einsum("bci,bci->bc", d[:, :, 0], torch.cross(d[:, :, 1], d[:, :, 2], dim=-1)))
