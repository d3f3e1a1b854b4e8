"""The density-model protocol the reference's scripts program against
(modules/model_wrappers/base.py:9-30, density_model_base.py:10-88).

When the reference package is already imported in this process (its sample.py / evaluate.py
driving our model) the HIP model derives from the reference's own ABC so that the
`functools.singledispatch` in utils/sampling_utils.py:17-46 and utils/loss_utils.py:91-119 picks it
up; otherwise an ABC with the same contract is defined here."""
from __future__ import annotations

import abc
import sys
from typing import Optional, Tuple

import torch.nn as nn
from torch import Tensor

_REF_MOD = "timewarp.modules.model_wrappers.density_model_base"


class _ConditionalDensityModel(nn.Module, abc.ABC):
    def forward(self, atom_types: Tensor, x_coords: Tensor, x_velocs: Tensor, y_coords: Tensor, y_velocs: Tensor,
                adj_list: Optional[Tensor], edge_batch_idx: Optional[Tensor], masked_elements: Tensor,
                logger=None) -> Tensor:
        """Mean negative log-likelihood per atom (density_model_base.py:14-47)."""
        num_atoms = (~masked_elements).sum(dim=1)
        ll = self.log_likelihood(
            atom_types=atom_types, x_coords=x_coords, x_velocs=x_velocs, y_coords=y_coords, y_velocs=y_velocs,
            adj_list=adj_list, edge_batch_idx=edge_batch_idx, masked_elements=masked_elements, logger=logger,
        )
        loss = -(ll / num_atoms).mean()
        if logger is not None:
            logger.log_scalar_async("nll_loss", loss)
        return loss

    @abc.abstractmethod
    def conditional_sample(self, atom_types, x_coords, x_velocs, adj_list, edge_batch_idx, masked_elements,
                           num_samples: int, logger=None) -> Tuple[Tensor, Tensor]:
        ...

    @abc.abstractmethod
    def log_likelihood(self, atom_types, x_coords, x_velocs, y_coords, y_velocs, adj_list, edge_batch_idx,
                       masked_elements, logger=None) -> Tensor:
        ...


def resolve_base():
    ref = sys.modules.get(_REF_MOD)
    if ref is not None and hasattr(ref, "ConditionalDensityModel"):
        return ref.ConditionalDensityModel
    return _ConditionalDensityModel


ConditionalDensityModel = resolve_base()
