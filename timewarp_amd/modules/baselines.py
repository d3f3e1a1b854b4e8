"""EulerMaruyamaGaussian: the `gaussian_baseline.yaml` plumbing model (modules/baselines.py:169-296).

Closed-form Gaussians around one Euler-Maruyama step; five tiny learnable vectors, no kernels worth
writing -- it exists so that `sample.py`-style drivers (utils/sampling_utils.py) can be exercised
end to end.  Elementwise torch ops on whatever device the inputs live on."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor


class EulerMaruyamaGaussian(nn.Module):
    # marker read by utils/sampling_utils.get_sample to pass `x_forces` (ConditionalDensityModelWithForce)
    takes_forces = True

    def __init__(self, step_width_init: float = 1):
        super().__init__()
        self.k_B = 1.380649e-23 * 1e-3 * 6.02214076e23  # kJ/(mol K)
        self.mass_vocab = [12.011, 1.00797, 14.0067, 15.9994, 32.06]  # C H N O S
        self.temperature = 310
        self.delta_t = step_width_init * 0.5 * 1e-3  # fs -> ps
        self.gamma = 0.3
        n_types = len(self.mass_vocab)
        self.delta_t_factor_param = nn.Parameter(torch.tensor([0.0]))
        self.atom_mass_params = nn.Parameter(torch.log(torch.tensor(self.mass_vocab)))
        self.atom_coord_std_params = nn.Parameter(-torch.ones(n_types))
        self.atom_veloc_std_params = nn.Parameter(-torch.ones(n_types))

    def _get_y_dist(self, atom_types: Tensor, x_coords: Tensor, x_velocs: Tensor, x_forces: Tensor, logger=None):
        coord_stds = torch.exp(self.atom_coord_std_params[atom_types])
        masses = torch.exp(self.atom_mass_params[atom_types])
        f = torch.exp(self.delta_t_factor_param)
        coord_mean = x_coords + self.delta_t * f * x_velocs
        force_term = (x_forces / masses[:, :, None]) * self.delta_t * f
        friction_term = -self.gamma * x_velocs * self.delta_t * f
        veloc_mean = x_velocs + force_term + friction_term
        veloc_stds = torch.sqrt(2.0 * self.gamma * self.k_B * self.temperature * self.delta_t * f / masses)
        veloc_stds = veloc_stds + torch.exp(self.atom_veloc_std_params[atom_types])
        p_c = torch.distributions.Normal(loc=coord_mean, scale=coord_stds[:, :, None].repeat(1, 1, 3))
        p_v = torch.distributions.Normal(loc=veloc_mean, scale=veloc_stds[:, :, None].repeat(1, 1, 3))
        if logger is not None:
            logger.log_scalar_async("coord_std", coord_stds.mean())
            logger.log_scalar_async("veloc_std", veloc_stds.mean())
        return p_c, p_v

    def log_likelihood(self, atom_types, x_coords, x_velocs, x_forces, y_coords, y_velocs, adj_list, edge_batch_idx,
                       masked_elements, logger=None) -> Tensor:
        p_c, p_v = self._get_y_dist(atom_types, x_coords, x_velocs, x_forces, logger)
        keep = ~masked_elements[:, :, None]
        lc = (keep * p_c.log_prob(y_coords)).sum(dim=(-1, -2))
        lv = (keep * p_v.log_prob(y_velocs)).sum(dim=(-1, -2))
        return lc + lv

    def forward(self, atom_types, x_coords, x_velocs, x_forces, y_coords, y_velocs, adj_list, edge_batch_idx,
                masked_elements, logger=None) -> Tensor:
        num_atoms = (~masked_elements).sum(dim=1)
        ll = self.log_likelihood(atom_types, x_coords, x_velocs, x_forces, y_coords, y_velocs, adj_list,
                                 edge_batch_idx, masked_elements, logger)
        return -(ll / num_atoms).mean()

    def conditional_sample(self, atom_types, x_coords, x_velocs, x_forces, adj_list, edge_batch_idx,
                           masked_elements, num_samples: int, logger=None) -> Tuple[Tensor, Tensor]:
        p_c, p_v = self._get_y_dist(atom_types, x_coords, x_velocs, x_forces, logger)
        return p_c.sample((num_samples,)), p_v.sample((num_samples,))
