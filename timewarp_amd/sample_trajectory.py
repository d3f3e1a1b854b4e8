"""Long-chain driver with on-disk segments and resume: the sampling loop of the reference's
sample_trajectory.py (:213-279) without its argument parsing / checkpoint loading.

A chain of `num_samples` states is produced as `num_samples // saving_interval` segments; segment i is
written to `<output_dir>/<protein>_trajectory_model_<i>.npz` with the reference's two arrays

    positions  float32 [ceil((saving_interval+1)/10), V, 3]   every 10th state of the segment (first row =
                                                               the state the segment started from)
    time       float                                           wall-clock seconds of the segment

and the next segment starts from the last state of the previous one.  When `output_dir` already holds
segments the chain resumes after them, starting - exactly as the reference does (:234-240) - from the
last row of the last file's *thinned* `positions` array, which is the segment's final state only when
saving_interval is a multiple of 10."""
from __future__ import annotations

import os
from timeit import default_timer as timer
from typing import Callable, Optional

import numpy as np
import torch

from .utils.chirality import compute_chirality_sign, find_chirality_centers
from .utils.evaluation_utils import sample_with_model

THIN = 10  # sample_trajectory.py:271 `positions=sampled_coords[::10]`


def segment_path(output_dir: str, protein: str, i: int) -> str:
    return os.path.join(output_dir, f"{protein}_trajectory_model_{i}.npz")


def resume_point(output_dir: str, protein: str):
    """(number of segments on disk, last saved coordinates [1,V,3] or None)."""
    try:
        n = len(os.listdir(output_dir))
        npz = np.load(segment_path(output_dir, protein, n - 1))
        return n, torch.from_numpy(npz["positions"][-1:])
    except FileNotFoundError:
        return 0, None


def sample_trajectory(batch, model, device, energy_fn, masses, output_dir: str, protein: str, num_samples: int,
                      saving_interval: int, mh: bool = True, random_velocities: bool = False,
                      resample_velocities: bool = False, initialize_randomly: bool = False, num_proposal_steps: int = 1,
                      adaptive_parallelism: bool = False, conserve_chirality: bool = False,
                      sampler: Optional[Callable] = None, verbose: bool = True, sim=None, openmm_on_current: bool = False,
                      openmm_on_proposal: bool = False, num_openmm_steps: int = 0) -> int:
    """Run (or resume) the chain; returns the number of segments written by this call.
    `sim`, `openmm_on_current`, `openmm_on_proposal`, `num_openmm_steps`: the hybrid-move options the reference's script hands
    to `sample_with_model` (sample_trajectory.py:218, 258-261) - the caller's `Simulation`-like object (or a
    `timewarp_amd.md.LangevinDynamics`, or "device") is passed on only when one of the two switches is set, as there
    (`sim=simulation if needs_sim else None`)."""
    num_iters = num_samples // saving_interval
    assert num_iters > 0, "num_samples must be larger than saving_interval."
    sampler = sampler or sample_with_model
    os.makedirs(output_dir, exist_ok=True)
    chirality_centers = reference_signs = None
    if conserve_chirality:
        chirality_centers = find_chirality_centers(batch.adj_list, batch.atom_types)
        reference_signs = compute_chirality_sign(batch.atom_coords, chirality_centers)
    needs_sim = openmm_on_proposal or openmm_on_current   # sample_trajectory.py:218
    done, last = resume_point(output_dir, protein)
    if last is not None:
        batch.atom_coords = last
        if verbose:
            print("Resuming sampling")
    written = 0
    for i in range(done, num_iters):
        if verbose:
            print(f"Iteration {i+1}/{num_iters}")
        start = timer()
        sampled_coords, _, _, _ = sampler(
            batch, model, device, energy_fn, masses, saving_interval, mh, random_velocs=random_velocities,
            resample_velocs=resample_velocities, initialize_randomly=initialize_randomly,
            sim=sim if needs_sim else None, openmm_on_current=openmm_on_current, openmm_on_proposal=openmm_on_proposal,
            num_openmm_steps=num_openmm_steps, num_proposal_steps=num_proposal_steps, adaptive_parallelism=adaptive_parallelism,
            reference_signs=reference_signs, chirality_centers=chirality_centers, disable_tqdm=True)
        duration = timer() - start
        path = segment_path(output_dir, protein, i)
        if verbose:
            print(f"Saving trajectory to {path}")
        np.savez(path, positions=sampled_coords[::THIN], time=duration)
        batch.atom_coords = torch.from_numpy(sampled_coords[-1:])
        written += 1
    return written
