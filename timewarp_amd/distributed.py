"""Chain-level data parallelism for sampling: every rank owns independent
Markov chains and a full weight replica; nothing is exchanged while sampling.  The only collective
is one all-gather of the per-rank trajectories (plus their lengths) at collection time -- RCCL over
xGMI with backend "nccl" on the GPUs, gloo in the CPU tests."""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the default
    process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def chain_seed(base_seed: int, rank: int, chain: int = 0, chains_per_rank: int = 1) -> int:
    """Distinct, reproducible seed per global chain id (rank r owns chains r*C .. r*C+C-1)."""
    return base_seed + rank * chains_per_rank + chain


def gather_trajectories(coords: torch.Tensor, extras: Dict[str, torch.Tensor] = None, force_collective: bool = False):
    """All-gather variable-length per-rank trajectories.

    coords [n_r, V, 3] on this rank's device -> list over ranks of [n_r, V, 3] (same on every rank).
    `extras`: per-state vectors [n_r] (ChainStats fields) gathered alongside.  One all_gather of the
    lengths (8 bytes/rank) and one of the zero-padded payload.  A world of one returns at once unless
    `force_collective` (the RCCL self-test: the real collectives on a single-rank group)."""
    extras = extras or {}
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        return [coords], {k: [v] for k, v in extras.items()}
    world = dist.get_world_size()
    dev = coords.device
    n = torch.tensor([coords.shape[0]], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    n_max = max(lens)
    V = coords.shape[1]
    width = V * 3 + len(extras)
    if n_max == 0:  # nobody has anything: no payload collective (a zero-byte all-gather is backend-dependent)
        empty = coords.new_zeros((0, V, 3), dtype=torch.float32)
        return [empty.clone() for _ in range(world)], {k: [coords.new_zeros((0,), dtype=torch.float32) for _ in range(world)]
                                                       for k in extras}
    payload = torch.zeros((n_max, width), dtype=torch.float32, device=dev)
    payload[: coords.shape[0], : V * 3] = coords.reshape(coords.shape[0], V * 3)
    for j, (k, v) in enumerate(extras.items()):
        payload[: v.shape[0], V * 3 + j] = v.to(torch.float32)
    out = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(out, payload)
    coords_all = [o[:l, : V * 3].reshape(l, V, 3) for o, l in zip(out, lens)]
    extras_all = {k: [o[:l, V * 3 + j] for o, l in zip(out, lens)] for j, k in enumerate(extras)}
    return coords_all, extras_all


def all_reduce_counters(values: List[float], device, force_collective: bool = False) -> List[float]:
    """Sum small counters (proposals, accepted, states) over ranks."""
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
