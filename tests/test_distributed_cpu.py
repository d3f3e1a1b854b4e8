"""The N>1 path on CPU: world_size-2 gloo process group, independent chains per rank and the one
all-gather of variable-length trajectories."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from timewarp_amd import distributed

    r, w, l = distributed.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n = 5 + 3 * rank  # ranks hold trajectories of different length
    g = torch.Generator().manual_seed(distributed.chain_seed(100, rank))
    coords = torch.randn(n, 22, 3, generator=g)
    stats = {"acceptance": torch.rand(n, generator=g), "p_xy": torch.randn(n, generator=g)}
    all_c, all_s = distributed.gather_trajectories(coords, stats)
    counters = distributed.all_reduce_counters([float(n), 1.0], "cpu")
    ok = len(all_c) == world and torch.equal(all_c[rank], coords)
    for other in range(world):
        go = torch.Generator().manual_seed(distributed.chain_seed(100, other))
        no = 5 + 3 * other
        ref = torch.randn(no, 22, 3, generator=go)
        ok = ok and all_c[other].shape == (no, 22, 3) and torch.equal(all_c[other], ref)
        ok = ok and all_s["acceptance"][other].shape == (no,)
    ok = ok and torch.equal(all_s["p_xy"][rank], stats["p_xy"])
    ok = ok and counters == [sum(5.0 + 3 * k for k in range(world)), float(world)]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_trajectories_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True}


def test_single_process_is_a_no_op():
    from timewarp_amd import distributed

    c = torch.randn(4, 22, 3)
    all_c, all_s = distributed.gather_trajectories(c, {"a": torch.arange(4.0)})
    assert len(all_c) == 1 and torch.equal(all_c[0], c) and torch.equal(all_s["a"][0], torch.arange(4.0))
    assert distributed.chain_seed(7, 3, 1, 4) == 7 + 13
