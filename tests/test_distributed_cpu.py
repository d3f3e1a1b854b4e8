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


def _ragged_worker(rank, world, port, q, lengths):
    """Unequal trajectory lengths including ranks that hold nothing, then a round in which NO rank holds anything."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from timewarp_amd import distributed

    distributed.init_from_env("gloo")

    def traj(r):
        g = torch.Generator().manual_seed(distributed.chain_seed(7, r))
        return torch.randn(lengths[r], 22, 3, generator=g), torch.rand(lengths[r], generator=g)

    coords, acc = traj(rank)
    all_c, all_s = distributed.gather_trajectories(coords, {"acceptance": acc})
    ok = len(all_c) == world
    for r in range(world):
        rc, ra = traj(r)
        ok = ok and all_c[r].shape == (lengths[r], 22, 3) and torch.equal(all_c[r], rc) and torch.equal(all_s["acceptance"][r], ra)
    none_c, none_s = distributed.gather_trajectories(torch.zeros(0, 22, 3), {"acceptance": torch.zeros(0)})
    ok = ok and len(none_c) == world and all(c.shape == (0, 22, 3) for c in none_c)
    ok = ok and all(a.shape == (0,) for a in none_s["acceptance"])
    total = distributed.all_reduce_counters([float(lengths[rank])], "cpu")
    ok = ok and total == [float(sum(lengths))]
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_trajectories_four_ranks_ragged_and_empty_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    lengths = [6, 0, 11, 1]
    procs = [ctx.Process(target=_ragged_worker, args=(r, 4, port, q, lengths)) for r in range(4)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == {0: True, 1: True, 2: True, 3: True}


def _bench_worker(rank, world, port, q):
    """bench.py's own N > 1 leg (end_timed_region + whole_job_rates) under gloo on CPU tensors: rank r 'ran' for
    0.1 (r + 1) s and accepted 10 (r + 1) samples."""
    import sys
    import time

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from timewarp_amd import distributed

    distributed.init_from_env("gloo")
    traj = torch.full((3 + rank, 22, 3), float(rank))
    t0 = time.perf_counter() - 0.1 * (rank + 1)  # pretend this rank started 0.1 (r + 1) s ago
    gathered, elapsed = bench.end_timed_region(traj, t0, "cpu", world)
    value, prop_s, states_s, accepted = bench.whole_job_rates(10.0 * (rank + 1), 1000.0, 50.0, elapsed, "cpu")
    ok = len(gathered) == world and all(g.shape == (3 + r, 22, 3) and bool((g == r).all()) for r, g in enumerate(gathered))
    ok = ok and 0.1 * world <= elapsed < 0.1 * world + 5.0          # max over ranks, same on every rank
    per_rank = bench.end_timed_region.per_rank_seconds               # what the JSON line reports as per_rank_ms
    ok = ok and len(per_rank) == world and abs(max(per_rank) - elapsed) < 1e-12
    ok = ok and all(per_rank[r] >= 0.1 * (r + 1) for r in range(world)) and per_rank[0] < per_rank[-1] + 1.0
    ok = ok and accepted == sum(10.0 * (r + 1) for r in range(world))  # whole-job sum
    ok = ok and abs(value - accepted / elapsed) < 1e-9 and abs(prop_s - 1000.0 * world / elapsed) < 1e-6
    q.put((rank, bool(ok), elapsed))
    dist.destroy_process_group()


def test_bench_multi_rank_leg_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in results) == [0, 1] and all(r[1] for r in results)
    assert abs(results[0][2] - results[1][2]) < 1e-12  # both ranks hold the same max-over-ranks time


def test_single_process_is_a_no_op():
    from timewarp_amd import distributed

    c = torch.randn(4, 22, 3)
    all_c, all_s = distributed.gather_trajectories(c, {"a": torch.arange(4.0)})
    assert len(all_c) == 1 and torch.equal(all_c[0], c) and torch.equal(all_s["a"][0], torch.arange(4.0))
    assert distributed.chain_seed(7, 3, 1, 4) == 7 + 13


def _eight_rank_worker(rank, world, port, q, lengths, tmp):
    """What an 8-GPU node does at start-up and at collection time, on CPU: every rank finds the library missing at the
    same moment (one must build, seven must wait and then load the finished file), then bench.py's N > 1 leg with ragged
    trajectory lengths, including ranks that hold nothing."""
    import ctypes
    import shutil
    import sys
    import time

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from timewarp_amd import build as B
    from timewarp_amd import distributed

    # a private copy of the build state: library and stamp absent -> stale for everybody
    real_lib = B.LIB_PATH
    B.LIB_DIR = tmp
    B.LIB_PATH = os.path.join(tmp, "libtimewarp_hip.so")
    B.STAMP_PATH = os.path.join(tmp, ".build_stamp")
    log = os.path.join(tmp, "builders.log")

    def fake_build(hipcc, verbose):  # stands in for the 45 s of hipcc: slow, and NOT atomic unless the lock works
        with open(log, "a") as f:
            f.write(f"{rank}\n")
        time.sleep(0.8)
        part = B.LIB_PATH + ".part"
        shutil.copyfile(real_lib, part)
        os.replace(part, B.LIB_PATH)
        with open(B.STAMP_PATH, "w") as f:
            f.write(B._source_digest())
        return B.LIB_PATH

    B._build_locked = fake_build
    B.hipcc_path = lambda: "/bin/true"
    distributed.init_from_env("gloo")
    dist.barrier()                      # all ranks reach the build together
    path = B.build_library()
    lib = ctypes.CDLL(path)             # complete file, whoever built it
    ok = lib.tw_abi_version() > 0 and not B.stale()
    dist.barrier()
    with open(log) as f:
        ok = ok and len(f.read().split()) == 1   # exactly one builder

    import bench

    traj = torch.full((lengths[rank], 22, 3), float(rank))
    t0 = time.perf_counter() - 0.05 * (rank + 1)
    gathered, elapsed = bench.end_timed_region(traj, t0, "cpu", world)
    ok = ok and len(gathered) == world and all(g.shape == (lengths[r], 22, 3) and bool((g == r).all()) for r, g in enumerate(gathered))
    value, _, _, accepted = bench.whole_job_rates(float(rank), 1000.0, float(lengths[rank]), elapsed, "cpu")
    ok = ok and accepted == sum(range(world)) and elapsed >= 0.05 * world
    ok = ok and len(bench.end_timed_region.per_rank_seconds) == world
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_eight_ranks_library_race_and_ragged_collection_gloo(tmp_path):
    """r03 review: nothing had ever run the N = 8 shape.  Eight gloo ranks: the `.so` build race (flock in
    timewarp_amd/build.py) and bench.end_timed_region with trajectory lengths 9 / 0 / 4 / 17 / 1 / 0 / 30 / 2."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    lengths = [9, 0, 4, 17, 1, 0, 30, 2]
    procs = [ctx.Process(target=_eight_rank_worker, args=(r, 8, port, q, lengths, str(tmp_path))) for r in range(8)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert results == {r: True for r in range(8)}


def _clean_env():
    """What a driver's bare `python3 bench.py ...` sees: no torchrun variables at all."""
    keep = ("PATH", "HOME", "LD_LIBRARY_PATH", "PYTHONPATH", "TMPDIR", "HSA_ENABLE_IPC_MODE_LEGACY", "ROCM_PATH", "HIP_VISIBLE_DEVICES")
    return {k: v for k, v in os.environ.items() if k in keep}


def _json_lines(text):
    import json

    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            out.append(json.loads(line))
    return out


def test_bare_bench_command_starts_its_own_ranks():
    """`python bench.py --gpus N` without torchrun used to run ONE rank and print "n_gpus": 1.  The bare
    form must start N ranks itself (re-exec under torch.distributed.run), and a torchrun environment of another size must be
    refused.  --plumbing-only: the launch + collection leg without a GPU (the GPU suite runs the real line the same way)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1", "--plumbing-only"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout   # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and len(line["per_rank_ms"]) == 2 and line["gathered_ok"] and line["states_gathered"] == 7.0
    assert line["plumbing_only"] is True and line["value"] is None   # can never be mistaken for a measurement
    # a torchrun environment that disagrees with --gpus: refused, nothing printed
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--plumbing-only"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not _json_lines(r.stdout) and "WORLD_SIZE=1" in r.stderr
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--plumbing-only"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not _json_lines(r.stdout)
