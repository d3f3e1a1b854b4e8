"""RCCL self-test on one GPU (r03 review: the "nccl" backend had never executed anywhere).  A single-rank process group
with backend "nccl" (= RCCL on ROCm) on cuda:0, and the path's two collectives forced through it: the all-gather of
variable-length trajectories and the all-reduce of the counters (timewarp_amd/distributed.py, `force_collective=True`
bypasses the world-of-one early return), then bench.py's timed-region epilogue.  Runs in a subprocess so the process
group does not outlive the test.  What stays unmeasured without an 8-GPU node: xGMI transfers, per-rank device
selection with 8 visible devices."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_SCRIPT = r'''
import os, sys, time
sys.path.insert(0, {root!r})
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from timewarp_amd import distributed
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
coords = torch.randn(37, 22, 3, generator=g).to(dev)
stats = {{"acceptance": torch.rand(37, generator=g).to(dev), "p_xy": torch.randn(37, generator=g).to(dev)}}
all_c, all_s = distributed.gather_trajectories(coords, stats, force_collective=True)
assert len(all_c) == 1 and torch.equal(all_c[0], coords) and all_c[0].data_ptr() != coords.data_ptr()   # went through the collective
assert torch.equal(all_s["p_xy"][0], stats["p_xy"]) and torch.equal(all_s["acceptance"][0], stats["acceptance"])
empty_c, _ = distributed.gather_trajectories(torch.zeros(0, 22, 3, device=dev), {{}}, force_collective=True)
assert empty_c[0].shape == (0, 22, 3)
assert distributed.all_reduce_counters([3.0, 1000.0, 41.0], dev, force_collective=True) == [3.0, 1000.0, 41.0]
t = torch.ones(1 << 20, device=dev)
dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
assert float(t.sum()) == float(1 << 20)
import bench
gathered, elapsed = bench.end_timed_region(coords, time.perf_counter() - 0.01, dev, 1)
assert len(gathered) == 1 and elapsed >= 0.01
dist.destroy_process_group()
import ctypes
maps = open("/proc/self/maps").read()
print("RCCL_OK", "librccl" in maps)
'''


def test_single_rank_nccl_group_runs_the_paths_collectives():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 36500 + os.getpid() % 2000
    r = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=root, port=port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_OK True" in r.stdout, r.stdout


def _clean_env(**extra):
    keep = ("PATH", "HOME", "LD_LIBRARY_PATH", "PYTHONPATH", "TMPDIR", "HSA_ENABLE_IPC_MODE_LEGACY", "ROCM_PATH", "HIP_VISIBLE_DEVICES")
    env = {k: v for k, v in os.environ.items() if k in keep}
    env.update(extra)
    return env


def _json_lines(text):
    import json

    return [json.loads(l) for l in (x.strip() for x in text.splitlines()) if l.startswith("{") and l.endswith("}")]


def test_bare_bench_command_runs_n_ranks_on_the_gpu():
    """The driver's form - `python bench.py --gpus N ...`, no torchrun variables - must measure N ranks.
    On a 1-GPU box: (a) N = 2 with TW_DIST_BACKEND=gloo (both ranks share the GPU: a plumbing line, two per-rank times, the
    all-gather inside the timed region); (b) N = 2 under RCCL is REFUSED (one GPU per rank or nothing) with a non-zero exit
    and no line; (c) N = 1 prints n_gpus 1."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--chains", "2"]   # --chains: a line without the headline's companions
    r = subprocess.run([sys.executable, bench, "--gpus", "2"] + common, env=_clean_env(TW_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and len(lines[0]["per_rank_ms"]) == 2 and lines[0]["value"] > 0, r.stdout
    assert lines[0]["proposals_per_s"] > 0 and lines[0]["roofline"]["frac"] > 0
    import torch

    if torch.cuda.device_count() == 1:
        r = subprocess.run([sys.executable, bench, "--gpus", "2"] + common, env=_clean_env(), capture_output=True, text=True, timeout=900)
        assert r.returncode != 0 and not _json_lines(r.stdout), r.stdout[-2000:]
        assert "one GPU per rank" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "1"] + common, env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and len(lines[0]["per_rank_ms"]) == 1
