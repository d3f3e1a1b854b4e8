"""Quick timing of the flow entry points at the BASELINE size (S=1000 proposals, 22 atoms)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as fo  # noqa: E402  (synthetic weights recipe only)
from tests import helpers as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=1000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--paths", default="1,3")
ap.add_argument("--debug-flags", type=int, default=0)
args = ap.parse_args()

from timewarp_amd import _lib
_lib.load().tw_debug_set_flags(args.debug_flags)
d, _ = H.load("kernel_full_ad_calibrated")
sd = H.full_kernel_sd(calibrated=True)
S = args.S
g = torch.Generator().manual_seed(5)
zc, zv = fo.draw_latents(sd, S, (1, 22, 3), g)
at, xc, xv, mk = d["atom_types"].cuda(), d["x_coords"].cuda(), d["x_velocs"].cuda(), d["masked"].cuda()
zc, zv = zc.cuda(), zv.cuda()
FLOP_PASS = 1.612e9 * S
for path in [int(p) for p in args.paths.split(",")]:
    m = H.tw_kernel_model(sd, path=path)
    def sample():
        return m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                              masked_elements=mk, num_samples=S, z_coords=zc, z_velocs=zv)
    with m.deferred_range_check():
        yc, yv, lp = sample()
    def loglik():
        return m.log_likelihood(atom_types=at.repeat(S, 1), x_coords=yc.squeeze(1), x_velocs=-yv.squeeze(1),
                                y_coords=xc.repeat(S, 1, 1), y_velocs=-xv.repeat(S, 1, 1), adj_list=None,
                                edge_batch_idx=None, masked_elements=mk.repeat(S, 1))
    m._defer_range_check += 1  # timing tool: no range-flag read-back (and no demotion when an experiment breaks the numbers)
    for name, fn in (("sample(reverse)", sample), ("loglik(forward)", loglik)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        print(f"path={path} {name}: {ms:.3f} ms  -> {FLOP_PASS / ms / 1e9:.1f} TFLOP/s algorithmic, {S / ms * 1e3:.0f} rows/s", flush=True)
