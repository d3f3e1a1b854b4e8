"""64-token waves (one molecule of 49-64 atoms per wave) against the wide layout (floor(192 / V) molecules per workgroup), forced
by tw_debug_set_flags 65536 / 131072, and what the launch code picks (flag 0): one reverse pass, ms and algorithmic TFLOP/s.
Evidence for h3_nt4_choice's cost model (csrc/tw_netblock_h3.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as H
from timewarp_amd import _lib

lib = _lib.load()
PATH = 4 if "--h1" in sys.argv else 3   # --h1: the fast mode (TW_PATH_FUSED_H1) on both layouts
sd = H.full_kernel_sd()
g = torch.Generator().manual_seed(0)
for V, S in ((60, 256), (60, 512), (60, 768), (60, 1024), (64, 512), (49, 512), (52, 1000)):
    at = torch.randint(0, 5, (1, V), generator=g).cuda()
    xc = (torch.randn(1, V, 3, generator=g) * 0.5).cuda(); xv = torch.randn(1, V, 3, generator=g).cuda()
    mk = torch.zeros(1, V, dtype=torch.bool).cuda()
    zc = torch.randn(S, 1, V, 3, generator=g).cuda(); zv = torch.randn(S, 1, V, 3, generator=g).cuda()
    res = {}
    for name, flags in (("wide", 131072), ("64-token", 65536), ("chosen", 0)):
        lib.tw_debug_set_flags(flags)
        m = H.tw_kernel_model(sd, path=PATH); m._defer_range_check += 1
        f = lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                                   masked_elements=mk, num_samples=S, z_coords=zc, z_velocs=zv)
        for _ in range(2): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 5 * 1e3
    lib.tw_debug_set_flags(0)
    flop = 16 * V * (4478976 + 4608 * V) * S
    print(f"V={V:3d} S={S:5d}: " + "  ".join(f"{k} {v:7.3f} ms ({flop / v / 1e9:6.1f} TFLOP/s)" for k, v in res.items()), flush=True)
