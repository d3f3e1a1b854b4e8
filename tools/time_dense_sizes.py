"""Reverse-pass time of the dense softmax flow (transformer_nvp) at given (atoms, proposals) pairs on the exact-f32 fused kernel
(path 1) and the split-fp16 kernel (path 3): `python tools/time_dense_sizes.py 60x512 64x512 22x1000`."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from timewarp_amd import _lib

lib = _lib.load()
g = torch.Generator().manual_seed(0)
for spec in sys.argv[1:] or ["60x512"]:
    V, S = (int(t) for t in spec.split("x"))
    at = torch.randint(0, 5, (1, V), generator=g).cuda()
    xc = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
    xv = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
    mk = torch.zeros(1, V, dtype=torch.bool).cuda()
    flop = 16 * V * 3726336 * S
    for path, name in ((1, "exact-f32 fused dense kernel"), (3, "split-fp16 dense kernel")):
        m = H.tw_dense_model(H.full_dense_sd(), path=path)
        f = lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                                   masked_elements=mk, num_samples=S)
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        print(f"dense V={V} S={S} {name}: {dt:.2f} ms per reverse pass, {flop / dt / 1e9:.1f} TFLOP/s algorithmic "
              f"[{lib.tw_last_netblock_kernel().decode()}]", flush=True)
