"""Reverse-pass time of the kernel flow at given (atoms, proposals) pairs on the per-op path, the split-fp16 kernel and the fast
mode: `python tools/time_sizes.py 88x512 95x256 ...` (r04: 81 .. 95 atoms joined the wide layout at a slot stride of 96; until
then they ran on the per-op path - profiles/r04_sizes_81_95.txt)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


g = torch.Generator().manual_seed(0)
for spec in sys.argv[1:] or ["88x512"]:
    V, S = (int(t) for t in spec.split("x"))
    at = torch.randint(0, 5, (1, V), generator=g).cuda()
    xc = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
    xv = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
    mk = torch.zeros(1, V, dtype=torch.bool).cuda()
    flop = 16 * V * (4478976 + 4608 * V) * S
    res = {}
    for path, name in ((2, "per-op path"), (3, "split-fp16 kernel"), (4, "fast mode (not a parity path)")):
        m = H.tw_kernel_model(H.full_kernel_sd(), path=path)
        res[path] = timed(lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None,
                                                                 edge_batch_idx=None, masked_elements=mk, num_samples=S),
                          iters=2 if path == 2 else 5)
        print(f"V={V} S={S} {name}: {res[path]:.2f} ms per reverse pass, {flop / res[path] / 1e9:.1f} TFLOP/s algorithmic")
    print(f"  V={V}: split-fp16 / per-op {res[2] / res[3]:.1f}x, fast / split-fp16 {res[3] / res[4]:.2f}x")
