"""Narrow (one 25-48-atom molecule per 48-token wave) against wide (floor(192 / V) molecules per workgroup) layout of the
split-fp16 kernel, forced by tw_debug_set_flags 16384 / 32768, and what the launch code picks by itself (flag 0):
one reverse pass, ms and algorithmic TFLOP/s.  Evidence for h3_wide_choice's cost model (csrc/tw_netblock_h3.hip)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from timewarp_amd import _lib


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


lib = _lib.load()
g = torch.Generator().manual_seed(0)
sd = H.full_kernel_sd()
for V in (26, 30, 36, 44):
    at = torch.randint(0, 5, (1, V), generator=g).cuda()
    xc = (torch.randn(1, V, 3, generator=g) * 0.35).cuda()
    xv = (torch.randn(1, V, 3, generator=g) * 0.5).cuda()
    mk = torch.zeros(1, V, dtype=torch.bool).cuda()
    for S in (512, 768, 1000, 1536):
        flop = 16 * V * (4478976 + 4608 * V) * S
        row = []
        for flag, name in ((16384, "narrow"), (32768, "wide"), (0, "chosen")):
            lib.tw_debug_set_flags(flag)
            m = H.tw_kernel_model(sd, path=3)
            ms = timed(lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                                              masked_elements=mk, num_samples=S))
            row.append(f"{name} {ms:6.2f} ms {flop / ms / 1e9:6.1f} TF")
        lib.tw_debug_set_flags(0)
        wg_n, wg_w = (S + 3) // 4, -(-S // (192 // V))
        print(f"V={V} S={S}: workgroups per net narrow {wg_n} / wide {wg_w} | " + " | ".join(row), flush=True)
