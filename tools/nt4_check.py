import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from tests import helpers as H
from timewarp_amd import _lib
lib = _lib.load()
d, _ = H.load("kernel_full_v60")
sd = H.full_kernel_sd()
keep = ~d["masked"][0]
KEYS = ("loglik", "s_y_coords", "s_y_velocs", "s_logp", "logp_yx")
for flags in (0, 65536):
    lib.tw_debug_set_flags(flags)
    m = H.tw_kernel_model(sd, path=3)
    out = H.run_model_case(m, d)
    errs = {k: H.rel_err(out[k][:, :, keep] if k.startswith("s_y") else out[k], d[k][:, :, keep] if k.startswith("s_y") else d[k]) for k in KEYS}
    print("flags", flags, {k: f"{v:.2e}" for k, v in errs.items()}, "demoted", m.demoted, flush=True)
# timing: config 3's size
S, V = 512, 60
g = torch.Generator().manual_seed(0)
at = torch.randint(0, 5, (1, V), generator=g).cuda()
xc = (torch.randn(1, V, 3, generator=g) * 0.5).cuda(); xv = torch.randn(1, V, 3, generator=g).cuda()
mk = torch.zeros(1, V, dtype=torch.bool).cuda()
zc = torch.randn(S, 1, V, 3, generator=g).cuda(); zv = torch.randn(S, 1, V, 3, generator=g).cuda()
for flags in (0, 65536):
    lib.tw_debug_set_flags(flags)
    m = H.tw_kernel_model(sd, path=3); m._defer_range_check += 1
    f = lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None, masked_elements=mk, num_samples=S, z_coords=zc, z_velocs=zv)
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    flop = 16 * V * (4478976 + 4608 * V) * S
    print(f"flags {flags}: reverse pass {ms:.3f} ms = {flop / ms / 1e9:.1f} TFLOP/s", flush=True)
lib.tw_debug_set_flags(0)
