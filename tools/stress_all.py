"""Bit-for-bit repeatability of every fused kernel family on both half-precision modes, forward and reverse pass: the same
inputs N times (STRESS_N, default 300) with the caches flushed and another kernel family's bytes left in LDS in between
(profiles/r04_stress_all.txt).  Companion of stress_wide.py, written after that one found the prefetch-at-exit race."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from oracle import flow_oracle as fo

N = int(os.environ.get("STRESS_N", "300"))
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
other = H.tw_kernel_model(H.full_kernel_sd(), path=1)


def batch(V, B, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    at = torch.randint(0, 5, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * (0.3 if V <= 30 else 0.5)
    x_v = torch.randn(B, V, 3, generator=g) * 0.5
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.02
    y_v = torch.randn(B, V, 3, generator=g) * 0.5
    mask = torch.zeros(B, V, dtype=torch.bool)
    if ragged:
        for b in range(0, B, 3):
            mask[b, V - 1 - (b % 5):] = True
    return [t.cuda() for t in (at, x_c, x_v, y_c, y_v, mask)]


small = batch(22, 64, 1)


def disturb(it):
    junk.fill_(float(it))
    at, x_c, x_v, y_c, y_v, mask = small
    other.log_likelihood(atom_types=at, x_coords=x_c, x_velocs=x_v, y_coords=y_c, y_velocs=y_v, adj_list=None,
                         edge_batch_idx=None, masked_elements=mask)


for label, dense, V, B, S in (("alanine dipeptide, kernel attention", False, 22, 1000, 1000), ("60 atoms (64-token waves)", False, 60, 512, 512),
                              ("44 atoms (48-token waves, one molecule each)", False, 44, 400, 400), ("dense softmax model", True, 22, 1000, 1000),
                              # r05: the statements that are new this round
                              ("65 atoms (wide layout, three-group windows)", False, 65, 512, 512),
                              ("110 atoms (paired 64-token waves)", False, 110, 256, 256),
                              ("150 atoms (wide layout, five-group windows)", False, 150, 200, 200),
                              ("176 atoms (wide layout, six-group windows)", False, 176, 128, 128)):
    sd = H.full_dense_sd() if dense else H.full_kernel_sd()
    c = batch(V, B, 11 + V)
    at, x_c, x_v, y_c, y_v, mask = c
    g = torch.Generator().manual_seed(5)
    zc, zv = torch.randn(S, 1, V, 3, generator=g).cuda() * 0.1, torch.randn(S, 1, V, 3, generator=g).cuda()
    for path, name in ((3, "split-fp16"), (4, "fast mode")):
        m = H.tw_dense_model(sd, path=path) if dense else H.tw_kernel_model(sd, path=path)
        fwd = lambda: m.log_likelihood(atom_types=at, x_coords=x_c, x_velocs=x_v, y_coords=y_c, y_velocs=y_v, adj_list=None,
                                       edge_batch_idx=None, masked_elements=mask).cpu()
        rev = lambda: torch.cat([t.reshape(-1).cpu() for t in m.conditional_sample_with_logp(
            atom_types=at[:1], x_coords=x_c[:1], x_velocs=x_v[:1], adj_list=None, edge_batch_idx=None, masked_elements=mask[:1] & False,
            num_samples=S, z_coords=zc, z_velocs=zv)])
        for what, fn in (("forward", fwd), ("reverse", rev)):
            first = fn()
            bad = 0
            for it in range(N):
                disturb(it)
                bad += not torch.equal(fn(), first)
            print(f"{label}, {name}, {what} pass: {bad}/{N} runs differ from the first; finite: {bool(torch.isfinite(first).all())}", flush=True)
