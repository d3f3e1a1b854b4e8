"""Repeatability of the wide layout's three statements (three-, five-, six-group windows) and of the fast path on them: the same
ragged batch N times with the caches flushed and another kernel family's bytes left in LDS in between - every run must return
the first run's bits (profiles/r04_stress_wide.txt).  STRESS_N=30 by default."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from timewarp_amd import _lib

sd = H.full_kernel_sd()
lib = _lib.load()


def case(V, lens, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    at = torch.randint(0, 5, (B, V), generator=g)
    x_c = torch.randn(B, V, 3, generator=g) * 0.5
    x_v = torch.randn(B, V, 3, generator=g) * 0.5
    y_c = x_c + torch.randn(B, V, 3, generator=g) * 0.02
    y_v = torch.randn(B, V, 3, generator=g) * 0.5
    mask = torch.zeros(B, V, dtype=torch.bool)
    for b, n in enumerate(lens):
        mask[b, n:] = True
    return [t.cuda() for t in (at, x_c, x_v, y_c, y_v, mask)]


def ll(m, c):
    at, x_c, x_v, y_c, y_v, mask = c
    return m.log_likelihood(atom_types=at, x_coords=x_c, x_velocs=x_v, y_coords=y_c, y_velocs=y_v, adj_list=None,
                            edge_batch_idx=None, masked_elements=mask).cpu()


N = int(os.environ.get("STRESS_N", "30"))
junk = torch.empty(1 << 29, dtype=torch.float32, device="cuda")  # 2 GiB: evicts L2 and the Infinity Cache
other = H.tw_kernel_model(sd, path=1)
small = case(22, [22] * 64, 1)
for label, V, lens, flag in (("30 atoms, forced wide (five groups)", 30, [30, 28, 30, 30, 25, 30, 30] * 40, 32768),
                             ("65 atoms (three groups, stride 96)", 65, [65, 60, 65] * 60, 0),
                             ("88 atoms (three groups, stride 96)", 88, [88, 88, 61, 88, 88] * 40, 0),
                             ("100 atoms (five groups)", 100, [100, 87, 100] * 40, 0),
                             ("176 atoms (six groups)", 176, [176, 150, 176] * 30, 0)):
    c = case(V, lens, 7 + V)
    for path, name in ((3, "split-fp16"), (4, "fast mode")):
        lib.tw_debug_set_flags(flag)
        m = H.tw_kernel_model(sd, path=path)
        S = 2 * len(lens) // 3
        g = torch.Generator().manual_seed(V)
        zc, zv = torch.randn(S, 1, V, 3, generator=g).cuda() * 0.1, torch.randn(S, 1, V, 3, generator=g).cuda()
        rev = lambda: torch.cat([t.reshape(-1).cpu() for t in m.conditional_sample_with_logp(
            atom_types=c[0][:1], x_coords=c[1][:1], x_velocs=c[2][:1], adj_list=None, edge_batch_idx=None,
            masked_elements=c[5][:1] & False, num_samples=S, z_coords=zc, z_velocs=zv)])
        for what, fn in (("forward", lambda: ll(m, c)), ("reverse", rev)):
            first = fn()
            bad = 0
            for it in range(N):
                junk.fill_(float(it))
                ll(other, small)   # another kernel family leaves its bytes in LDS
                bad += not torch.equal(fn(), first)
            print(f"{label}, {len(lens) if what == 'forward' else S} rows, {name}, {what} pass: {bad}/{N} runs differ from the first; "
                  f"finite: {bool(torch.isfinite(first).all())}", flush=True)
        lib.tw_debug_set_flags(0)
