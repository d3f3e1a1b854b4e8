"""Errors of every execution path on the cases above 25 atoms (torch.cdist's matmul branch): the 60-atom golden from the
reference, and the score matrix itself against the oracle's (torch-CPU) scores."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flow_oracle as fo
from tests import helpers as H
import ctypes as C
from timewarp_amd import _lib

d, _ = H.load("kernel_full_v60")
sd = H.full_kernel_sd()
for path in (2, 1, 3):
    m = H.tw_kernel_model(sd, path=path)
    out = H.run_model_case(m, d)
    keep = ~d["masked"][0]
    e = {k: H.rel_err(out[k][:, :, keep] if k.startswith("s_y") else out[k], d[k][:, :, keep] if k.startswith("s_y") else d[k])
         for k in ("loglik", "s_y_coords", "s_y_velocs", "s_logp", "logp_yx")}
    print("v60 golden path", path, {k: f"{v:.2e}" for k, v in e.items()})
# scores: tw_kernel_scores vs oracle
lib = _lib.load()
for V in (26, 40, 60, 100):
    g = torch.Generator().manual_seed(V)
    x = torch.randn(3, V, 3, generator=g) * 0.5
    mk = torch.zeros(3, V, dtype=torch.bool)
    ls = torch.tensor([0.1, 0.2, 0.5, 0.7, 1.0, 1.2])
    ref = fo.kernel_scores(x, mk, ls)
    out = torch.empty(3, 6, V, V, device="cuda")
    xc, mc, lc = x.cuda(), mk.to(torch.uint8).cuda(), ls.cuda()
    _lib.check(lib.tw_kernel_scores(xc.data_ptr(), mc.data_ptr(), lc.data_ptr(), 6, 3, V, 1, 1, out.data_ptr(), None), "scores")
    diff = (out.cpu() - ref).abs()
    print(f"scores V={V}: max abs diff {float(diff.max()):.2e} (max score {float(ref.max()):.2f}), identical entries {float((out.cpu() == ref).float().mean()):.4f}")
