"""Reverse-pass time of the kernel flow on the PER-OP path at given (atoms, proposals) pairs - the path every molecule above 192
atoms takes: `python tools/time_per_op.py 192x256 691x16` (under `rocprofv3 --kernel-trace --stats` for the per-kernel shares)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H

# --path=2 (default): the exact-f32 per-op kernels; --path=5: TW_PATH_SIMPLE_H3, the linears as split-fp16 MFMA GEMMs (r06)
path = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--path=")), 2)
# --dense: the dense softmax variant (transformer_nvp) instead of kernel attention
dense = "--dense" in sys.argv
m = H.tw_dense_model(H.full_dense_sd(), path=path) if dense else H.tw_kernel_model(H.full_kernel_sd(), path=path)
# --flags=N: tw_debug_set_flags(N) for A/B runs (e.g. 268435456: in / out MLPs as GEMM pairs)
flags = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--flags=")), 0)
if flags:
    from timewarp_amd import _lib
    _lib.load().tw_debug_set_flags(flags)
g = torch.Generator().manual_seed(0)
for spec in [a for a in sys.argv[1:] if not a.startswith("--")] or ["192x256"]:
    V, S = (int(t) for t in spec.split("x"))
    at = torch.randint(0, 5, (1, V), generator=g).cuda()
    xc = (torch.randn(1, V, 3, generator=g) * 0.8).cuda()
    xv = torch.randn(1, V, 3, generator=g).cuda()
    mk = torch.zeros(1, V, dtype=torch.bool).cuda()
    f = lambda: m.conditional_sample_with_logp(atom_types=at, x_coords=xc, x_velocs=xv, adj_list=None, edge_batch_idx=None,
                                               masked_elements=mk, num_samples=S)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    flop = 16 * V * ((3692544 + 1536 * V) if dense else (4478976 + 4608 * V)) * S   # 2 x MACs per token: linears + mixing
    if "--forward" in sys.argv:
        # the reverse move of an MH iteration: every row conditioned on its own state (n_cond = S: scores per row)
        yc, yv, _ = f()
        atS, mkS = at.repeat(S, 1), mk.repeat(S, 1)
        ff = lambda: m.log_likelihood(atom_types=atS, x_coords=yc.squeeze(1), x_velocs=yv.squeeze(1), y_coords=xc.repeat(S, 1, 1),
                                      y_velocs=xv.repeat(S, 1, 1), adj_list=None, edge_batch_idx=None, masked_elements=mkS)
        ff(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            ff()
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / 3 * 1e3
        print(f"per-op path {path} V={V} S={S}: {dtf:.1f} ms per FORWARD pass (n_cond = S), {flop / dtf / 1e9:.1f} TFLOP/s algorithmic", flush=True)
    print(f"per-op path {path} V={V} S={S}: {dt:.1f} ms per reverse pass, {flop / dt / 1e9:.1f} TFLOP/s algorithmic", flush=True)
