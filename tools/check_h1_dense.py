"""The dense model (transformer_nvp, BASELINE configs[4]) on the fast path (TW_PATH_FUSED_H1: in / FFN / out sections with one
fp16 MFMA per product, the softmax attention block kept in split-fp16 form) next to the split-fp16 kernel: error against the
reference vectors (plain and padded goldens) and the time of a 1000-proposal reverse + forward pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as H

KEYS = ("loglik", "s_y_coords", "s_y_velocs", "s_logp", "logp_yx")
for name in ("dense_full_ad", "dense_full_padded"):
    d, _ = H.load(name)
    keep = ~d["masked"][0]
    for path in (3, 4):
        m = H.tw_dense_model(H.full_dense_sd(), path=path)
        out = H.run_model_case(m, d)
        sel = lambda k, t: t[:, :, keep] if k.startswith("s_y") else t
        errs = {k: H.rel_err(sel(k, out[k]), sel(k, d[k])) for k in KEYS if k in out}
        print(name, "path", path, {k: f"{v:.2e}" for k, v in errs.items()}, "demoted", m.demoted, flush=True)

d, _ = H.load("dense_full_ad")
S = 1000
g = torch.Generator().manual_seed(1)
for path in (3, 4):
    m = H.tw_dense_model(H.full_dense_sd(), path=path)
    m._defer_range_check += 1
    a = {k: d[k].cuda() for k in ("atom_types", "x_coords", "x_velocs", "masked")}
    zc = torch.randn(S, 1, 22, 3, generator=g).cuda()
    zv = torch.randn(S, 1, 22, 3, generator=g).cuda()

    def one():
        yc, yv, lp = m.conditional_sample_with_logp(atom_types=a["atom_types"], x_coords=a["x_coords"], x_velocs=a["x_velocs"],
                                                    adj_list=None, edge_batch_idx=None, masked_elements=a["masked"], num_samples=S,
                                                    z_coords=zc, z_velocs=zv)
        return yc
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    flop = 16 * 22 * 3726336 * S   # dense F_blk(22) per token per net block
    print(f"path {path}: reverse pass of {S} proposals {ms:.3f} ms = {flop / ms / 1e9:.1f} TFLOP/s algorithmic", flush=True)
