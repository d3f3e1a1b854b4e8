"""First look at the single-MFMA path (TW_PATH_FUSED_H1): error against the reference vectors next to the split-fp16
path, on the un-calibrated full-size model (worst case) and on the bench calibration; then timing (tools/time_flow.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402

for name, cal in (("kernel_full_ad", False), ("kernel_full_ad_calibrated", True)):
    d, _ = H.load(name)
    sd = H.full_kernel_sd(calibrated=cal)
    keep = ~d["masked"][0]
    for path in (3, 4):
        m = H.tw_kernel_model(sd, path=path)
        out = H.run_model_case(m, d)
        errs = {k: H.rel_err(out[k][:, :, keep] if k.startswith("s_y") else out[k], d[k][:, :, keep] if k.startswith("s_y") else d[k])
                for k in ("loglik", "s_y_coords", "s_y_velocs", "s_logp", "logp_yx")}
        absl = float((out["s_logp"] - d["s_logp"]).abs().max()), float((out["logp_yx"] - d["logp_yx"]).abs().max())
        print(name, "path", path, {k: f"{v:.2e}" for k, v in errs.items()}, "abs logp", absl, "nan", bool(torch.isnan(out["s_logp"]).any()), flush=True)
