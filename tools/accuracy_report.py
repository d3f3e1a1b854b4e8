"""Actual relative errors of every execution path against the reference vectors (GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402

from timewarp_amd import _lib  # noqa: E402

for name, cal in (("kernel_full_ad", False), ("kernel_cheb_full_ad", False)):
    d, _ = H.load(name)
    for path, label in ((2, "simple"), (1, "fused-f32"), (3, "fused-h3"), (3, "fused-h3 per-section build (flag 4096)")):
        _lib.load().tw_debug_set_flags(4096 if "4096" in label else 0)
        if "cheb" in name:
            m = H.tw_kernel_model(H.full_cheb_sd(), path=path, attention_type="chebyshev_kernel", cheb_order=6, force_asymptotic_zero=True)
        else:
            m = H.tw_kernel_model(H.full_kernel_sd(cal), path=path)
        out = H.run_model_case(m, d)
        keep = ~d["masked"][0]
        errs = {k: H.rel_err(out[k][:, :, keep] if out[k].dim() == 4 else out[k], d[k][:, :, keep] if d[k].dim() == 4 else d[k])
                for k in ("loglik", "s_y_coords", "s_y_velocs", "s_logp", "logp_yx")}
        print(name, label, {k: f"{v:.2e}" for k, v in errs.items()}, flush=True)
_lib.load().tw_debug_set_flags(0)
