"""Shared test helpers: golden loading and the name-seeded full-size weights."""
import os

import numpy as np
import torch

from oracle import flow_oracle as fo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FULL_KERNEL_SPEC = fo.FlowSpec(variant="kernel")
FULL_DENSE_SPEC = fo.FlowSpec(variant="dense", n_head=8)
TINY_KERNEL_SPEC = fo.FlowSpec(variant="kernel", num_coupling_layers=2, num_transformer_layers=2)
TINY_DENSE_SPEC = fo.FlowSpec(variant="dense", num_coupling_layers=2, num_transformer_layers=2, n_head=2)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    data = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("sd::") and z[k].dtype.kind != "U"}
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return data, sd


_cache = {}


def full_kernel_sd(calibrated=False):
    key = ("k", calibrated)
    if key not in _cache:
        _cache[key] = fo.synth_state_dict(fo.make_template(FULL_KERNEL_SPEC), 0, calibrated)
    return _cache[key]


def full_dense_sd():
    if "d" not in _cache:
        _cache["d"] = fo.synth_state_dict(fo.make_template(FULL_DENSE_SPEC), 0)
    return _cache["d"]


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
