"""ConditionalFlowDensityModel backed by libtimewarp_hip.so.

Same constructor-visible state, method names, keyword arguments, return shapes and error
behaviour as the reference class (modules/model_wrappers/flow.py:106-336); the arithmetic runs in
hand-written HIP kernels through the C ABI in include/timewarp_hip.h.  Inference only: outputs
carry no autograd graph (training is outside this build's scope)."""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from .. import _lib
from ..weights import FlowDims, pack_raw, strip_module_prefix
from .density_model_base import ConditionalDensityModel


# execution_path value of ConditionalFlowDensityModel only (not a C-ABI path): the split-fp16 kernel wherever the
# library supports it for the call's molecule size (tw_flow_path_supported: d_model 128; kernel attention up to 192 atoms
# - 48-token waves up to 48 atoms, 64-token waves for 49 .. 64, the wide layout from 25 - dense softmax attention up to 48), else AUTO.
PREFER_SPLIT_FP16 = -1
# Opt-in "fast" mode (TW_EXECUTION_PATH=h1): the single-MFMA kernel (TW_PATH_FUSED_H1: fp16 operands, one MFMA per product)
# wherever it applies, the split-fp16 kernel or AUTO for the rest.  NOT a parity mode: results deviate from the reference's
# fp32 arithmetic by ~1e-4 relative; never the default.
PREFER_SINGLE_FP16 = -2
_HALF_PATHS = (_lib.TW_PATH_FUSED_H3, _lib.TW_PATH_FUSED_H1, _lib.TW_PATH_SIMPLE_H3)


class ConditionalFlowDensityModel(ConditionalDensityModel):
    def __init__(self, flow: nn.Module, dims: FlowDims, scale_requires_grad: bool = True,
                 execution_path: int = _lib.TW_PATH_AUTO):
        super().__init__()
        self.flow = flow
        self.coords_prior_log_scale = nn.Parameter(torch.tensor(0.0), requires_grad=scale_requires_grad)
        self.velocs_prior_log_scale = nn.Parameter(torch.tensor(0.0), requires_grad=scale_requires_grad)
        self.dims = dims
        self.ignore_conditional_velocity = dims.ignore_cond_velocity
        self.use_displacement_as_target = dims.displacement
        self.execution_path = execution_path
        self._dev_weights = None  # {"device", "raw", "f32", "h3"}
        self._workspace = None
        self._dirty = True
        self.used_split_fp16 = False  # some call ran on the split-fp16 kernel since the last demotion (range guard)
        self.demoted = False          # the range guard has moved this model to the exact-f32 kernels
        self._defer_range_check = 0
        self._range_flags = {}        # device -> int32[1]: THIS model's range-guard word (tw_flow_desc.range_flag, ABI 7)

    # ------------------------------------------------------------------ weight cache
    def _apply(self, fn, *a, **k):
        self._dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._dirty = True
        return super().load_state_dict(strip_module_prefix(state_dict), strict=strict, **kw)

    def refresh_weights(self) -> None:
        """Re-pack the device weight buffers after parameters were modified in place."""
        self._dirty = True

    def _path_for(self, n_atoms: int) -> int:
        """The C-ABI execution path of a call on molecules of `n_atoms` atoms."""
        path = self.execution_path
        if path in (PREFER_SPLIT_FP16, PREFER_SINGLE_FP16):
            desc = self.dims.to_desc()
            sup = _lib.load().tw_flow_path_supported
            if path == PREFER_SINGLE_FP16 and sup(C.byref(desc), int(n_atoms), _lib.TW_PATH_FUSED_H1) == 1:
                path = _lib.TW_PATH_FUSED_H1
            else:
                ok = sup(C.byref(desc), int(n_atoms), _lib.TW_PATH_FUSED_H3) == 1
                if ok:
                    path = _lib.TW_PATH_FUSED_H3
                elif sup(C.byref(desc), int(n_atoms), _lib.TW_PATH_FUSED) == 1:
                    path = _lib.TW_PATH_AUTO          # the exact-f32 fused kernel serves the shape
                else:
                    # r06: no fused layout at all (kernel attention above 192 atoms, dense softmax above 64, other widths):
                    # the per-op path with its linears on split-fp16 MFMAs instead of the fp32 matrix pipe
                    path = _lib.TW_PATH_SIMPLE_H3
        if path in _HALF_PATHS:
            self.used_split_fp16 = True
        return path

    # ------------------------------------------------------------------ split-fp16 range guard
    # The split-fp16 kernel holds its operands in fp16: a checkpoint whose activations leave +-65504 makes it return
    # non-finite scale / shift values where the exact-f32 kernels would not.  The coupling step raises a sticky device
    # flag (tw_flow_nonfinite).  Policy: never sample through it and never abort a chain over it - the model is DEMOTED
    # to the exact-f32 kernels (execution_path = TW_PATH_AUTO, one warning) and the affected work is redone there: a
    # public call re-runs itself, the MH loops replay the iterations since their last read-back from the recorded
    # draws (utils/evaluation_utils.py).  `check_finite` is the raising form for callers that drive the C ABI themselves.
    def _desc(self, device):
        """The C descriptor of a call on `device`, carrying this model's own range-guard word there (ABI 7): two models on
        one device no longer see - or swallow - each other's overflow, as they did with the per-device word."""
        d = self.dims.to_desc()
        device = torch.device(device)
        if device.type == "cuda":
            key = device.index if device.index is not None else torch.cuda.current_device()
            if key not in self._range_flags:
                self._range_flags[key] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
            d.range_flag = self._range_flags[key].data_ptr()
        return d

    def split_fp16_overflowed(self, device=None) -> bool:
        """True if a split-fp16 launch of THIS model produced non-finite coupling parameters since the last call (reads and
        clears the model's flag word on that device; synchronises).  Always False for a model that never ran on that kernel."""
        if not self.used_split_fp16:
            return False
        key = torch.device(device).index if device is not None else None
        if key is None:
            key = torch.cuda.current_device()
        flag = self._range_flags.get(key)
        if flag is None:
            # no word of its own on that device (a descriptor built through dims.to_desc(): raw C-ABI use of this model's
            # weights, code written against ABI 6): such calls report to the per-device word - look there
            out = C.c_int32(0)
            with torch.cuda.device(key):
                _lib.check(_lib.load().tw_flow_nonfinite(1, C.byref(out)), "tw_flow_nonfinite")
            return out.value != 0
        if int(flag.item()) == 0:
            return False
        flag.zero_()
        return True

    # the range-guard words are per-process device scratch, not state: a pickled / deep-copied model gets fresh ones lazily
    def __getstate__(self):
        state = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        state = dict(state)
        state["_range_flags"] = {}
        state["_dev_weights"], state["_workspace"], state["_dirty"] = None, None, True
        return state

    def __deepcopy__(self, memo):
        import copy

        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_range_flags":
                new.__dict__[k] = {}
            elif k in ("_dev_weights", "_workspace"):
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_dirty"] = True
        return new

    def demote_to_f32(self) -> None:
        """Leave the split-fp16 kernel for good: every later call runs on the exact-f32 kernels."""
        if self.execution_path in (PREFER_SPLIT_FP16, PREFER_SINGLE_FP16) + _HALF_PATHS:
            warnings.warn(
                "timewarp_amd: this checkpoint's activations leave the fp16 range (+-65504) on the split-fp16 kernel; "
                "switching this model to the exact-f32 kernels and redoing the affected calls there "
                "(TW_EXECUTION_PATH=f32 selects them from the start).", RuntimeWarning, stacklevel=3)
            self.execution_path = _lib.TW_PATH_AUTO
        self.used_split_fp16 = False
        self.demoted = True

    class _Deferred:
        def __init__(self, model):
            self.model = model

        def __enter__(self):
            self.model._defer_range_check += 1

        def __exit__(self, *exc):
            self.model._defer_range_check -= 1

    def deferred_range_check(self):
        """Context manager: public calls inside it do not synchronise to look at the range flag - the caller does
        (split_fp16_overflowed) where it reads results back, and redoes the work itself (the MH loops)."""
        return self._Deferred(self)

    def _guarded(self, device, n_atoms: int, run):
        """run(path) on the call's execution path; on a split-fp16 range overflow, demote and run again."""
        path = self._path_for(n_atoms)
        out = run(path)
        if path in _HALF_PATHS and self._defer_range_check == 0 and self.split_fp16_overflowed(device):
            self.demote_to_f32()
            out = run(self._path_for(n_atoms))
        return out

    def check_finite(self, device=None) -> None:
        """Raise if the split-fp16 kernel produced non-finite coupling parameters since the last check.  For callers of
        the C ABI / `deferred_range_check` users that prefer an error over the demotion above.  Synchronises."""
        if self.split_fp16_overflowed(device):
            raise RuntimeError(
                "timewarp_amd: the split-fp16 execution path returned non-finite scale/shift values - this checkpoint's "
                "activations leave the fp16 range.  Use the exact-f32 kernels (TW_EXECUTION_PATH=f32, or "
                "model.execution_path = TW_PATH_AUTO).")

    def _weights(self, device: torch.device, path: Optional[int] = None):
        """(raw, packed) device buffers; `packed` is the stream of the active execution path (the
        f32 fragment stream, or the split-fp16 stage stream for TW_PATH_FUSED_H3), built lazily."""
        if self.training:
            self._dirty = True  # parameters may be changing under us: never serve a stale pack
        if self._dirty or self._dev_weights is None or self._dev_weights["device"] != device:
            lib = _lib.load()
            desc = self.dims.to_desc()
            raw_cpu = pack_raw(self.state_dict(), self.dims)
            expect = lib.tw_flow_raw_floats(C.byref(desc))
            if expect != raw_cpu.numel():
                raise RuntimeError(f"raw weight layout mismatch: host packed {raw_cpu.numel()} floats, library expects {expect}")
            self._dev_weights = {"device": device, "raw": raw_cpu.to(device), "f32": None, "h3": None, "h1": None}
            self._dirty = False
        w = self._dev_weights
        lib = _lib.load()
        desc = self.dims.to_desc()
        path = self.execution_path if path is None else path
        if path == _lib.TW_PATH_FUSED_H1:
            if w["h1"] is None:
                n = lib.tw_flow_packed_h1_bytes(C.byref(desc))
                if n <= 0:
                    raise RuntimeError("the single-MFMA path does not support this model configuration")
                buf = torch.empty(n, dtype=torch.uint8, device=device)
                with torch.cuda.device(device):
                    _lib.check(lib.tw_flow_pack_h1(C.byref(desc), w["raw"].data_ptr(), buf.data_ptr(),
                                                   _lib.stream_ptr(device)), "tw_flow_pack_h1")
                w["h1"] = buf
            return w["raw"], w["h1"]
        if path == _lib.TW_PATH_SIMPLE_H3:
            # the per-op path's own pack: the split-fp16 stream (fused FFN launches) + the folded attention projections
            if "s5" not in w:
                n = lib.tw_flow_packed_simple_h3_bytes(C.byref(desc))
                w["s5"] = None
                if n > 0:
                    buf = torch.empty(n, dtype=torch.uint8, device=device)
                    with torch.cuda.device(device):
                        _lib.check(lib.tw_flow_pack_simple_h3(C.byref(desc), w["raw"].data_ptr(), buf.data_ptr(),
                                                              _lib.stream_ptr(device)), "tw_flow_pack_simple_h3")
                    w["s5"] = buf
            return w["raw"], w["s5"]     # (None: no split-fp16 stream for this width - every linear as its own GEMM)
        if path == _lib.TW_PATH_FUSED_H3:
            if w["h3"] is None:
                n = lib.tw_flow_packed_h3_bytes(C.byref(desc))
                if n <= 0:
                    raise RuntimeError("the split-fp16 path does not support this model configuration")
                buf = torch.empty(n, dtype=torch.uint8, device=device)
                with torch.cuda.device(device):
                    _lib.check(lib.tw_flow_pack_h3(C.byref(desc), w["raw"].data_ptr(), buf.data_ptr(),
                                                   _lib.stream_ptr(device)), "tw_flow_pack_h3")
                w["h3"] = buf
            return w["raw"], w["h3"]
        if w["f32"] is None and path not in (_lib.TW_PATH_SIMPLE, _lib.TW_PATH_SIMPLE_H3):
            n = lib.tw_flow_packed_floats(C.byref(desc))
            if n > 0:
                buf = torch.empty(n, dtype=torch.float32, device=device)
                with torch.cuda.device(device):
                    _lib.check(lib.tw_flow_pack(C.byref(desc), w["raw"].data_ptr(), buf.data_ptr(),
                                                _lib.stream_ptr(device)), "tw_flow_pack")
                w["f32"] = buf
        return w["raw"], w["f32"]

    def _ws(self, device: torch.device, n_rows: int, n_atoms: int):
        lib = _lib.load()
        desc = self.dims.to_desc()
        need = lib.tw_flow_workspace_bytes(C.byref(desc), n_rows, n_atoms)
        if need < 0:
            raise RuntimeError("tw_flow_workspace_bytes failed: " + lib.tw_last_error().decode())
        if self._workspace is None or self._workspace.device != device or self._workspace.numel() < need:
            self._workspace = torch.empty(int(need), dtype=torch.uint8, device=device)
        return self._workspace

    @staticmethod
    def _prep(atom_types, masked_elements, *floats):
        dev = floats[0].device
        out = [_lib.require_gpu_tensor(atom_types, torch.int32, "atom_types"),
               _lib.require_gpu_tensor(masked_elements.to(torch.uint8), torch.uint8, "masked_elements")]
        out += [_lib.require_gpu_tensor(f.to(dev), torch.float32, "coords/velocs") for f in floats]
        return out

    # ------------------------------------------------------------------ API (flow.py:131-336)
    @torch.no_grad()
    def log_likelihood(self, atom_types: Tensor, x_coords: Tensor, x_velocs: Tensor, y_coords: Tensor,
                       y_velocs: Tensor, adj_list: Optional[Tensor], edge_batch_idx: Optional[Tensor],
                       masked_elements: Tensor, logger=None) -> Tensor:
        """log p(y | x) per batch element, [B] (flow.py:131-215).  adj_list / edge_batch_idx are
        accepted and ignored, as in the reference."""
        at, mk, xc, xv, yc, yv = self._prep(atom_types, masked_elements, x_coords, x_velocs, y_coords, y_velocs)
        dev = xc.device
        B, V = xc.shape[0], xc.shape[1]
        ws = self._ws(dev, B, V)
        out = torch.empty(B, dtype=torch.float32, device=dev)
        lib = _lib.load()
        desc = self._desc(dev)

        def run(path):
            raw, packed = self._weights(dev, path)
            with torch.cuda.device(dev):
                _lib.check(lib.tw_flow_log_likelihood(
                    C.byref(desc), raw.data_ptr(), _lib.ptr(packed), at.data_ptr(), xc.data_ptr(), xv.data_ptr(),
                    yc.data_ptr(), yv.data_ptr(), mk.data_ptr(), out.data_ptr(), B, V, path,
                    ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)), "tw_flow_log_likelihood")
            return out

        return self._guarded(dev, V, run)

    def conditional_sample(self, atom_types, x_coords, x_velocs, adj_list, edge_batch_idx, masked_elements,
                           num_samples: int, logger=None) -> Tuple[Tensor, Tensor]:
        y_c, y_v, _ = self.conditional_sample_with_logp(
            atom_types=atom_types, x_coords=x_coords, x_velocs=x_velocs, adj_list=adj_list,
            edge_batch_idx=edge_batch_idx, masked_elements=masked_elements, num_samples=num_samples, logger=logger)
        return y_c, y_v

    @torch.no_grad()
    def conditional_sample_with_logp(self, atom_types: Tensor, x_coords: Tensor, x_velocs: Tensor,
                                     adj_list: Optional[Tensor], edge_batch_idx: Optional[Tensor],
                                     masked_elements: Tensor, num_samples: int, logger=None,
                                     z_coords: Optional[Tensor] = None, z_velocs: Optional[Tensor] = None,
                                     allow_multi: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
        """Samples y ~ p(.|x) and their log-density: ([S,B,V,3], [S,B,V,3], [S,B]) (flow.py:242-336).

        Extension over the reference signature: `z_coords` / `z_velocs` [S,B,V,3] may carry the
        latent noise explicitly (already scaled by exp(prior log-scale)); when omitted it is drawn
        on the device in the reference's order (coords first, flow.py:274-275).  `allow_multi=True` lifts the
        reference's B == 1 or S == 1 restriction (several chains' proposals in one launch)."""
        at, mk, xc, xv = self._prep(atom_types, masked_elements, x_coords, x_velocs)
        dev = xc.device
        B, V = xc.shape[0], xc.shape[1]
        S = int(num_samples)
        if not (B == 1 or S == 1 or allow_multi):
            # flow.py:326 multiplies a [B,V,1] mask into [S*B,V,3]: torch raises for B>1 and S>1
            raise RuntimeError(f"The size of tensor a ({B}) must match the size of tensor b ({S * B}) at non-singleton dimension 0")
        if z_coords is None:
            z_coords = torch.randn((S, B, V, 3), device=dev) * torch.exp(self.coords_prior_log_scale.detach()).to(dev)
        if z_velocs is None:
            z_velocs = torch.randn((S, B, V, 3), device=dev) * torch.exp(self.velocs_prior_log_scale.detach()).to(dev)
        zc = _lib.require_gpu_tensor(z_coords.to(dev), torch.float32, "z_coords")
        zv = _lib.require_gpu_tensor(z_velocs.to(dev), torch.float32, "z_velocs")
        if tuple(zc.shape) != (S, B, V, 3) or tuple(zv.shape) != (S, B, V, 3):
            raise ValueError("z_coords / z_velocs must have shape [num_samples, B, V, 3]")
        ws = self._ws(dev, S * B, V)
        y_c = torch.empty((S, B, V, 3), dtype=torch.float32, device=dev)
        y_v = torch.empty((S, B, V, 3), dtype=torch.float32, device=dev)
        logp = torch.empty((S, B), dtype=torch.float32, device=dev)
        lib = _lib.load()
        desc = self._desc(dev)

        def run(path):
            raw, packed = self._weights(dev, path)
            with torch.cuda.device(dev):
                entry = lib.tw_flow_sample_with_logp_multi if allow_multi else lib.tw_flow_sample_with_logp
                _lib.check(entry(
                    C.byref(desc), raw.data_ptr(), _lib.ptr(packed), at.data_ptr(), xc.data_ptr(), xv.data_ptr(),
                    mk.data_ptr(), zc.data_ptr(), zv.data_ptr(), y_c.data_ptr(), y_v.data_ptr(), logp.data_ptr(),
                    S, B, V, path, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)),
                    "tw_flow_sample_with_logp")
            return y_c, y_v, logp

        return self._guarded(dev, V, run)

    # ------------------------------------------------------------------ inspection (tests)
    @torch.no_grad()
    def debug_netblock(self, coupling: int, net: int, atom_types, x_coords_centred, x_velocs, masked_elements,
                       z_other, path: int):
        """Run one coupling net and return the activations after in_mlp, every encoder layer and
        out_mlp: ([n_layers+1, N, V, d_model], [N, V, 3]).  Test hook for the HIP kernels."""
        at, mk, xc, xv, zo = self._prep(atom_types, masked_elements, x_coords_centred, x_velocs, z_other)
        dev = xc.device
        n_cond, V = xc.shape[0], xc.shape[1]
        N = zo.shape[0]
        raw, packed = self._weights(dev, path)
        ws = self._ws(dev, N, V)
        L, dm = self.dims.n_layers, self.dims.d_model
        dump = torch.zeros((L + 1) * N * V * dm + N * V * 3, dtype=torch.float32, device=dev)
        lib = _lib.load()
        desc = self._desc(dev)
        with torch.cuda.device(dev):
            _lib.check(lib.tw_debug_netblock(
                C.byref(desc), raw.data_ptr(), _lib.ptr(packed), coupling, net, at.data_ptr(), xc.data_ptr(),
                xv.data_ptr(), mk.data_ptr(), n_cond, zo.data_ptr(), N, V, path, dump.data_ptr(), ws.data_ptr(),
                ws.numel(), _lib.stream_ptr(dev)), "tw_debug_netblock")
        acts = dump[: (L + 1) * N * V * dm].reshape(L + 1, N, V, dm)
        out = dump[(L + 1) * N * V * dm:].reshape(N, V, 3)
        return acts, out
