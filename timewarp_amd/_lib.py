"""ctypes binding of libtimewarp_hip.so (include/timewarp_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
`import torch` must happen before the library is loaded so that the process has exactly one HIP
runtime (torch's bundled libamdhip64.so.7, same soname as /opt/rocm's).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (loads the HIP runtime first)

from . import build as _build

_LIB: Optional[C.CDLL] = None

TW_PATH_AUTO, TW_PATH_FUSED, TW_PATH_SIMPLE, TW_PATH_FUSED_H3, TW_PATH_FUSED_H1, TW_PATH_SIMPLE_H3 = 0, 1, 2, 3, 4, 5


class FlowDesc(C.Structure):
    """Mirror of `tw_flow_desc`."""

    _fields_ = [
        ("variant", C.c_int32),
        ("n_coupling", C.c_int32),
        ("n_layers", C.c_int32),
        ("d_model", C.c_int32),
        ("d_ff", C.c_int32),
        ("d_hidden", C.c_int32),
        ("d_emb", C.c_int32),
        ("n_heads", C.c_int32),
        ("d_rff", C.c_int32),
        ("n_elements", C.c_int32),
        ("pos_mod2", C.c_int32),
        ("displacement", C.c_int32),
        ("ignore_cond_velocity", C.c_int32),
        ("normalise", C.c_int32),
        ("ln_eps", C.c_float),
        ("cheb_order", C.c_int32),
        ("cheb_force_zero", C.c_int32),
        ("range_flag", C.c_void_p),   # ABI 7: device int32 the flow kernels raise on a non-finite scale / shift (None: per device)
    ]


class ForceField(C.Structure):
    """Mirror of `tw_forcefield` (all pointers are device pointers)."""

    _fields_ = [
        ("n_atoms", C.c_int32),
        ("n_bonds", C.c_int32),
        ("n_angles", C.c_int32),
        ("n_torsions", C.c_int32),
        ("n_exceptions", C.c_int32),
        ("has_gbsa", C.c_int32),
        ("cutoff", C.c_double),
        ("rf_dielectric", C.c_double),
        ("solute_dielectric", C.c_double),
        ("solvent_dielectric", C.c_double),
        ("surface_area_energy", C.c_double),
        ("bond_idx", C.c_void_p),
        ("bond_par", C.c_void_p),
        ("angle_idx", C.c_void_p),
        ("angle_par", C.c_void_p),
        ("torsion_idx", C.c_void_p),
        ("torsion_par", C.c_void_p),
        ("exc_idx", C.c_void_p),
        ("exc_par", C.c_void_p),
        ("atom_par", C.c_void_p),
    ]


class MHOptions(C.Structure):
    """Mirror of `tw_mh_options`."""

    _fields_ = [
        ("random_velocs", C.c_int32),
        ("n_centres", C.c_int32),
        ("centres", C.c_void_p),
        ("reference_signs", C.c_void_p),
        ("masses", C.c_void_p),
        ("kbT", C.c_float),
    ]


class MHDraws(C.Structure):
    """Mirror of `tw_mh_draws` (ABI 8)."""

    _fields_ = [("seed", C.c_uint64), ("iteration", C.c_int64), ("first_chain", C.c_int32), ("resample_velocs", C.c_int32)]


ABI_VERSION = 8
_P = C.c_void_p
_I32, _I64, _F = C.c_int32, C.c_int64, C.c_float
_DESC = C.POINTER(FlowDesc)

# name -> (restype, argtypes); every symbol include/timewarp_hip.h declares
SIGNATURES = {
    "tw_last_error": (C.c_char_p, []),
    "tw_abi_version": (C.c_int, []),
    "tw_device_count": (C.c_int, []),
    "tw_flow_raw_floats": (_I64, [_DESC]),
    "tw_flow_packed_floats": (_I64, [_DESC]),
    "tw_flow_pack": (C.c_int, [_DESC, _P, _P, _P]),
    "tw_flow_packed_simple_h3_bytes": (_I64, [_DESC]),
    "tw_flow_pack_simple_h3": (C.c_int, [_DESC, _P, _P, _P]),
    "tw_flow_packed_h3_bytes": (_I64, [_DESC]),
    "tw_flow_pack_h3": (C.c_int, [_DESC, _P, _P, _P]),
    "tw_flow_packed_h1_bytes": (_I64, [_DESC]),
    "tw_flow_pack_h1": (C.c_int, [_DESC, _P, _P, _P]),
    "tw_flow_path_supported": (C.c_int, [_DESC, _I32, _I32]),
    "tw_flow_workspace_bytes": (_I64, [_DESC, _I64, _I32]),
    "tw_flow_pass": (C.c_int, [_DESC, _P, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _I64, _P]),
    "tw_flow_log_likelihood": (C.c_int, [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _I64, _P]),
    "tw_flow_sample_with_logp": (
        C.c_int,
        [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _I64, _P],
    ),
    "tw_flow_sample_with_logp_multi": (
        C.c_int,
        [_DESC, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _I64, _P],
    ),
    "tw_kernel_scores": (C.c_int, [_P, _P, _P, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "tw_kernel_scores_cheb": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "tw_centre": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P]),
    "tw_kinetic_energy": (C.c_int, [_P, _P, _I32, _F, _P, _I64, _I32, _P]),
    "tw_amber_energy": (C.c_int, [C.POINTER(ForceField), _P, _P, _P, _I64, _P]),
    "tw_amber_energy_forces": (C.c_int, [C.POINTER(ForceField), _P, _P, _P, _I64, _P]),
    "tw_langevin_steps": (C.c_int, [C.POINTER(ForceField), _P, _P, _P, _I32, C.c_double, C.c_double, C.c_double, _I32,
                                    C.c_uint64, _I64, _P, _I64, _P]),
    "tw_mh_accept": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P]),
    "tw_mh_accept_chains": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I32, _P]),
    "tw_chirality_changed": (C.c_int, [_P, _P, _P, _I32, _P, _I64, _I32, _P]),
    "tw_mh_iteration_workspace_bytes": (_I64, [_DESC, _I64, _I32]),
    "tw_mh_iteration": (C.c_int, [_DESC, _P, _P, _I32, C.POINTER(ForceField), C.POINTER(MHOptions), _P, _P, _I32, _P, _P, _P, _P,
                                  _P, _P, _P, _P, _P, _P, _I64, _P, _I64, _P]),
    "tw_mh_iteration_chains_workspace_bytes": (_I64, [_DESC, _I64, _I64, _I32]),
    "tw_mh_iteration_chains": (C.c_int, [_DESC, _P, _P, _I32, C.POINTER(ForceField), C.POINTER(MHOptions), C.POINTER(MHDraws), _P, _P,
                                         _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P, _I64, _P]),
    "tw_mh_draw_chains": (C.c_int, [_DESC, _P, C.POINTER(MHDraws), _P, _P, _P, _P, _I64, _I64, _I32, _P]),
    "tw_flow_nonfinite": (C.c_int, [_I32, C.POINTER(C.c_int32)]),
    "tw_last_netblock_kernel": (C.c_char_p, []),
    "tw_flow_selected_kernel": (C.c_char_p, [_DESC, _I32, _I64, _I32]),
    "tw_debug_set_flags": (C.c_int, [C.c_int]),
    "tw_profile_begin": (C.c_int, []),
    "tw_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "tw_probe_mfma_clock": (C.c_int, [_I32, _I32, C.POINTER(C.c_int64), C.POINTER(C.c_double), _P]),
    "tw_debug_netblock": (
        C.c_int,
        [_DESC, _P, _P, _I32, _I32, _P, _P, _P, _P, _I64, _P, _I64, _I32, _I32, _P, _P, _I64, _P],
    ),
}


def lib_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building with hipcc first if the .so is absent or stale - by content hash of csrc/ - and hipcc is
    available; a fresh library makes build_library() return at once)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB_PATH
    if build_if_missing and (_build.hipcc_path() is not None or not os.path.exists(path)):
        try:
            _build.build_library()
        except (OSError, RuntimeError) as e:
            # a read-only install, or a compiler that is present but unusable: an existing library that passes the ABI
            # check below is still the HIP path - load it and say so, instead of failing on the rebuild
            if not os.path.exists(path):
                raise
            import warnings
            warnings.warn(f"timewarp_amd: could not rebuild {path} ({e}); loading the existing library", RuntimeWarning)
    elif os.path.exists(path) and _build.stale():
        import warnings
        warnings.warn(f"timewarp_amd: {path} was not built from the csrc/ sources next to it (content hash differs or is "
                      "missing) and there is no hipcc here to rebuild it; loading it as it is", RuntimeWarning)
    if not os.path.exists(path):
        raise RuntimeError(
            f"timewarp_amd: {path} is missing. Build it with `python -m timewarp_amd.build` "
            "(needs hipcc); there is no CPU fallback for the HIP path."
        )
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.tw_abi_version() != ABI_VERSION:
        raise RuntimeError("timewarp_amd: ABI version mismatch between _lib.py and libtimewarp_hip.so")
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().tw_last_error().decode(errors="replace")
        raise RuntimeError(f"timewarp_amd: {what} failed ({rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(
            f"timewarp_amd: `{name}` is on {t.device}; the HIP path needs tensors on an MI355X "
            "(device type 'cuda' under ROCm). There is no CPU fallback."
        )
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()
