"""ctypes binding of libmerizo_search_amd.so (C ABI: include/merizo_search_amd.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded every product
entry point raises ``MerizoHipError``.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MS_LIB_OVERRIDE") or os.path.join(_PKG, "libmerizo_search_amd.so")
CSRC = os.path.join(_PKG, "csrc")

MODE_IP_PRENORM = 0
MODE_COSINE_RAW = 1
MODE_COSINE_UNIT = 2
MODE_IP_NORMQ = 3
DIM = 128
PREFILTER_MAX_K = 48       # MS_PREFILTER_MAX_K (include/merizo_search_amd.h): the prefiltered search serves k up to this
PREFILTER_MIN_ROWS = 65536


class MerizoHipError(RuntimeError):
    """The gfx950 library is unavailable or one of its entry points failed."""


_lib: Optional[ctypes.CDLL] = None

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

ABI_VERSION = 210
# prefilter image formats (include/merizo_search_amd.h)
PF_BF16X3, PF_F16X2, PF_F16X1 = 0, 1, 2

# name -> (restype, argtypes); exactly the symbols include/merizo_search_amd.h declares
SIGNATURES = {
    "ms_version": (_int, []),
    "ms_last_error": (ctypes.c_char_p, []),
    "ms_device_count": (_int, []),
    "ms_device_cu_count": (_int, []),
    "ms_device_pci_bus_id": (_int, [ctypes.c_char_p, _int]),
    "ms_small_batch_thresholds": (None, [_vp, _vp]),
    "ms_prefilter_max_k": (_int, []),
    "ms_pf_few_min_rows": (_i64, [_int]),
    "ms_l2_normalize_rows": (_int, [_vp, _i64, _int, _f, _vp]),
    "ms_l2_normalize_rows_to": (_int, [_vp, _vp, _i64, _int, _f, _vp]),
    "ms_row_inv_norms": (_int, [_vp, _i64, _int, _f, _vp, _vp]),
    "ms_ip_topk_workspace_bytes": (_sz, [_i64, _int, _int]),
    "ms_ip_topk": (_int, [_vp, _i64, _i64, _vp, _int, _int, _int, _vp, _vp, _vp, _f, _vp, _vp, _vp, _sz, _vp]),
    "ms_ip_topk_prepare": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _vp, _vp, _f, _vp, _sz, _vp]),
    "ms_ip_topk_scan": (_int, [_vp, _i64, _vp, _int, _int, _int, _vp, _vp, _vp, _f, _vp, _sz, _vp]),
    "ms_ip_topk_finish": (_int, [_i64, _i64, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "ms_pf_image_bytes": (_sz, [_i64, _int]),
    "ms_pf_build_image": (_int, [_vp, _i64, _int, _f, _vp, _vp]),
    "ms_pf_err_coef": (_f, [_int]),
    "ms_ip_topk_prefiltered_workspace_bytes": (_sz, [_i64, _int, _int]),
    "ms_ip_topk_prefiltered": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _int, _int, _int, _vp, _vp, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "ms_ip_topk_prefiltered_prepare": (_int, [_vp, _vp, _int, _i64, _vp, _int, _int, _int, _vp, _vp, _f, _f, _vp, _sz, _vp]),
    "ms_ip_topk_prefiltered_scan": (_int, [_vp, _vp, _int, _i64, _vp, _int, _int, _int, _vp, _vp, _f, _f, _vp, _sz, _vp]),
    "ms_ip_topk_prefiltered_finish": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _int, _int, _int, _vp, _vp, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "ms_debug_prefilter_state": (_int, [_vp, _vp, _vp, _vp]),
    "ms_debug_prefilter_poison": (_int, [_vp, ctypes.c_uint, ctypes.c_uint]),
    "ms_debug_prefilter_lists": (_int, [_vp, _i64, _int, _int, _int, _vp, _vp, _vp]),
    "ms_topk_merge": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "ms_topk_merge_strided": (_int, [_vp, _vp, _i64, _i64, _int, _int, _int, _vp, _vp, _vp]),
    "ms_egnn_weight_floats": (_sz, []),
    "ms_egnn_prepared_bytes": (_sz, []),
    "ms_egnn_prepare_weights": (_int, [_vp, _vp, _vp]),
    "ms_egnn_workspace_bytes": (_sz, [_int, _i64, _i64]),
    "ms_egnn_embed": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _int, _vp, _vp, _sz, _vp]),
}


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", CSRC] + (["-B"] if force else [])
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise MerizoHipError("hipcc build failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the library and bind every declared symbol (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MerizoHipError(
            f"{LIB_PATH} not found: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "This package has no CPU fallback.")
    # torch bundles its own HIP runtime: import it FIRST so that this library's libamdhip64
    # dependency resolves to the copy torch already loaded (one runtime per process; loading
    # the system copy first makes torch.cuda report no devices).
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise MerizoHipError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise MerizoHipError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    # the ABI this module binds: 200 = pf_format in the prefilter entry points (an older library would take shifted arguments); 210 = + two entry points
    if lib.ms_version() != ABI_VERSION:
        raise MerizoHipError(f"{LIB_PATH} reports ABI version {lib.ms_version()}, this package binds {ABI_VERSION}: rebuild it (make -C {CSRC})")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ms_last_error()
        raise MerizoHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


_gpu_torch = None


def require_gpu():
    """Return the torch module after verifying a HIP device and the library are present (checked once per process:
    every ops entry point calls this)."""
    global _gpu_torch
    if _gpu_torch is not None:
        return _gpu_torch
    import torch

    lib = load()
    if not torch.cuda.is_available() or lib.ms_device_count() < 1:
        raise MerizoHipError("no MI355X / HIP device visible: the merizo_search_amd hot path runs on the GPU only "
                             "(use the reference implementation for CPU runs)")
    _gpu_torch = torch
    return torch


def ptr(t) -> int:
    """Device pointer of a torch tensor (or None)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
