"""
ctypes binding of libggc_hip.so — the C-ABI declared in include/ggc.h.

This is the only place the Python host touches native code.  There is no CPU
fallback: if the library is missing, or no MI355X is visible, every compute
entry point raises.  Device memory, streams and multi-process plumbing come
from PyTorch-ROCm; tensors are handed to the library as raw device pointers.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

_PKG_DIR = Path(__file__).resolve().parent
_LIB_PATH = _PKG_DIR.parent / "libggc_hip.so"

GGC_OK = 0
ERROR_NAMES = {
    -1: "GGC_E_INVALID_ARG", -2: "GGC_E_SHAPE", -3: "GGC_E_OOM",
    -4: "GGC_E_DEVICE", -5: "GGC_E_UNSUPPORTED", -6: "GGC_E_STATE",
}

_vp, _i, _f, _d, _i64, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_int64, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/ggc.h
SIGNATURES = {
    "ggc_version": [],
    "ggc_ctx_create": [_i, C.POINTER(_vp)],
    "ggc_ctx_destroy": [_vp],
    "ggc_last_error": [_vp],
    "ggc_profile_enable": [_vp, _i],
    "ggc_profile_query": [_vp, C.c_char_p, C.POINTER(_i), C.POINTER(C.c_double)],
    "ggc_debug_read_scratch": [_vp, C.c_char_p, _vp, C.c_size_t],
    "ggc_preprocess": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "ggc_slic": [_vp, _vp, _i, _i, _i, _vp, _i, _f, _f, _i, _vp, _vp],
    "ggc_slic_rgb": [_vp, _vp, _i, _i, _i, _vp, _i, _d, _d, _vp, _vp],
    "ggc_slic_enforce_connectivity": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp],
    "ggc_graph_prior_sigmas": [_vp, _d, _d],
    "ggc_graph_count": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "ggc_graph_fill": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "ggc_guided_filter": [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _f, _vp],
    "ggc_resgcn_configure": [_vp, _i, _i],
    "ggc_resgcn_load_weight": [_vp, C.c_char_p, _vp, _i64],
    "ggc_resgcn_ready": [_vp],
    "ggc_resgcn_forward": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ggc_gcnnet_configure": [_vp, _i, _i],
    "ggc_gcnnet_load_weight": [_vp, C.c_char_p, _vp, _i64],
    "ggc_gcnnet_ready": [_vp],
    "ggc_gcnnet_forward": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ggc_gat_configure": [_vp, _i, _i, _i],
    "ggc_gat_load_weight": [_vp, C.c_char_p, _vp, _i64],
    "ggc_gat_ready": [_vp],
    "ggc_gat_forward": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ggc_gcn_aggregate": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ggc_build_csr": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "ggc_refine_trimap": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _i, _vp],
    "ggc_seed_from_prior": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, C.c_double, _vp],
    "ggc_grabcut": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _u64, _vp],
    "ggc_clean_mask": [_vp, _vp, _i, _i, _i, _vp, _f, _i, _vp],
    "ggc_compose_outputs": [_vp, _vp, _i, _i, _i, _vp, _vp, _f, _i, _i, _i, _vp, _vp],
    "ggc_mask_iou": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "ggc_region_label_stats": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "ggc_eval_counts": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "ggc_convert_color8": [_vp, _vp, _i64, _vp, _i, _vp],
}
_RESTYPES = {"ggc_last_error": C.c_char_p}

_lib: Optional[C.CDLL] = None


class GGCError(RuntimeError):
    """A libggc_hip.so call returned a negative status."""

    def __init__(self, code: int, message: str):
        self.code = code
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")


def library_path() -> Path:
    return Path(os.environ.get("GGC_HIP_LIBRARY", str(_LIB_PATH)))


def load_library() -> C.CDLL:
    """Load libggc_hip.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise RuntimeError(
            f"{path} is missing: build it with `make -C {_PKG_DIR.parent}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "gcn_grabcut has no CPU fallback."
        )
    try:
        import torch  # noqa: F401  (loads libamdhip64.so.7 first; our library binds to the same one)
    except ImportError:
        pass
    lib = C.CDLL(str(path))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def ptr(t) -> Optional[int]:
    """Raw device (or host) address of a tensor / array, or None."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


class Context:
    """One ggc_ctx bound to one GPU (one per process in the multi-GPU layout)."""

    def __init__(self, device_index: int = 0):
        self.lib = load_library()
        handle = _vp()
        rc = self.lib.ggc_ctx_create(int(device_index), C.byref(handle))
        if rc != GGC_OK:
            msg = self.lib.ggc_last_error(None)
            raise GGCError(rc, msg.decode() if msg else "ggc_ctx_create failed")
        self.handle = handle
        self.device_index = int(device_index)
        self.resident: dict = {}       # which model's weights this context holds ("resgcn" / "gcnnet" -> fingerprint)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.ggc_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int) -> None:
        if rc != GGC_OK:
            msg = self.lib.ggc_last_error(self.handle)
            raise GGCError(rc, msg.decode() if msg else "")

    def call(self, name: str, *args) -> None:
        self.check(getattr(self.lib, name)(self.handle, *args))

    def profile_enable(self, on=True) -> None:
        """True / 1: every instrumented scope; 2: only the graded GCNConv aggregation; False / 0: off."""
        self.call("ggc_profile_enable", int(on))

    def profile_query(self, kernel: str) -> tuple[int, float]:
        """(launches, total milliseconds) of one profiled kernel since profile_enable."""
        n, ms = C.c_int(0), C.c_double(0.0)
        self.call("ggc_profile_query", kernel.encode(), C.byref(n), C.byref(ms))
        return n.value, ms.value


_contexts: dict[int, Context] = {}


def get_context(device_index: int = 0) -> Context:
    """Process-wide context cache, one per device index."""
    ctx = _contexts.get(device_index)
    if ctx is None or ctx.handle is None:
        ctx = Context(device_index)
        _contexts[device_index] = ctx
    return ctx


def current_stream(device_index: int = 0) -> int:
    """hipStream_t of torch's current stream on that device, as an integer."""
    import torch
    return int(torch.cuda.current_stream(device_index).cuda_stream)
