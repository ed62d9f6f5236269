"""ctypes binding of libvsel.so (C-ABI in include/vsel.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libvsel.so")

VSEL_BF16, VSEL_F32 = 0, 1
STATUS = {0: "VSEL_OK", 1: "VSEL_ERR_INVALID", 2: "VSEL_ERR_WORKSPACE", 3: "VSEL_ERR_HIP", 4: "VSEL_ERR_UNSUPPORTED", 5: "VSEL_ERR_BUSY"}


class VselError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"{STATUS.get(status, status)}: {msg}")
        self.status = status


class Segments(C.Structure):
    _fields_ = [("n_seg", C.c_int64), ("rows_per_seg", C.c_int64), ("total_rows", C.c_int64), ("k", C.c_int64),
                ("total_out", C.c_int64), ("seg_rows", C.c_void_p), ("seg_out", C.c_void_p)]


class Scorer(C.Structure):
    _fields_ = [("wq", C.c_void_p), ("bq", C.c_void_p), ("wk", C.c_void_p), ("bk", C.c_void_p),
                ("d", C.c_int64), ("hd", C.c_int64), ("wdtype", C.c_int)]


# name -> (restype, argtypes); must list every symbol include/vsel.h declares (tests check this)
_P, _I64, _F, _SZ = C.c_void_p, C.c_int64, C.c_float, C.c_size_t
_SEG, _SC = C.POINTER(Segments), C.POINTER(Scorer)
SIGNATURES = {
    "vsel_version": (C.c_char_p, []),
    "vsel_last_error": (C.c_char_p, []),
    "vsel_profile_start": (C.c_int, []),
    "vsel_profile_stop": (C.c_int, [C.c_char_p, _SZ, _P, _P, C.c_int, _P]),
    "vsel_lis_workspace_bytes": (_SZ, [_SEG, _I64, _I64]),
    "vsel_lis_select": (C.c_int, [_P, _P, C.c_int, _SEG, _SC, _P, _SZ, _P, _P, _P]),
    "vsel_lis_select_permuted": (C.c_int, [_P, _P, C.c_int, _SEG, _SC, _P, _SZ, _P, _P, _P, _P, _P]),
    "vsel_lis_scores": (C.c_int, [_P, _P, C.c_int, _SEG, _SC, _P, _SZ, _P]),
    "vsel_topk_select": (C.c_int, [_P, _P, _SEG, _P, _P]),
    "vsel_gather_rows": (C.c_int, [_P, _P, C.c_int, _I64, _SEG, _P, _P]),
    "vsel_soft_topk_fwd": (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _P]),
    "vsel_soft_topk_fwd_bf16ref": (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _P]),
    "vsel_soft_topk_bwd": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P]),
    "vsel_lis_train_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "vsel_lis_train_fwd": (C.c_int, [_P, _P, C.c_int, _I64, _I64, _SC, _P, _SZ, _P, _P, _P, _P, _P, _P]),
    "vsel_lis_train_bwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, _SC, _P, _P, _P, _P, _P, _F, _P, _SZ, _P, _P, _P, _P, _P]),
    "vsel_lis_factors_to_grads": (C.c_int, [_P, _P, _I64, _I64, _I64, _F, _P, _P, _P, _P]),
    "vsel_lis_train_bwd_factors": (C.c_int, [_P, _P, _P, C.c_int, _I64, _SC, _P, _P, _P, _P, _P, _F, _P, _SZ, _P, _P, _P, _P]),
    "vsel_lis_scores_bwd": (C.c_int, [_P, _P, _P, C.c_int, _I64, _SC, _P, _SZ, _P, _P, _P, _P, _P]),
    "vsel_gelu_colsum_workspace_bytes": (_SZ, [_SEG, _I64]),
    "vsel_gelu_colsum": (C.c_int, [_P, _P, C.c_int, _SEG, _I64, _P, _P, _P, _SZ]),
    "vsel_colsum_linear_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "vsel_colsum_linear": (C.c_int, [_P, _P, _SEG, _P, _P, C.c_int, _I64, _I64, _P, _P, _SZ]),
    "vsel_lis_select_presummed": (C.c_int, [_P, _P, C.c_int, _SEG, _SC, _P, _P, _SZ, _P, _P, _P, _P, _P]),
    "vsel_splice": (C.c_int, [_P, _P, _I64, _I64, _P, _I64, _I64, _P, _P, C.c_int, _I64, _P, _I64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vsel_splice_batched": (C.c_int, [_P, _P, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, C.c_int, _I64, _P, _I64,
                                      _P, _P, _P, _P, _P, _P, _P]),
    "vsel_lis_select_splice": (C.c_int, [_P, _P, C.c_int, _SEG, _SC, _P, _SZ, _P, _P, _P, _P, _I64, _P, _I64, _I64, _P, _P, _I64, _P,
                                         _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vsel_topk_select_splice": (C.c_int, [_P, _P, C.c_int, _I64, _SEG, _P, _P, _P, _I64, _P, _I64, _I64, _P, _P, _I64, _P, _P, _P,
                                          _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vsel_varlen_attn_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, C.c_int, _P]),
    "vsel_varlen_attn_fwd_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I64, _I64, _I64, C.c_int]),
    "vsel_varlen_attn_fwd_ws": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, C.c_int, _P, _P, _P, _SZ]),
    "vsel_varlen_attn_fwd_lse": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, C.c_int, _P, _P]),
    "vsel_varlen_attn_bwd_workspace_bytes": (_SZ, [_I64, _I64, _I64, _I64, _I64]),
    "vsel_varlen_attn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, C.c_int, _P, _SZ,
                                       _P, _P, _P]),
    "vsel_varlen_attn_fwd_kv": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _F, C.c_int, _P]),
    "vsel_varlen_attn_fwd_strided": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64,
                                               _I64, _F, C.c_int, _P]),
    "vsel_paged_attn_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, C.c_int, _P]),
}

# include/vsel_debug.h (diagnostic knobs; not part of the drop-in boundary)
DEBUG_SIGNATURES = {
    "vsel_debug_set": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "vsel_debug_get": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "vsel_debug_reset": (None, []),
}
KNOBS = {"lis_pipeline": 0, "lis_small_path": 1, "lis_fused_select": 2, "attn_use_tr": 3, "attn_waves": 4, "attn_pack": 5,
         "attn_split": 6, "attn_split_q64": 7, "attn_bwd_split": 8, "lis_splice_fused": 9, "attn_bwd_waves": 10, "attn_tail_first": 11, "lis_seg_sums": 12,
         "attn_xcd_queue": 13, "attn_rows64": 14, "attn_bwd_dq64": 15, "attn_bwd_dkdv64": 16, "attn_static": 17, "attn_skip_empty": 18, "attn_gqa": 19, "attn_gqa_form": 20,
         "attn_bwd_updown": 21, "attn_key_parts": 22, "lis_gather": 23, "train_fused": 24}

_lib = None


def lib() -> C.CDLL:
    """Load libvsel.so once.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        # torch first: its bundled HIP runtime must be the one in the process (loading libvsel.so before torch
        # pulls in /opt/rocm's libamdhip64 instead, which then sees no device under the torch wheel).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m visionselector_amd.build` "
                "(hipcc --offload-arch=gfx950).  visionselector_amd has no CPU / eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status: int) -> None:
    if status != 0:
        raise VselError(status, lib().vsel_last_error().decode())


class debug_knob:
    """with debug_knob("attn_waves", 8): ...   -- force a kernel form for the duration of the block (include/vsel_debug.h);
    the previous value is restored on exit, also when the block raises.  Several knobs: debug_knob(a=1, b=2)."""

    def __init__(self, name: str = None, value: int = None, **more):
        self.items = ([(name, value)] if name is not None else []) + list(more.items())
        for n, _ in self.items:
            if n not in KNOBS:
                raise KeyError(f"unknown knob {n!r}; known: {sorted(KNOBS)}")
        self.prev = []

    def __enter__(self):
        for n, v in self.items:
            old = C.c_int(0)
            check(lib().vsel_debug_set(KNOBS[n], int(v), C.byref(old)))
            self.prev.append((n, old.value))
        return self

    def __exit__(self, *exc):
        for n, v in reversed(self.prev):
            lib().vsel_debug_set(KNOBS[n], v, None)
        self.prev = []
        return False


def debug_get(name: str) -> int:
    v = C.c_int(0)
    check(lib().vsel_debug_get(KNOBS[name], C.byref(v)))
    return v.value


def profile_start() -> None:
    check(lib().vsel_profile_start())


def profile_stop() -> dict:
    """-> {kernel_name: (total_ms, launches)} for everything launched since profile_start()."""
    names = C.create_string_buffer(4096)
    ms = (C.c_float * 64)()
    calls = (C.c_int64 * 64)()
    n = C.c_int(0)
    check(lib().vsel_profile_stop(names, 4096, C.cast(ms, C.c_void_p), C.cast(calls, C.c_void_p), 64, C.cast(C.byref(n), C.c_void_p)))
    keys = names.value.decode().split(",") if n.value else []
    return {k: (float(ms[i]), int(calls[i])) for i, k in enumerate(keys)}
