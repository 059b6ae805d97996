"""ctypes view of libmillion_hip.so (include/million_hip.h).  No fallback: if the library is missing
or fails to load, importing the ops raises — the product path never runs on a CPU substitute."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libmillion_hip.so"
if os.environ.get("MILLION_HIP_LIB"):      # development A/B builds (tools/ab_build.py); the product path uses the in-tree library
    LIB_PATH = Path(os.environ["MILLION_HIP_LIB"]).resolve()

c_i32, c_i64, c_u32, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t

MILLION_CODES_ROWMAJOR, MILLION_CODES_KPAGES, MILLION_CODES_VPAGES = 0, 1, 2
MILLION_KV_ROWMAJOR, MILLION_KV_PAGED = 0, 1


class EncodeDesc(ctypes.Structure):
    _fields_ = [("struct_size", c_u32), ("bs", c_i32), ("nh_k", c_i32), ("n", c_i32),
                ("d", c_i32), ("M", c_i32), ("C", c_i32),
                ("x_stride_b", c_i64), ("x_stride_h", c_i64), ("x_stride_n", c_i64),
                ("x_row_start", c_i32), ("x_row_mod", c_i32),
                ("dst_layout", c_i32), ("dst_token_start", c_i32),
                ("dst_stride_b", c_i64), ("dst_stride_h", c_i64),
                ("page_size", c_i32), ("n_pages_cap", c_i32), ("dev_lengths", c_vp), ("cents_prepared", c_vp)]


class AttnDesc(ctypes.Structure):
    _fields_ = [("struct_size", c_u32), ("bs", c_i32), ("nh", c_i32), ("nh_k", c_i32),
                ("d", c_i32), ("M", c_i32), ("C", c_i32),
                ("n_tokens", c_i32), ("r", c_i32), ("resid_start", c_i32), ("resid_cap", c_i32),
                ("resid_stride_b", c_i64), ("resid_stride_h", c_i64),
                ("k_layout", c_i32), ("v_layout", c_i32), ("page_size", c_i32), ("n_pages_cap", c_i32),
                ("page_ids_i64", c_i32), ("v_pages_dense", c_i32),
                ("k_stride_b", c_i64), ("k_stride_h", c_i64), ("v_stride_b", c_i64), ("v_stride_h", c_i64),
                ("dev_lengths", c_vp), ("k_pool_pages", c_i32), ("v_pool_pages", c_i32)]


class PrefillDesc(ctypes.Structure):
    _fields_ = [("struct_size", c_u32), ("bs", c_i32), ("nh", c_i32), ("nh_k", c_i32), ("d", c_i32),
                ("n_q", c_i32), ("n_kv", c_i32), ("q_pos0", c_i32), ("causal", c_i32),
                ("q_stride_b", c_i64), ("q_stride_h", c_i64), ("q_stride_n", c_i64),
                ("k_stride_b", c_i64), ("k_stride_h", c_i64), ("k_stride_n", c_i64),
                ("v_stride_b", c_i64), ("v_stride_h", c_i64), ("v_stride_n", c_i64),
                ("o_stride_b", c_i64), ("o_stride_h", c_i64), ("o_stride_n", c_i64)]


# every symbol include/million_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "million_version": (c_i32, []),
    "million_last_error": (ctypes.c_char_p, []),
    "million_prepared_cents_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "million_prepare_cents": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "million_pq_encode": (c_i32, [ctypes.POINTER(EncodeDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "million_pq_decode": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "million_pq_flush": (c_i32, [ctypes.POINTER(EncodeDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "million_pq_flush_layers": (c_i32, [ctypes.POINTER(EncodeDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                                        c_i32, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "million_transpose_v_codes": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_vp]),
    "million_attn_workspace_bytes": (c_sz, [ctypes.POINTER(AttnDesc)]),
    "million_workspace_init": (c_i32, [c_vp, c_sz, c_vp]),
    "million_pq_decode_attn": (c_i32, [ctypes.POINTER(AttnDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_vp, c_vp, c_sz, c_vp]),
    "million_pq_decode_attn_append": (c_i32, [ctypes.POINTER(AttnDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                              c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "million_attn_kernel_kind": (c_i32, [ctypes.POINTER(AttnDesc)]),
    "million_prefill_attn": (c_i32, [ctypes.POINTER(PrefillDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "million_set_force_generic": (None, [c_i32]),
    "million_debug_set_stamp_buffer": (None, [c_vp]),
    "million_debug_bad_page_ids": (c_i32, []),
    "million_debug_tail_faults": (c_i32, []),
    "million_debug_rows_reduce": (c_i32, [c_vp, c_vp, c_vp, c_vp]),
    "million_lengths_advance": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp]),
    "million_residual_append": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64,
                                        c_i32, c_i32, c_vp, c_vp]),
}

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: run `make bindings` (hipcc --offload-arch=gfx950). "
                              "There is no CPU fallback for the PQ-KV hot path.")
        # torch first: its bundled HIP runtime must be the one this library binds to.  Loaded the other way round
        # (this library pulling in the system libamdhip64 before torch brings its own) the process ends up with two
        # runtimes and the second one reports "no ROCm-capable device".
        import torch  # noqa: F401
        L = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)      # AttributeError if the ABI drifted
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().million_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
