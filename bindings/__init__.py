"""`bindings` — drop-in for the reference's CUDA extension module of the same name.

The reference builds `bindings` with `make bindings` (reference makefile:1-4 ->
scripts/modeldb/bindings/setup.py:26-54), which stamps one torch-typed C++ function per
(f16, u8, Ns, Lt=d, d, M, C) tuple (bindings.template.cpp:11-63) and its callers resolve them by name:
`getattr(__import__('bindings'), fname)` (scripts/utils/pq_utils.py:64-65), `import_module("bindings")`
(scripts/modeldb/bindings/test_kernel.py:15).  This module exports the same names with the same
positional signatures; behind ALL of them sits one C ABI (include/million_hip.h) of hand-written
gfx950 kernels in million_amd/libmillion_hip.so.

Exported families
  flash_decoding_allocated_buffer_f16u8_Ns{Ns}Lt{d}d{d}M{M}C{C}                       (10 args, production path)
  flash_decoding_allocated_paged_buffer_* / _paged_split_qkv_buffer_* / _paged_lastblock_sync_buffer_*
                                                                                     (same 10 args; the
      reference's experimental variants, setup.py:28-31 — here aliases of the production function)
  flash_decoding_paged_v_f16u8_Ns{Ns}Lt{Lt}d{d}M{M}C{C}                               (13 args: the call the
      reference's PagedPQCache makes, scripts/utils/paged_pq_utils.py:547,621-635, whose symbol the
      reference never generated)
  pq_encode_f16u8_d{d}M{M}C{C}(X, cents) -> codes                                     (replaces sa_encode_4d_keops)
Ns in {1,2,4,8,16,32} (Ns=1 is what l2Ns returns for l <= 64, pq_utils.py:21-22; the reference only
compiled {2,..,32}).  The split count of the name is NOT the launch geometry: the kernel picks its
own split count from the CU count; partial_out_buffer / partial_lse_buffer are accepted and left
untouched (contents are "unspecified after the call" in the reference too).

Errors: bad dtype/shape raise RuntimeError (the reference raises c10::Error -> RuntimeError from
data_ptr<T>() and calls exit() on launch errors, Interface.template.cu:3-11; this module never exits).
"""
from __future__ import annotations

from itertools import product

import torch

from million_amd import ops as _ops

_NS = (1, 2, 4, 8, 16, 32)
_D = (64, 128)
_M = (16, 32, 64)
_C = (128, 256)
_LT_PAGED = (64, 128, 256)


def _check_common(name, query, key_cents, value_cents, key_residuals, value_residuals, r, d, M, C):
    if query.dim() != 4 or query.shape[2] != 1 or query.shape[3] != d:
        raise RuntimeError(f"{name}: query must be (bs, nh, 1, {d}), got {tuple(query.shape)}")
    for nm, c in (("key_cents", key_cents), ("value_cents", value_cents)):
        if tuple(c.shape) != (M, C, d // M):
            raise RuntimeError(f"{name}: {nm} must be ({M}, {C}, {d // M}), got {tuple(c.shape)}")
        if c.dtype != torch.float16:
            raise RuntimeError(f"{name}: expected scalar type Half for {nm} but found {c.dtype}")
    if query.dtype != torch.float16:
        raise RuntimeError(f"{name}: expected scalar type Half for query but found {query.dtype}")
    if key_residuals.shape[3] != d or key_residuals.shape != value_residuals.shape:
        raise RuntimeError(f"{name}: residuals must be (bs, nh_k, Lt, {d})")
    if not (0 <= int(r) <= key_residuals.shape[2]):
        raise RuntimeError(f"{name}: r={r} outside [0, {key_residuals.shape[2]}]")


def _make_flash_decoding(name, Ns, Lt, d, M, C):
    def flash_decoding(query, key_codes, value_codes, key_cents, value_cents, key_residuals, value_residuals,
                       r, partial_out_buffer, partial_lse_buffer):
        def check():      # run when the call's signature (shapes, strides, dtypes, codebook objects) is new: ops.decode_attn_planned
            _check_common(name, query, key_cents, value_cents, key_residuals, value_residuals, r, d, M, C)
            if key_codes.dtype != torch.uint8 or value_codes.dtype != torch.uint8:
                raise RuntimeError(f"{name}: expected scalar type Byte for codes")
        return _ops.decode_attn_planned(query, key_codes, value_codes, key_cents, value_cents, key_residuals, value_residuals,
                                        r, M=M, C=C, check=check)

    flash_decoding.__name__ = name
    flash_decoding.__doc__ = (f"(query, key_codes, value_codes, key_cents, value_cents, key_residuals, value_residuals, r, "
                              f"partial_out_buffer, partial_lse_buffer) -> Tensor(bs, nh, 1, {d}); "
                              "reference Interface.template.cu:26-38")
    return flash_decoding


def _make_paged_v(name, Ns, Lt, d, M, C):
    def flash_decoding_paged_v(query, key_codes, key_cents, key_residuals, value_page_ids, value_page_pool,
                               value_cents, value_residuals, r, n_pages, page_size, partial_out_buffer,
                               partial_lse_buffer):
        def check():      # see _make_flash_decoding
            _check_common(name, query, key_cents, value_cents, key_residuals, value_residuals, r, d, M, C)
            if value_page_pool.dim() != 3 or value_page_pool.shape[1] != M or value_page_pool.shape[2] != page_size:
                raise RuntimeError(f"{name}: value_page_pool must be (max_pages, {M}, {page_size})")
            if value_page_ids.dim() != 3 or value_page_ids.shape[2] < n_pages:
                raise RuntimeError(f"{name}: value_page_ids must be (bs, nh_k, >= n_pages)")
        if key_codes.shape[2] > n_pages * page_size:
            raise RuntimeError(f"{name}: {key_codes.shape[2]} key tokens but only {n_pages} value pages of {page_size}")
        return _ops.decode_attn_planned(query, key_codes, value_page_pool, key_cents, value_cents, key_residuals,
                                        value_residuals, r, M=M, C=C, v_page_ids=value_page_ids, page_size=int(page_size),
                                        check=check)

    flash_decoding_paged_v.__name__ = name
    flash_decoding_paged_v.__doc__ = ("(query, key_codes, key_cents, key_residuals, value_page_ids, value_page_pool, "
                                      "value_cents, value_residuals, r, n_pages, page_size, partial_out_buffer, "
                                      "partial_lse_buffer) -> Tensor; reference paged_pq_utils.py:621-635")
    return flash_decoding_paged_v


def _make_encode(name, d, M, C):
    def pq_encode(X, cents):
        if tuple(cents.shape) != (M, C, d // M) or X.shape[-1] != d:
            raise RuntimeError(f"{name}: X (bs, nh_k, n, {d}) and cents ({M}, {C}, {d // M}) expected")
        return _ops.pq_encode(X, cents)

    pq_encode.__name__ = name
    pq_encode.__doc__ = "(X (bs,nh_k,n,d) f16, cents (M,C,d/M) f16) -> codes (bs,nh_k,n,M) u8; replaces sa_encode_4d_keops"
    return pq_encode


def _export():
    g = globals()
    names = []
    for Ns, d, M, C in product(_NS, _D, _M, _C):
        Lt = d   # "Best practice: Lt = d", reference setup.py:27
        base = f"f16u8_Ns{Ns}Lt{Lt}d{d}M{M}C{C}"
        fn = _make_flash_decoding(f"flash_decoding_allocated_buffer_{base}", Ns, Lt, d, M, C)
        for fam in ("flash_decoding_allocated_buffer_", "flash_decoding_allocated_paged_buffer_",
                    "flash_decoding_allocated_paged_split_qkv_buffer_",
                    "flash_decoding_allocated_paged_lastblock_sync_buffer_"):
            g[fam + base] = fn
            names.append(fam + base)
        for Ltp in _LT_PAGED:
            nm = f"flash_decoding_paged_v_f16u8_Ns{Ns}Lt{Ltp}d{d}M{M}C{C}"
            g[nm] = _make_paged_v(nm, Ns, Ltp, d, M, C)
            names.append(nm)
    for d, M, C in product(_D, _M, _C):
        nm = f"pq_encode_f16u8_d{d}M{M}C{C}"
        g[nm] = _make_encode(nm, d, M, C)
        names.append(nm)
    return names


__all__ = _export()
