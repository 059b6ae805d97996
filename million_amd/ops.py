"""Torch-facing wrappers over the C ABI (include/million_hip.h).  PyTorch is plumbing here: device
memory, the current HIP stream and nothing else — all arithmetic runs in libmillion_hip.so.

There is deliberately no CPU/eager fallback: a CPU tensor, a missing library or an unsupported shape
raises.
"""
from __future__ import annotations

import collections
import ctypes
import weakref
from typing import Optional

import torch

from . import _lib as L

_ws_cache = {}
_prep_cache = {}


def _stream() -> int:
    # the raw handle of torch's current stream: ~0.2 us (torch.cuda.current_stream().cuda_stream builds a Stream object: ~1.5 us)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("million_amd ops need device tensors (no CPU fallback)")


def prepare_cents(cents: torch.Tensor, cache: bool = True) -> torch.Tensor:
    """(M, C, d_m) fp16 codebook -> LDS-ready images (row image + column image), see million_prepare_cents.

    Cached per (storage, version): the reference passes the same codebook tensor on every call
    (pq_utils.py:149-159, one table for all layers and heads)."""
    _need_cuda(cents)
    if cents.dtype != torch.float16 or cents.dim() != 3:
        raise RuntimeError(f"codebook must be fp16 (M, C, d_m), got {cents.dtype} {tuple(cents.shape)}")
    # Cache keyed on the tensor OBJECT (weak reference) + its version counter: a data_ptr key would hand a
    # stale table to a new codebook that happens to be allocated where a freed one lived.
    key = id(cents)
    if cache:
        hit = _prep_cache.get(key)
        if hit is not None and hit[0]() is cents and hit[1] == cents._version:
            return hit[2]
    lib = L.load()
    c = cents.contiguous()
    M, C, dm = c.shape
    out = torch.empty(lib.million_prepared_cents_bytes(M, C, dm) // 2, dtype=torch.float16, device=c.device)
    L.check(lib.million_prepare_cents(c.data_ptr(), M, C, dm, out.data_ptr(), _stream()), "million_prepare_cents")
    if cache:
        for k in [k for k, v in _prep_cache.items() if v[0]() is None]:
            del _prep_cache[k]
        _prep_cache[key] = (weakref.ref(cents), cents._version, out)
    return out


def pq_encode_into(X: torch.Tensor, cents: torch.Tensor, dst: torch.Tensor, *, layout: int = L.MILLION_CODES_ROWMAJOR,
                   token_start: int = 0, n: Optional[int] = None, page_ids: Optional[torch.Tensor] = None,
                   page_size: int = 0, x_row_start: int = 0, x_row_mod: int = 0,
                   dev_lengths: Optional[torch.Tensor] = None, use_prepared: bool = True,
                   prepared: Optional[torch.Tensor] = None) -> None:
    """Encode rows of X (bs, nh_k, n_rows, d) fp16 and write the codes into `dst` in their final layout.
    use_prepared: hand the kernel the (cached) prepared codebook, whose fp32 image makes the distance loop ~1.6x
    faster; the codes are bit-identical either way.  prepared: that blob, when the caller already holds it (the caches
    do: nothing is allocated or launched on their behalf inside a captured step)."""
    _need_cuda(X, cents, dst, page_ids)
    code_dtype = code_dtype_for(cents.shape[1])
    if X.dtype != torch.float16 or cents.dtype != torch.float16 or dst.dtype != code_dtype:
        raise RuntimeError(f"pq_encode: X and cents must be fp16, dst {code_dtype} for C={cents.shape[1]}")
    if X.dim() != 4 or X.stride(3) != 1:
        raise RuntimeError("pq_encode: X must be (bs, nh_k, n, d) with a contiguous last dim")
    cents = cents.contiguous()
    bs, nhk, n_rows, d = X.shape
    M, C, dm = cents.shape
    n = n_rows if n is None else n
    desc = L.EncodeDesc()
    desc.struct_size = ctypes.sizeof(L.EncodeDesc)
    desc.bs, desc.nh_k, desc.n, desc.d, desc.M, desc.C = bs, nhk, n, d, M, C
    desc.x_stride_b, desc.x_stride_h, desc.x_stride_n = X.stride(0), X.stride(1), X.stride(2)
    desc.x_row_start, desc.x_row_mod = x_row_start, x_row_mod
    desc.dst_layout, desc.dst_token_start = layout, token_start
    if layout == L.MILLION_CODES_ROWMAJOR:
        if dst.dim() != 4 or dst.stride(3) != 1 or dst.stride(2) != M or dst.shape[3] != M:
            raise RuntimeError("pq_encode: row-major dst must be (bs, nh_k, T_cap, M) with dense rows")
        if dst.shape[2] < token_start + n:
            raise RuntimeError("pq_encode: dst too short")
        desc.dst_stride_b, desc.dst_stride_h = dst.stride(0) * dst.element_size(), dst.stride(1) * dst.element_size()
    else:
        if page_ids is None or page_ids.dtype != torch.int32 or not page_ids.is_contiguous():
            raise RuntimeError("pq_encode: paged dst needs contiguous int32 page_ids (bs, nh_k, n_pages_cap)")
        if not dst.is_contiguous():
            raise RuntimeError("pq_encode: page pool must be contiguous")
        desc.page_size, desc.n_pages_cap = page_size, page_ids.shape[2]
    desc.dev_lengths = _ptr(dev_lengths)
    if prepared is None and use_prepared and C <= 256:
        prepared = prepare_cents(cents)            # cached per codebook tensor; callers that own one pass it in
    desc.cents_prepared = _ptr(prepared)
    lib = L.load()
    L.check(lib.million_pq_encode(ctypes.byref(desc), X.data_ptr(), cents.data_ptr(), dst.data_ptr(),
                                  _ptr(page_ids), _stream()), "million_pq_encode")
    _vshadow_drop(dst)


def pq_decode(codes: torch.Tensor, cents: torch.Tensor) -> torch.Tensor:
    """Drop-in for sa_decode_4d (reference pq_utils.py:501-540): codes (..., M) u8 (u16 for C > 256) + (M, C, d_m) fp16
    codebook -> (..., d) fp16 reconstruction (exact gather)."""
    _need_cuda(codes, cents)
    if cents.dtype != torch.float16 or cents.dim() != 3:
        raise RuntimeError("pq_decode: cents must be fp16 (M, C, d_m)")
    M, C, dm = cents.shape
    if codes.dtype != code_dtype_for(C):
        raise RuntimeError(f"pq_decode: codes must be {code_dtype_for(C)} for C={C}, got {codes.dtype}")
    if codes.shape[-1] != M:
        raise RuntimeError(f"pq_decode: codes have {codes.shape[-1]} subspaces, codebook {M}")
    codes, cents = codes.contiguous(), cents.contiguous()
    out = torch.empty(*codes.shape[:-1], M * dm, dtype=torch.float16, device=codes.device)
    n_rows = codes.numel() // M
    L.check(L.load().million_pq_decode(codes.data_ptr(), cents.data_ptr(), out.data_ptr(), n_rows, M * dm, M, C,
                                       _stream()), "million_pq_decode")
    return out


def code_dtype_for(C: int):
    """nbits2dtype of the reference (pq_utils.py:542-552) in terms of the codebook size: uint8 up to 256 centroids,
    uint16 up to 65536."""
    if C <= 256:
        return torch.uint8
    if C <= 65536:
        return torch.uint16
    raise RuntimeError(f"codebooks of {C} centroids are not supported (<= 65536)")


def pq_encode(X: torch.Tensor, cents: torch.Tensor) -> torch.Tensor:
    """Drop-in for sa_encode_4d_keops (reference pq_utils.py:451-499): (bs, nh_k, n, d) -> (bs, nh_k, n, M) u8 (u16 for
    nbits 9..16, as sa_encode_4d_keops(target_dtype=nbits2dtype(nbits)) returns)."""
    bs, nhk, n, d = X.shape
    codes = torch.empty(bs, nhk, n, cents.shape[0], dtype=code_dtype_for(cents.shape[1]), device=X.device)
    if n:
        pq_encode_into(X, cents, codes)
    return codes


def pq_flush(k_rows: torch.Tensor, v_rows: torch.Tensor, k_cents: torch.Tensor, v_cents: torch.Tensor,
             k_pool: torch.Tensor, v_pool: torch.Tensor, page_ids: torch.Tensor, *, n: int, page_size: int,
             token_start: int = 0, x_row_start: int = 0, dev_lengths: Optional[torch.Tensor] = None, min_r: int = 0,
             advance: bool = True) -> None:
    """One launch per window flush (reference PagedPQCache.flush_to_pages, paged_pq_utils.py:130-210): encode the oldest n
    rows of the K and V windows (rings of k_rows.shape[2] rows) into a K page and a transposed V page; with dev_lengths
    the destination token and the ring start are read on the device and advanced there; min_r > 0 then skips the batch
    items whose window holds fewer rows (ragged batches: only the full windows flush).

    Several layers in one launch: k_rows / v_rows (layers, bs, nh_k, cap, d), page_ids (layers, bs, nh_k, n_pages_cap),
    dev_lengths (layers, bs, 4) - the layers of one cache allocated side by side.  advance=False: encode only (encode-ahead;
    the window moves later: lengths_advance)."""
    _need_cuda(k_rows, v_rows, k_cents, v_cents, k_pool, v_pool, page_ids, dev_lengths)
    n_layers, x_ls, ids_ls, len_ls = 1, 0, 0, 0
    if k_rows.dim() == 5:
        if page_ids.dim() != 4 or page_ids.shape[0] != k_rows.shape[0] or (dev_lengths is not None and (dev_lengths.dim() != 3 or dev_lengths.shape[0] != k_rows.shape[0])):
            raise RuntimeError("pq_flush: layered windows need layered page_ids (and dev_lengths)")
        if k_rows.stride(0) != v_rows.stride(0) or not page_ids.is_contiguous() or (dev_lengths is not None and not dev_lengths.is_contiguous()):
            raise RuntimeError("pq_flush: layered windows must share the layer stride; contiguous page_ids / dev_lengths")
        n_layers, x_ls, ids_ls = k_rows.shape[0], k_rows.stride(0), page_ids.stride(0)
        len_ls = dev_lengths.stride(0) if dev_lengths is not None else 0
        k_rows, v_rows, page_ids = k_rows[0], v_rows[0], page_ids[0]
    if k_rows.shape != v_rows.shape or k_rows.stride() != v_rows.stride() or k_rows.dtype != torch.float16 or v_rows.dtype != torch.float16:
        raise RuntimeError("pq_flush: K and V windows must be fp16 with the same shape and strides")
    if k_rows.dim() != 4 or k_rows.stride(3) != 1:
        raise RuntimeError("pq_flush: windows must be (bs, nh_k, cap, d) with a contiguous last dim")
    if page_ids.dtype != torch.int32 or not page_ids.is_contiguous() or not k_pool.is_contiguous() or not v_pool.is_contiguous():
        raise RuntimeError("pq_flush: contiguous pools and contiguous int32 page_ids (bs, nh_k, n_pages_cap) expected")
    k_cents, v_cents = k_cents.contiguous(), v_cents.contiguous()
    bs, nhk, cap, d = k_rows.shape
    M, C, dm = k_cents.shape
    if v_cents.shape != k_cents.shape or C > 256 or k_pool.dtype != torch.uint8 or v_pool.dtype != torch.uint8:
        raise RuntimeError("pq_flush: uint8 codes, equal codebook shapes")
    desc = L.EncodeDesc()
    desc.struct_size = ctypes.sizeof(L.EncodeDesc)
    desc.bs, desc.nh_k, desc.n, desc.d, desc.M, desc.C = bs, nhk, n, d, M, C
    desc.x_stride_b, desc.x_stride_h, desc.x_stride_n = k_rows.stride(0), k_rows.stride(1), k_rows.stride(2)
    desc.x_row_start, desc.x_row_mod = x_row_start, cap
    desc.dst_layout, desc.dst_token_start = L.MILLION_CODES_KPAGES, token_start
    desc.page_size, desc.n_pages_cap = page_size, page_ids.shape[2]
    desc.dev_lengths = _ptr(dev_lengths)
    L.check(L.load().million_pq_flush_layers(ctypes.byref(desc), k_rows.data_ptr(), v_rows.data_ptr(), k_cents.data_ptr(),
                                             v_cents.data_ptr(), k_pool.data_ptr(), v_pool.data_ptr(), page_ids.data_ptr(),
                                             _ptr(dev_lengths), cap, min_r if dev_lengths is not None else 0,
                                             n_layers, x_ls, ids_ls, len_ls, int(bool(advance)), _stream()), "million_pq_flush_layers")


def transpose_v_codes(v_codes: torch.Tensor, n_tokens: Optional[int] = None) -> torch.Tensor:
    """Row-major V codes (bs, nh_k, T, M) u8 -> dense transposed 64-token pages ((bs*nh_k)*ceil(T/64), M, 64)."""
    _need_cuda(v_codes)
    if v_codes.dtype != torch.uint8 or v_codes.dim() != 4:
        raise RuntimeError("transpose_v_codes: (bs, nh_k, T, M) uint8 expected")
    bs, nhk, T, M = v_codes.shape
    T = T if n_tokens is None else n_tokens
    if T <= 0 or v_codes.stride(3) != 1 or v_codes.stride(2) != M:
        raise RuntimeError("transpose_v_codes: dense rows and T > 0 expected")
    pages = torch.empty(bs * nhk * ((T + 63) // 64), M, 64, dtype=torch.uint8, device=v_codes.device)
    L.check(L.load().million_transpose_v_codes(v_codes.data_ptr(), pages.data_ptr(), bs, nhk, T, M, v_codes.stride(0),
                                               v_codes.stride(1), _stream()), "million_transpose_v_codes")
    return pages


# Transposed shadow of row-major V code tensors.  The reference's production call (Interface.template.cu:26-38) hands the
# SAME value_codes tensor to the kernel on every decode step between two flushes (DynamicPQCache.value_cache[layer] is
# replaced by torch.cat only when the window is flushed, pq_utils.py:140-147).  The fast kernels want transposed pages, so
# the first call on a tensor transposes it once and later calls reuse the pages.  Keyed on the tensor OBJECT (weak
# reference, so a new tensor at a recycled address never hits) + its version counter (in-place writes through torch
# invalidate) + geometry; bounded (least recently used) so abandoned tensors do not pin memory.
_vshadow: "dict[int, tuple]" = {}
_VSHADOW_MAX = 128


def _base_of(t: torch.Tensor) -> torch.Tensor:
    b = t._base
    return t if b is None else b


def _v_pages_of(v_codes: torch.Tensor, n_tokens: int) -> torch.Tensor:
    """Keyed on the BASE tensor of v_codes (weak reference: a new tensor at a recycled address never hits) + the shared
    version counter + the view's geometry: `store[:, :, :T]` taken afresh on every call (what DynamicPQCache.decoding(
    fused=False) and the reference's PagedPQCache pass) finds the pages made for the previous view of the same store."""
    base = _base_of(v_codes)
    key = id(base)
    sig = (v_codes._version, n_tokens, v_codes.shape, v_codes.stride(), v_codes.data_ptr())
    hit = _vshadow.get(key)
    if hit is not None and hit[0]() is base and hit[1] == sig:
        return hit[2]
    pages = transpose_v_codes(v_codes, n_tokens)
    for k in [k for k, v in _vshadow.items() if v[0]() is None]:
        del _vshadow[k]
    while len(_vshadow) >= _VSHADOW_MAX:
        del _vshadow[next(iter(_vshadow))]
    _vshadow.pop(key, None)
    _vshadow[key] = (weakref.ref(base), sig, pages)
    return pages


def _vshadow_drop(dst: torch.Tensor) -> None:
    """dst was written behind torch's back (a kernel of this library: no version bump): forget a transposed shadow of it."""
    _vshadow.pop(id(_base_of(dst)), None)


def attn_workspace(desc: L.AttnDesc, device: torch.device) -> torch.Tensor:
    lib = L.load()
    need = lib.million_attn_workspace_bytes(ctypes.byref(desc))
    # one workspace per (device, stream) serves every shape (its head does not move with the shape, million_hip.h);
    # shapes with more than 2048 (b, kv head) pairs get one of their own
    pairs = desc.bs * desc.nh_k
    key = (device.index, torch.cuda.current_stream().cuda_stream, pairs if pairs > 2048 else 0)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        # zeroed once; every call leaves it ready.  Grown with headroom: the row-major path keeps a
        # transposed copy of the V codes here, which grows with the context.
        ws = torch.zeros(need + need // 2, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def make_attn_desc(q, k_res, *, nh_k, M, C, n_tokens, r, resid_start=0, k_paged=False, v_paged=False,
                   page_size=0, n_pages_cap=0, page_ids_i64=False, k_codes=None, v_codes=None,
                   dev_lengths=None, v_pages_dense=False) -> L.AttnDesc:
    bs, nh, _, d = q.shape
    desc = L.AttnDesc()
    desc.struct_size = ctypes.sizeof(L.AttnDesc)
    desc.bs, desc.nh, desc.nh_k, desc.d, desc.M, desc.C = bs, nh, nh_k, d, M, C
    desc.n_tokens, desc.r, desc.resid_start, desc.resid_cap = n_tokens, r, resid_start, k_res.shape[2]
    desc.resid_stride_b, desc.resid_stride_h = k_res.stride(0), k_res.stride(1)
    desc.k_layout = L.MILLION_KV_PAGED if k_paged else L.MILLION_KV_ROWMAJOR
    desc.v_layout = L.MILLION_KV_PAGED if v_paged else L.MILLION_KV_ROWMAJOR
    desc.page_size, desc.n_pages_cap, desc.page_ids_i64 = page_size, n_pages_cap, int(page_ids_i64)
    desc.v_pages_dense = int(v_pages_dense)
    if not k_paged and k_codes is not None:
        desc.k_stride_b, desc.k_stride_h = k_codes.stride(0), k_codes.stride(1)
    if not v_paged and v_codes is not None:
        desc.v_stride_b, desc.v_stride_h = v_codes.stride(0), v_codes.stride(1)
    desc.dev_lengths = _ptr(dev_lengths)
    # pool sizes: read only by the MILLION_DEBUG_CHECK_IDS build of the library (page ids are trusted otherwise)
    if k_paged and k_codes is not None:
        desc.k_pool_pages = k_codes.shape[0]
    if v_paged and not v_pages_dense and v_codes is not None:
        desc.v_pool_pages = v_codes.shape[0]
    return desc


def _check_rowmajor(name, codes, M, n_tokens):
    if codes.dim() != 4 or codes.shape[3] != M:
        raise RuntimeError(f"pq_decode_attn: {name} must be (bs, nh_k, T, M)")
    if codes.shape[2] and (codes.stride(3) != 1 or codes.stride(2) != M):
        raise RuntimeError(f"pq_decode_attn: {name} rows must be dense (stride M)")
    if n_tokens > codes.shape[2]:
        raise RuntimeError(f"pq_decode_attn: n_tokens exceeds {name}")


def pq_decode_attn(q: torch.Tensor, k_codes: torch.Tensor, v_codes: torch.Tensor, k_prep: torch.Tensor,
                   v_prep: torch.Tensor, k_res: torch.Tensor, v_res: torch.Tensor, r: int, *, M: int, C: int,
                   n_tokens: Optional[int] = None, resid_start: int = 0,
                   k_page_ids: Optional[torch.Tensor] = None, v_page_ids: Optional[torch.Tensor] = None,
                   page_size: int = 0, out: Optional[torch.Tensor] = None,
                   dev_lengths: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                   k_new: Optional[torch.Tensor] = None, v_new: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One fused launch: score/softmax/value-reconstruct over the code store + residual window + merge.

    A side is row-major when its page ids are None: codes (bs, nh_k, T_cap, M) u8 — the reference's
    10-arg call (Interface.template.cu:26-38).  Paged: K pool (n_pool, page_size, M), V pool
    (n_pool, M, page_size), page ids (bs, nh_k, n_pages_cap) int32/int64 (paged_pq_utils.py:621-635 passes
    row-major K with paged V).
    """
    _need_cuda(q, k_codes, v_codes, k_prep, v_prep, k_res, v_res, k_page_ids, v_page_ids, out, dev_lengths)
    if q.dtype != torch.float16 or k_res.dtype != torch.float16 or v_res.dtype != torch.float16:
        raise RuntimeError("pq_decode_attn: q and residuals must be fp16")
    if k_codes.dtype != torch.uint8 or v_codes.dtype != torch.uint8:
        raise RuntimeError("pq_decode_attn: codes must be uint8")
    if q.dim() != 4 or q.shape[2] != 1 or not q.is_contiguous():
        raise RuntimeError("pq_decode_attn: q must be contiguous (bs, nh, 1, d)")
    if (k_res.shape != v_res.shape or k_res.stride() != v_res.stride() or k_res.stride(3) != 1
            or k_res.stride(2) != k_res.shape[3]):
        raise RuntimeError("pq_decode_attn: residual buffers must share shape/strides with dense rows")
    nh_k = k_res.shape[1]
    k_paged, v_paged = k_page_ids is not None, v_page_ids is not None
    n_pages_cap, ids64 = 0, False
    for ids in (k_page_ids, v_page_ids):
        if ids is None:
            continue
        if ids.dtype not in (torch.int32, torch.int64) or not ids.is_contiguous() or ids.dim() != 3:
            raise RuntimeError("pq_decode_attn: page ids must be contiguous int32/int64 (bs, nh_k, n_pages)")
        if n_pages_cap and (ids.shape[2] != n_pages_cap or (ids.dtype == torch.int64) != ids64):
            raise RuntimeError("pq_decode_attn: K and V page ids must agree in dtype and length")
        n_pages_cap, ids64 = ids.shape[2], ids.dtype == torch.int64
    if n_tokens is None:
        if k_paged and v_paged:
            raise RuntimeError("pq_decode_attn: fully paged mode needs n_tokens")
        n_tokens = (v_codes if k_paged else k_codes).shape[2]
    if k_paged:
        if not k_codes.is_contiguous():
            raise RuntimeError("pq_decode_attn: K page pool must be contiguous")
    else:
        _check_rowmajor("key_codes", k_codes, M, n_tokens)
    if v_paged:
        if not v_codes.is_contiguous():
            raise RuntimeError("pq_decode_attn: V page pool must be contiguous")
    else:
        _check_rowmajor("value_codes", v_codes, M, n_tokens)
    v_dense = False
    if (not v_paged and not k_paged and n_tokens > 0 and C in (128, 256) and q.shape[3] in (64, 128) and M in (16, 32, 64)
            and dev_lengths is None):
        # the reference's 10-argument layout on the MFMA shapes (streaming and tile kernels: the whole binding surface):
        # transposed pages of V, made once per code tensor
        v_codes = _v_pages_of(v_codes, n_tokens)
        v_paged, v_dense, page_size, n_pages_cap = True, True, 64, (n_tokens + 63) // 64
    desc = make_attn_desc(q, k_res, nh_k=nh_k, M=M, C=C, n_tokens=n_tokens, r=r, resid_start=resid_start,
                          k_paged=k_paged, v_paged=v_paged, page_size=page_size, n_pages_cap=n_pages_cap,
                          page_ids_i64=ids64, k_codes=k_codes, v_codes=None if v_dense else v_codes,
                          dev_lengths=dev_lengths, v_pages_dense=v_dense)
    if out is None:
        out = torch.empty_like(q)
    ws = workspace if workspace is not None else attn_workspace(desc, q.device)
    lib = L.load()
    if k_new is not None:
        # fused residual-window append: r = valid rows BEFORE the call; the new row becomes row r
        _need_cuda(k_new, v_new)
        bs_, nh_ = q.shape[0], q.shape[1]
        if k_new.shape != (bs_, nh_k, 1, q.shape[3]) or v_new is None or v_new.shape != k_new.shape:
            raise RuntimeError("pq_decode_attn: k_new / v_new must be (bs, nh_k, 1, d)")
        if k_new.dtype != torch.float16 or v_new.dtype != torch.float16:
            raise RuntimeError("pq_decode_attn: k_new / v_new must be fp16")
        k_new, v_new = k_new.contiguous(), v_new.contiguous()
        L.check(lib.million_pq_decode_attn_append(ctypes.byref(desc), q.data_ptr(), k_new.data_ptr(), v_new.data_ptr(),
                                                  _ptr(k_codes), _ptr(v_codes), _ptr(k_page_ids), _ptr(v_page_ids),
                                                  k_prep.data_ptr(), v_prep.data_ptr(), k_res.data_ptr(),
                                                  v_res.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  _stream()), "million_pq_decode_attn_append")
        return out
    L.check(lib.million_pq_decode_attn(ctypes.byref(desc), q.data_ptr(), _ptr(k_codes), _ptr(v_codes),
                                       _ptr(k_page_ids), _ptr(v_page_ids),
                                       k_prep.data_ptr(), v_prep.data_ptr(), k_res.data_ptr(), v_res.data_ptr(),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
            "million_pq_decode_attn")
    return out


def prefill_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, causal: bool = True, q_pos0: int = 0,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Prompt attention on fp16 K/V (million_prefill_attn): q (bs, nh, n_q, d), k / v (bs, nh_k, n_kv, d) -> (bs, nh, n_q, d).
    Replaces scaled_dot_product_attention(q, repeat_kv(k), repeat_kv(v), is_causal=True) of the reference's prompt pass
    (pq_utils.py:249-260) without materialising repeat_kv.  Row strides are free (transposed (bs, n, h, d) projections are
    fine), the d elements of a row must be contiguous."""
    _need_cuda(q, k, v, out)
    if q.dtype != torch.float16 or k.dtype != torch.float16 or v.dtype != torch.float16:
        raise RuntimeError("prefill_attn: q, k, v must be fp16")
    if q.dim() != 4 or k.dim() != 4 or k.shape != v.shape or q.shape[0] != k.shape[0] or q.shape[3] != k.shape[3]:
        raise RuntimeError(f"prefill_attn: q (bs, nh, n_q, d) and k, v (bs, nh_k, n_kv, d) expected, got {tuple(q.shape)} {tuple(k.shape)} {tuple(v.shape)}")
    bs, nh, n_q, d = q.shape
    nh_k, n_kv = k.shape[1], k.shape[2]
    if nh % nh_k:
        raise RuntimeError(f"prefill_attn: nh={nh} is not a multiple of nh_k={nh_k}")
    fix = lambda t: t if (t.stride(3) == 1 and all(s_ % 8 == 0 for s_ in t.stride()[:3])) else t.contiguous()
    q, k, v = fix(q), fix(k), fix(v)
    if out is None:
        # token-major memory under the (bs, nh, n_q, d) shape: the o_proj of a prompt pass wants (bs, n, nh * d), and
        # out.transpose(1, 2).reshape(bs, n, nh * d) is then a view instead of a copy of the whole output
        out = torch.empty(bs, n_q, nh, d, dtype=torch.float16, device=q.device).transpose(1, 2)
    elif out.shape != (bs, nh, n_q, d) or out.dtype != torch.float16 or out.stride(3) != 1:
        raise RuntimeError("prefill_attn: out must be fp16 (bs, nh, n_q, d) with contiguous rows")
    desc = L.PrefillDesc()
    desc.struct_size = ctypes.sizeof(L.PrefillDesc)
    desc.bs, desc.nh, desc.nh_k, desc.d = bs, nh, nh_k, d
    desc.n_q, desc.n_kv, desc.q_pos0, desc.causal = n_q, n_kv, q_pos0, int(causal)
    desc.q_stride_b, desc.q_stride_h, desc.q_stride_n = q.stride(0), q.stride(1), q.stride(2)
    desc.k_stride_b, desc.k_stride_h, desc.k_stride_n = k.stride(0), k.stride(1), k.stride(2)
    desc.v_stride_b, desc.v_stride_h, desc.v_stride_n = v.stride(0), v.stride(1), v.stride(2)
    desc.o_stride_b, desc.o_stride_h, desc.o_stride_n = out.stride(0), out.stride(1), out.stride(2)
    L.check(L.load().million_prefill_attn(ctypes.byref(desc), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _stream()),
            "million_prefill_attn")
    return out


# ---- the reference's 10- / 13-argument calls, steady state ----------------------------------------------------------
# An unchanged reference harness calls `bindings.*` eagerly, 32 times per token (pq_utils.py:61-94); its boundary is a
# compiled pybind11 module (bindings.template.cpp:11-63).  pq_decode_attn above validates everything and rebuilds the
# descriptor on every call (~21 us of Python per call for a ~8 us kernel).  decode_attn_planned keeps, per call signature
# (shapes, strides, dtypes, device), the validated descriptor and workspace of the first call: a steady-state call is a
# signature tuple, a dict lookup, two descriptor fields, the data pointers and ONE ctypes call.
class _Plan:
    __slots__ = ("desc", "desc_ref", "ws", "ws_ptr", "ws_bytes", "v_dense", "kc", "kver", "kprep", "vc", "vver", "vprep")


_plans: "collections.OrderedDict[tuple, _Plan]" = collections.OrderedDict()      # least recently used first
_MAX_PLANS = 256


def _plan_cents(plan: _Plan, key_cents, value_cents, check):
    if plan.kc() is not key_cents or plan.kver != key_cents._version:
        if check is not None:
            check()                      # a codebook the plan has not seen: the caller's shape / dtype checks again
        plan.kprep, plan.kc, plan.kver = prepare_cents(key_cents), weakref.ref(key_cents), key_cents._version
    if value_cents is key_cents:
        return plan.kprep, plan.kprep
    if plan.vc is None or plan.vc() is not value_cents or plan.vver != value_cents._version:
        if check is not None:
            check()
        plan.vprep, plan.vc, plan.vver = prepare_cents(value_cents), weakref.ref(value_cents), value_cents._version
    return plan.kprep, plan.vprep


def _plan_signature(stream, query, key_codes, value_codes, key_residuals, value_residuals, v_page_ids, page_size, M, C):
    """Only what does NOT change from one decode step to the next is in the signature: the reference's contiguous
    (bs, nh_k, T, M) stores change their length and their batch / head strides at every flush, so those are written into the
    descriptor per call (round 3 keyed on them: every new T added a plan, and after 256 plans nothing was cached any more).
    What the signature leaves out is checked per call by _check_stores_per_call."""
    paged = v_page_ids is not None
    return (stream, query.shape, query.dtype, query.is_contiguous(), key_codes.shape[1], key_codes.stride()[2:], key_codes.dtype,
            value_codes.shape if paged else value_codes.stride()[2:], value_codes.dtype,
            key_residuals.shape, key_residuals.stride(), key_residuals.dtype, value_residuals.shape, value_residuals.stride(),
            value_residuals.dtype,
            (v_page_ids.shape, v_page_ids.dtype, v_page_ids.is_contiguous(), page_size) if paged else None,
            query.device.index, M, C)


def _check_stores_per_call(query, key_codes, value_codes, T, M, paged):
    """The T-dependent facts a cached plan cannot vouch for (the signature holds neither the stores' lengths nor their batch /
    head extents): the key store belongs to this query batch, and a row-major value store is at least as long as the key store
    it is read beside (million_transpose_v_codes reads T rows of it: a shorter store would be read out of bounds)."""
    if key_codes.dim() != 4 or key_codes.shape[0] != query.shape[0] or key_codes.shape[3] != M:
        raise RuntimeError(f"pq_decode_attn: key_codes {tuple(key_codes.shape)} do not match query batch {query.shape[0]} / M={M}")
    if not paged:
        if (value_codes.dim() != 4 or value_codes.shape[:2] != key_codes.shape[:2] or value_codes.shape[3] != M
                or value_codes.shape[2] < T):
            raise RuntimeError(f"pq_decode_attn: n_tokens exceeds value_codes (value_codes {tuple(value_codes.shape)} beside "
                               f"key_codes {tuple(key_codes.shape)})")


def decode_attn_planned(query, key_codes, value_codes, key_cents, value_cents, key_residuals, value_residuals, r, *, M, C,
                        v_page_ids=None, page_size=0, check=None):
    """The reference's production call (row-major K codes; V row-major, or a transposed page pool with page ids) with the
    per-call host work cut to the minimum.  Falls to pq_decode_attn (full validation) on the first call of a signature and
    for anything unusual (non-contiguous query, shapes off the MFMA kernels, empty store).  check: the caller's own argument
    checks, run on those slow-path calls only (a signature that passed them once passes them again: dtypes, the T-independent
    shapes and strides of every tensor are in the signature, the codebooks are identified by object and version; the stores'
    lengths and extents, which change at every flush, are checked on every call: _check_stores_per_call).
    Threads: a signature (the stream is part of it) must be driven from ONE thread at a time - the plan's descriptor is written
    per call; different streams have different plans and may be driven concurrently."""
    kshape = key_codes.shape
    T = kshape[2]
    paged = v_page_ids is not None
    stream = _stream()      # part of the signature: the workspace of a plan belongs to one (device, stream)
    # (the tuple of _plan_signature, spelled out: this is the per-call path of an eager harness - a Python call costs ~0.2 us)
    sig = (stream, query.shape, query.dtype, query.is_contiguous(), kshape[1], key_codes.stride()[2:], key_codes.dtype,
           value_codes.shape if paged else value_codes.stride()[2:], value_codes.dtype,
           key_residuals.shape, key_residuals.stride(), key_residuals.dtype, value_residuals.shape, value_residuals.stride(),
           value_residuals.dtype,
           (v_page_ids.shape, v_page_ids.dtype, v_page_ids.is_contiguous(), page_size) if paged else None,
           query.device.index, M, C)
    plan = _plans.get(sig)
    if plan is None:
        if check is not None:
            check()
        kp = prepare_cents(key_cents)
        vp = kp if value_cents is key_cents else prepare_cents(value_cents)
        out = pq_decode_attn(query.contiguous(), key_codes, value_codes, kp, vp, key_residuals, value_residuals, int(r), M=M, C=C,
                             **(dict(n_tokens=T, v_page_ids=v_page_ids.contiguous(), page_size=int(page_size)) if paged else {}))
        d = query.shape[3]
        fast_shape = (C in (128, 256) and d in (64, 128) and M in (16, 32, 64) and T > 0 and query.is_contiguous()
                      and key_codes.is_cuda and (not paged or v_page_ids.is_contiguous()))
        if fast_shape:                             # the call above has validated this signature
            while len(_plans) >= _MAX_PLANS:       # least recently used plan (and its workspace) goes
                _plans.popitem(last=False)
            plan = _Plan()
            if paged:
                plan.desc = make_attn_desc(query, key_residuals, nh_k=key_residuals.shape[1], M=M, C=C, n_tokens=T, r=int(r),
                                           k_paged=False, v_paged=True, page_size=int(page_size), n_pages_cap=v_page_ids.shape[2],
                                           page_ids_i64=v_page_ids.dtype == torch.int64, k_codes=key_codes, v_codes=value_codes)
            else:
                plan.desc = make_attn_desc(query, key_residuals, nh_k=key_residuals.shape[1], M=M, C=C, n_tokens=T, r=int(r),
                                           k_paged=False, v_paged=True, page_size=64, n_pages_cap=(T + 63) // 64,
                                           k_codes=key_codes, v_pages_dense=True)
            plan.desc_ref = ctypes.byref(plan.desc)
            plan.ws = attn_workspace(plan.desc, query.device)
            plan.ws_ptr, plan.ws_bytes = plan.ws.data_ptr(), plan.ws.numel()
            plan.v_dense = not paged
            plan.kc, plan.kver, plan.kprep = weakref.ref(key_cents), key_cents._version, kp
            plan.vc, plan.vver, plan.vprep = (None, 0, kp) if value_cents is key_cents else (weakref.ref(value_cents), value_cents._version, vp)
            _plans[sig] = plan
        return out
    try:
        _plans.move_to_end(sig)
    except KeyError:      # another thread's miss path evicted this signature between the lookup and here: the plan in hand is still valid
        pass
    # what the signature cannot vouch for (_check_stores_per_call, inlined): extents of the stores, which change at every flush
    if len(kshape) != 4 or kshape[0] != query.shape[0] or kshape[3] != M:
        _check_stores_per_call(query, key_codes, value_codes, T, M, paged)
    if not paged:
        vshape = value_codes.shape
        if len(vshape) != 4 or vshape[0] != kshape[0] or vshape[1] != kshape[1] or vshape[3] != M or vshape[2] < T:
            _check_stores_per_call(query, key_codes, value_codes, T, M, paged)
    if T <= 0:
        raise RuntimeError("decode_attn_planned: empty code store")      # (never cached: fast_shape needs T > 0)
    r = int(r)
    if not 0 <= r <= key_residuals.shape[2]:
        raise RuntimeError(f"pq_decode_attn: r={r} outside [0, {key_residuals.shape[2]}]")
    kp, vp = _plan_cents(plan, key_cents, value_cents, check)
    desc = plan.desc
    desc.n_tokens, desc.r = T, r
    desc.k_stride_b, desc.k_stride_h = key_codes.stride(0), key_codes.stride(1)      # (16-byte alignment: checked by the C entry)
    if plan.v_dense:
        desc.n_pages_cap = (T + 63) // 64
        v_ptr, ids_ptr = _v_pages_of(value_codes, T).data_ptr(), 0
    else:
        if v_page_ids.shape[2] * page_size < T:
            raise RuntimeError("pq_decode_attn: page table shorter than the key codes")
        v_ptr, ids_ptr = value_codes.data_ptr(), v_page_ids.data_ptr()
    ws_ptr, ws_bytes = plan.ws_ptr, plan.ws_bytes
    out = torch.empty_like(query)
    rc = L.load().million_pq_decode_attn(plan.desc_ref, query.data_ptr(), key_codes.data_ptr(), v_ptr, 0, ids_ptr, kp.data_ptr(),
                                         vp.data_ptr(), key_residuals.data_ptr(), value_residuals.data_ptr(), out.data_ptr(),
                                         ws_ptr, ws_bytes, stream)
    if rc:
        L.check(rc, "million_pq_decode_attn")
    return out


def residual_append(k_new: torch.Tensor, v_new: torch.Tensor, k_res: torch.Tensor, v_res: torch.Tensor, r: int,
                    resid_start: int = 0, dev_lengths: Optional[torch.Tensor] = None) -> None:
    """Write the new token's K/V rows into the residual window (replaces pq_utils.py:304-312)."""
    _need_cuda(k_new, v_new, k_res, v_res, dev_lengths)
    bs, nhk, cap, d = k_res.shape
    if k_new.shape != (bs, nhk, 1, d) or v_new.shape != (bs, nhk, 1, d):
        raise RuntimeError("residual_append: new rows must be (bs, nh_k, 1, d)")
    k_new, v_new = k_new.contiguous(), v_new.contiguous()
    lib = L.load()
    L.check(lib.million_residual_append(k_new.data_ptr(), v_new.data_ptr(), k_res.data_ptr(), v_res.data_ptr(),
                                        bs, nhk, d, cap, k_res.stride(0), k_res.stride(1), r, resid_start,
                                        _ptr(dev_lengths), _stream()), "million_residual_append")


def lengths_advance(dev_lengths: torch.Tensor, n_flushed: int, resid_cap: int) -> None:
    _need_cuda(dev_lengths)
    L.check(L.load().million_lengths_advance(dev_lengths.data_ptr(), dev_lengths.shape[0], n_flushed, resid_cap,
                                             _stream()), "million_lengths_advance")


def set_force_generic(on) -> None:
    """0 / False = auto, 1 / True = scalar fallback kernel only, 2 = grouped MFMA kernel instead of the streaming one, 4 = auto
    with the merge helpers giving up at once (every give-up bit preset), 8 = auto with helpers that have no patience (each gives
    up through the real atomic path), 16 = auto, but the lean kernel's shapes stay on the streaming / tile kernels and nothing runs as
    virtual kv heads, 64 = auto, but prompt attention runs its plain tile loop at d = 128 - million_hip.h: million_set_force_generic."""
    L.load().million_set_force_generic(int(on))


def tail_faults() -> int:
    """million_debug_tail_faults: waits for the device and returns (and clears) the number of split merges that gave up waiting
    for a partial since the last call (their heads were written as NaN).  0 in every healthy run; bench.py and smoke() read it
    after their timed regions and fail on anything else.  After a non-zero answer zero the workspace again."""
    return int(L.load().million_debug_tail_faults())


def attn_kernel_kind(desc: L.AttnDesc) -> int:
    return L.load().million_attn_kernel_kind(ctypes.byref(desc))
