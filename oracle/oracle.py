"""oracle/oracle.py — CPU restatement of MILLION's PQ-KV hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker.  Nothing under ``million_amd/`` or ``bindings/`` imports it.

Two layers:
  * numpy restatements (readable, vectorised) of the reference's Python functions;
  * a ctypes loader for ``oracle/libpq_oracle.so`` (``pq_oracle.c``: plain C, ``-ffp-contract=off``)
    which is the bit-exact definition used against the HIP kernels.

Reference files restated (all under /root/reference/scripts):
  utils/pq_utils.py:8-22      l2Ns
  utils/pq_utils.py:410-449   sa_encode_4d      (cdist + argmin; CPU-runnable form)
  utils/pq_utils.py:451-499   sa_encode_4d_keops (fp32 direct form ((x-c)**2).sum(-1).argmin)
  utils/pq_utils.py:501-540   sa_decode_4d
  utils/pq_utils.py:360-368   oracle attention formula (non-causal softmax over [K_hat; K_res[:r]])
  utils/pq_utils.py:281-328   DynamicPQCache.decoding flush policy
  utils/paged_pq_utils.py:130-210,341-397  PagedPQCache flush policy and page layout
  modeldb/bindings/Interface.template.cu:26-120 + Kernel.cuh:11-166,1038-1270  split structure

Parity status: see the header of pq_oracle.c and DESIGN.md ("Oracle").
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    """Compile pq_oracle.c with gcc (a few hundred ms).  Building the checker is not using it."""
    so = _HERE / "libpq_oracle.so"
    src = _HERE / "pq_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-s", "libpq_oracle.so"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(str(so))
        c_fp = ctypes.POINTER(ctypes.c_float)
        c_dp = ctypes.POINTER(ctypes.c_double)
        c_u8 = ctypes.POINTER(ctypes.c_uint8)
        i64, i32 = ctypes.c_int64, ctypes.c_int
        L.pq_encode_direct.argtypes = [c_fp, c_fp, c_u8, i64, i32, i32, i32]
        L.pq_encode_direct.restype = None
        L.pq_encode_direct_gap.argtypes = [c_fp, c_fp, c_u8, c_fp, i64, i32, i32, i32]
        L.pq_encode_direct_gap.restype = None
        L.pq_decode.argtypes = [c_u8, c_fp, c_fp, i64, i32, i32, i32]
        L.pq_decode.restype = None
        c_u16 = ctypes.POINTER(ctypes.c_uint16)
        L.pq_encode_direct_u16.argtypes = [c_fp, c_fp, c_u16, i64, i32, i32, i32]
        L.pq_encode_direct_u16.restype = None
        L.pq_decode_u16.argtypes = [c_u16, c_fp, c_fp, i64, i32, i32, i32]
        L.pq_decode_u16.restype = None
        L.decode_attn_f64.argtypes = [c_fp, c_u8, c_u8, c_fp, c_fp, c_fp, c_fp, c_dp, c_dp,
                                      i32, i32, i32, i64, i32, i32, i32, i32, i32]
        L.decode_attn_f64.restype = None
        L.decode_attn_split_f32.argtypes = [c_fp, c_u8, c_u8, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                            i32, i32, i32, i64, i32, i32, i32, i32, i32, i32]
        L.decode_attn_split_f32.restype = None
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _u8(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


# --------------------------------------------------------------------------------------------------
# l2Ns (pq_utils.py:8-22)
# --------------------------------------------------------------------------------------------------
def l2Ns(l: int) -> int:
    if l > 2048:
        return 32
    if l > 256:
        return 16
    if l > 128:
        return 4
    if l > 64:
        return 2
    return 1


# --------------------------------------------------------------------------------------------------
# encode
# --------------------------------------------------------------------------------------------------
def pq_encode(X, cents) -> np.ndarray:
    """Direct-form fp32 encode (C oracle).  X (bs, nh_k, n, d), cents (M, C, d_m) -> (bs, nh_k, n, M) u8.

    Follows sa_encode_4d_keops, pq_utils.py:451-499: the reference permutes X to (bs*nh_k*M, n, d_m),
    upcasts to fp32 (:483-484), takes argmin_c of ((x-c)**2).sum(-1) (:491-494) and permutes the
    indices back to (bs, nh_k, n, M) (:497-499).  The permutes only reorder independent (vector, m)
    problems, so the C loop walks (vector, m) directly.
    """
    X = _f32(X)
    cents = _f32(cents)
    bs, nhk, n, d = X.shape
    M, C, dm = cents.shape
    assert M * dm == d and C <= 65536
    if C > 256:      # nbits 9..16 -> uint16 codes (nbits2dtype, pq_utils.py:542-552)
        codes = np.empty((bs, nhk, n, M), dtype=np.uint16)
        lib().pq_encode_direct_u16(_p(X, ctypes.c_float), _p(cents, ctypes.c_float), _p(codes, ctypes.c_uint16),
                                   bs * nhk * n, d, M, C)
        return codes
    codes = np.empty((bs, nhk, n, M), dtype=np.uint8)
    lib().pq_encode_direct(_p(X, ctypes.c_float), _p(cents, ctypes.c_float), _p(codes, ctypes.c_uint8),
                           bs * nhk * n, d, M, C)
    return codes


def pq_encode_with_gap(X, cents):
    X = _f32(X)
    cents = _f32(cents)
    bs, nhk, n, d = X.shape
    M, C, dm = cents.shape
    codes = np.empty((bs, nhk, n, M), dtype=np.uint8)
    gap = np.empty((bs, nhk, n, M), dtype=np.float32)
    lib().pq_encode_direct_gap(_p(X, ctypes.c_float), _p(cents, ctypes.c_float),
                               _p(codes, ctypes.c_uint8), _p(gap, ctypes.c_float),
                               bs * nhk * n, d, M, C)
    return codes, gap


def pq_encode_numpy(X, cents) -> np.ndarray:
    """Same arithmetic as pq_encode in numpy (sub, mul, sequential add; first-min argmin)."""
    X = _f32(X)
    cents = _f32(cents)
    bs, nhk, n, d = X.shape
    M, C, dm = cents.shape
    Xr = X.reshape(bs * nhk * n, M, 1, dm)
    dt = np.uint8 if C <= 256 else np.uint16
    out = np.empty((bs * nhk * n, M), dtype=dt)
    step = max(1, 4096 * 256 // C)
    for i0 in range(0, Xr.shape[0], step):
        e = Xr[i0:i0 + step] - cents[None]            # (v, M, C, dm) fp32
        sq = e * e
        acc = np.zeros(sq.shape[:-1], dtype=np.float32) + sq[..., 0]
        for k in range(1, dm):
            acc = acc + sq[..., k]
        out[i0:i0 + step] = np.argmin(acc, axis=-1).astype(dt)
    return out.reshape(bs, nhk, n, M)


def pq_encode_cdist_torch(X, cents):
    """Restatement of sa_encode_4d (pq_utils.py:410-449): torch.cdist(p=2) + argmin, input dtype."""
    import torch
    X = torch.as_tensor(np.asarray(X))
    C = torch.as_tensor(np.asarray(cents))
    bs, nh, n, d = X.shape
    M, c, dm = C.shape
    Ce = C.unsqueeze(0).unsqueeze(0).expand(bs, nh, -1, -1, -1).reshape(bs * nh * M, c, dm)
    Xe = X.reshape(bs, nh, n, M, dm).permute(0, 1, 3, 2, 4).reshape(bs * nh * M, n, dm)
    dis = torch.cdist(Xe, Ce, p=2).reshape(bs, nh, M, n, c).permute(0, 1, 3, 2, 4)
    return torch.argmin(dis, dim=-1).contiguous().to(torch.uint8).numpy()


def pq_encode_direct_torch(X, cents):
    """The direct form ((x - c) ** 2).sum(-1).argmin over c of sa_encode_4d_keops (pq_utils.py:483-494) as dense torch ops on
    the host's cores, fp32 (torch.argmin: lowest index on ties).  CPU-baseline timing only (bench.py); the bit-exact
    checker is pq_encode above."""
    import torch
    X = torch.as_tensor(np.asarray(X)).float()
    C = torch.as_tensor(np.asarray(cents)).float()
    bs, nh, n, d = X.shape
    M, c, dm = C.shape
    Xe = X.reshape(bs * nh * n, M, 1, dm)
    out = torch.empty(bs * nh * n, M, dtype=torch.uint8 if c <= 256 else torch.int32)
    step = max(1, (1 << 24) // (M * c * dm))      # ~64 MB of differences per chunk
    for i0 in range(0, Xe.shape[0], step):
        e = Xe[i0:i0 + step] - C[None]
        out[i0:i0 + step] = (e * e).sum(-1).argmin(-1).to(out.dtype)
    return out.reshape(bs, nh, n, M).numpy()


# --------------------------------------------------------------------------------------------------
# decode (sa_decode_4d)
# --------------------------------------------------------------------------------------------------
def pq_decode(codes, cents) -> np.ndarray:
    cents_f = _f32(cents)
    if np.asarray(codes).dtype == np.uint16:
        codes = np.ascontiguousarray(codes)
        bs, nhk, n, M = codes.shape
        Mc, C, dm = cents_f.shape
        assert M == Mc
        out = np.empty((bs, nhk, n, M * dm), dtype=np.float32)
        lib().pq_decode_u16(_p(codes, ctypes.c_uint16), _p(cents_f, ctypes.c_float), _p(out, ctypes.c_float),
                            bs * nhk * n, M * dm, M, C)
        return out
    codes = _u8(codes)
    bs, nhk, n, M = codes.shape
    Mc, C, dm = cents_f.shape
    assert M == Mc
    out = np.empty((bs, nhk, n, M * dm), dtype=np.float32)
    lib().pq_decode(_p(codes, ctypes.c_uint8), _p(cents_f, ctypes.c_float), _p(out, ctypes.c_float),
                    bs * nhk * n, M * dm, M, C)
    return out


def pq_decode_numpy(codes, cents) -> np.ndarray:
    codes = np.asarray(codes)
    cents = np.asarray(cents)
    bs, nhk, n, M = codes.shape
    m_idx = np.arange(M)[None, None, None, :]
    return cents[m_idx, codes.astype(np.int64)].reshape(bs, nhk, n, M * cents.shape[-1])


# --------------------------------------------------------------------------------------------------
# page layout helpers (paged_pq_utils.py:173-175, 464-500; design doc MILLION_技术分析文档.md:1330-1345)
# --------------------------------------------------------------------------------------------------
def v_rowmajor_to_pool(v_codes, page_size: int):
    """(bs, nh_k, T, M) row-major codes -> (pool (n_pool, M, page_size) u8, page_ids (bs, nh_k, n_pages) i64).

    Page p of (b, hk) holds tokens [p*ps, (p+1)*ps) transposed: pool[pid, m, off] = code[t = p*ps+off, m]
    (addressing of MILLION_技术分析文档.md:1330-1340).  The tail of the last page is zero-padded
    (paged_pq_utils.py:464-470).  Page ids are assigned b-major, hk, then p (any bijection is legal).
    """
    v_codes = _u8(v_codes)
    bs, nhk, T, M = v_codes.shape
    n_pages = (T + page_size - 1) // page_size
    pool = np.zeros((max(bs * nhk * n_pages, 1), M, page_size), dtype=np.uint8)
    ids = np.zeros((bs, nhk, n_pages), dtype=np.int64)
    pid = 0
    for b in range(bs):
        for h in range(nhk):
            for p in range(n_pages):
                t0, t1 = p * page_size, min((p + 1) * page_size, T)
                pool[pid, :, : t1 - t0] = v_codes[b, h, t0:t1, :].T
                ids[b, h, p] = pid
                pid += 1
    return pool, ids


def k_rowmajor_to_pool(k_codes, page_size: int):
    """K pages keep tokens row-major inside the page: pool[pid, off, m] = code[t = p*ps+off, m]."""
    k_codes = _u8(k_codes)
    bs, nhk, T, M = k_codes.shape
    n_pages = (T + page_size - 1) // page_size
    pool = np.zeros((max(bs * nhk * n_pages, 1), page_size, M), dtype=np.uint8)
    ids = np.zeros((bs, nhk, n_pages), dtype=np.int64)
    pid = 0
    for b in range(bs):
        for h in range(nhk):
            for p in range(n_pages):
                t0, t1 = p * page_size, min((p + 1) * page_size, T)
                pool[pid, : t1 - t0, :] = k_codes[b, h, t0:t1, :]
                ids[b, h, p] = pid
                pid += 1
    return pool, ids


def pool_to_v_rowmajor(pool, ids, T: int):
    pool = np.asarray(pool)
    ids = np.asarray(ids)
    bs, nhk, n_pages = ids.shape
    _, M, ps = pool.shape
    out = np.zeros((bs, nhk, T, M), dtype=np.uint8)
    for b in range(bs):
        for h in range(nhk):
            for p in range(n_pages):
                t0, t1 = p * ps, min((p + 1) * ps, T)
                if t1 > t0:
                    out[b, h, t0:t1, :] = pool[ids[b, h, p], :, : t1 - t0].T
    return out


def pool_to_k_rowmajor(pool, ids, T: int):
    """Inverse of k_rowmajor_to_pool: pool (n_pool, page_size, M) + page ids -> (bs, nh_k, T, M)."""
    pool = np.asarray(pool)
    ids = np.asarray(ids)
    bs, nhk, n_pages = ids.shape
    _, ps, M = pool.shape
    out = np.zeros((bs, nhk, T, M), dtype=np.uint8)
    for b in range(bs):
        for h in range(nhk):
            for p in range(n_pages):
                t0, t1 = p * ps, min((p + 1) * ps, T)
                if t1 > t0:
                    out[b, h, t0:t1, :] = pool[ids[b, h, p], : t1 - t0, :]
    return out


# --------------------------------------------------------------------------------------------------
# decode-step attention
# --------------------------------------------------------------------------------------------------
def decode_attn(q, k_codes, v_codes, k_cents, v_cents, k_res, v_res, r: int, return_lse=False):
    """fp64 gold.  q (bs, nh, 1, d) or (bs, nh, d); codes (bs, nh_k, T, M); residuals (bs, nh_k, Lt, d).

    Returns (bs, nh, 1, d) float64 (and lse (bs, nh) if asked).  Formula: pq_utils.py:360-368.
    """
    q = _f32(q)
    if q.ndim == 4:
        q = q[:, :, 0, :]
    q = np.ascontiguousarray(q)
    k_codes, v_codes = _u8(k_codes), _u8(v_codes)
    k_cents, v_cents = _f32(k_cents), _f32(v_cents)
    k_res, v_res = _f32(k_res), _f32(v_res)
    bs, nh, d = q.shape
    _, nhk, T, M = k_codes.shape
    C = k_cents.shape[1]
    Lt = k_res.shape[2]
    assert 0 <= r <= Lt
    out = np.empty((bs, nh, d), dtype=np.float64)
    lse = np.empty((bs, nh), dtype=np.float64)
    lib().decode_attn_f64(_p(q, ctypes.c_float), _p(k_codes, ctypes.c_uint8), _p(v_codes, ctypes.c_uint8),
                          _p(k_cents, ctypes.c_float), _p(v_cents, ctypes.c_float),
                          _p(k_res, ctypes.c_float), _p(v_res, ctypes.c_float),
                          _p(out, ctypes.c_double), _p(lse, ctypes.c_double),
                          bs, nh, nhk, T, r, Lt, d, M, C)
    out = out[:, :, None, :]
    return (out, lse) if return_lse else out


def decode_attn_split(q, k_codes, v_codes, k_cents, v_cents, k_res, v_res, r: int, Ns: int):
    """fp32 restatement of the reference's split-KV launch structure; returns (out, partial_out, partial_lse)."""
    q = _f32(q)
    if q.ndim == 4:
        q = q[:, :, 0, :]
    q = np.ascontiguousarray(q)
    k_codes, v_codes = _u8(k_codes), _u8(v_codes)
    k_cents, v_cents = _f32(k_cents), _f32(v_cents)
    k_res, v_res = _f32(k_res), _f32(v_res)
    bs, nh, d = q.shape
    _, nhk, T, M = k_codes.shape
    C = k_cents.shape[1]
    Lt = k_res.shape[2]
    po = np.empty((bs, nh, Ns + 1, d), dtype=np.float32)
    pl = np.empty((bs, nh, Ns + 1), dtype=np.float32)
    out = np.empty((bs, nh, d), dtype=np.float32)
    lib().decode_attn_split_f32(_p(q, ctypes.c_float), _p(k_codes, ctypes.c_uint8), _p(v_codes, ctypes.c_uint8),
                                _p(k_cents, ctypes.c_float), _p(v_cents, ctypes.c_float),
                                _p(k_res, ctypes.c_float), _p(v_res, ctypes.c_float),
                                _p(po, ctypes.c_float), _p(pl, ctypes.c_float), _p(out, ctypes.c_float),
                                bs, nh, nhk, T, r, Lt, d, M, C, Ns)
    return out[:, :, None, :], po, pl


def decode_attn_dense_numpy(q, k_codes, v_codes, k_cents, v_cents, k_res, v_res, r: int, dtype=np.float64):
    """The reference's own check, literally: sa_decode_4d K,V -> cat residual[:r] -> softmax(qK^T/sqrt(d))V
    (pq_utils.py:360-368), in numpy at `dtype`."""
    q = np.asarray(q, dtype=dtype)
    if q.ndim == 3:
        q = q[:, :, None, :]
    K = np.concatenate([pq_decode_numpy(k_codes, np.asarray(k_cents, dtype=dtype)),
                        np.asarray(k_res, dtype=dtype)[:, :, :r]], axis=2)
    V = np.concatenate([pq_decode_numpy(v_codes, np.asarray(v_cents, dtype=dtype)),
                        np.asarray(v_res, dtype=dtype)[:, :, :r]], axis=2)
    bs, nh, _, d = q.shape
    G = nh // K.shape[1]
    K = np.repeat(K, G, axis=1)
    V = np.repeat(V, G, axis=1)
    s = np.einsum("bhqd,bhtd->bhqt", q, K) / np.sqrt(dtype(d))
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=-1, keepdims=True)
    return np.einsum("bhqt,bhtd->bhqd", p, V)


# --------------------------------------------------------------------------------------------------
# cache policies (host logic oracle): token accounting of the two reference caches
# --------------------------------------------------------------------------------------------------
class DynamicPolicy:
    """DynamicPQCache.decoding (pq_utils.py:281-328): when the residual is full (r == Lt) encode all
    Lt rows and append them to the code store, r = 0; then append the new token at row r."""

    def __init__(self, Lt: int = 128, prefill: int = 0):
        self.Lt, self.T, self.r = Lt, prefill, 0       # prefill quantises every prompt token (:235-240)
        self.flushes = 0

    def step(self):
        if self.r == self.Lt:
            self.T += self.Lt
            self.r = 0
            self.flushes += 1
        self.r += 1
        return self.T, self.r


class PagedPolicy:
    """PagedPQCache.decoding_with_pages (paged_pq_utils.py:341-386) with flush_to_pages (:130-210):
    when r >= extended_residual_size flush the OLDEST page_size rows, keep the rest (shifted to the
    front in the reference, :188-200), then append.  [QUIRK not reproduced] the reference also adds
    page_size to seen_tokens at :208 although those tokens were counted on append."""

    def __init__(self, page_size: int = 64, residual: int = 128, prefill: int = 0):
        self.ps, self.cap, self.T, self.r = page_size, residual, prefill, 0
        self.flushes = 0

    def step(self):
        if self.r >= self.cap:
            self.T += self.ps
            self.r -= self.ps
            self.flushes += 1
        self.r += 1
        return self.T, self.r


# --------------------------------------------------------------------------------------------------
# CPU baseline (bench.py cpu_baseline leg): the reference's CPU-runnable PyTorch path, restated
# --------------------------------------------------------------------------------------------------
def decode_attn_torch_cpu(q, k_codes, v_codes, k_cents, v_cents, k_res, v_res, r: int, dtype="float32"):
    """torch-CPU restatement of the reference's fallback math with the causal bug removed:
    sa_decode_4d(K), sa_decode_4d(V) (pq_utils.py:501-540: expand + torch.gather) -> cat residual[:r]
    -> repeat_kv -> scaled_dot_product_attention (paged_pq_utils.py:860-888, is_causal dropped).
    Inputs are torch CPU tensors; runs on torch's intra-op thread pool."""
    import torch

    def sa_decode(codes, C):
        bs, nh, n, M = codes.shape
        _, c, dm = C.shape
        Ce = C.unsqueeze(0).unsqueeze(0).expand(bs, nh, -1, -1, -1)
        idx = codes.to(torch.long).unsqueeze(-1).expand(-1, -1, -1, -1, dm).unsqueeze(-2)
        dec = torch.gather(Ce.unsqueeze(2).expand(-1, -1, n, -1, -1, -1), 4, idx).squeeze(4)
        return dec.reshape(bs, nh, n, M * dm)

    dt = getattr(torch, dtype)
    q, k_cents, v_cents = q.to(dt), k_cents.to(dt), v_cents.to(dt)
    K = torch.cat([sa_decode(k_codes, k_cents), k_res.to(dt)[:, :, :r]], dim=2)
    V = torch.cat([sa_decode(v_codes, v_cents), v_res.to(dt)[:, :, :r]], dim=2)
    G = q.shape[1] // K.shape[1]
    K, V = K.repeat_interleave(G, dim=1), V.repeat_interleave(G, dim=1)
    return torch.nn.functional.scaled_dot_product_attention(q, K, V)
