"""Host-side mirror of the reference's KV-cache objects for the PQ-KV hot path.

Same names, constructor arguments and method meaning as the reference (scripts/utils/pq_utils.py:98-408
DynamicPQCache, scripts/utils/paged_pq_utils.py:10-397 PagedPQCache, scripts/utils/
dynamic_paged_pq_utils.py:10-321 PageManager, pq_utils.py:8-96 l2Ns / KernelRegistry), re-designed for
MI355X:

  * storage is PREALLOCATED (288 GB HBM): no torch.cat growth (pq_utils.py:145-146 reallocates O(T) per
    flush), no per-step pad/transpose of all V codes (paged_pq_utils.py:464-486);
  * the encode kernel writes codes straight into their final place (row-major store, K pages,
    transposed V pages);
  * the residual window is a ring buffer: a flush advances `resid_start`, nothing is shifted
    (paged_pq_utils.py:188-204 clones and copies the survivors);
  * lengths live on the device as well (int32 (bs, 4) = {n_tokens, r, resid_start, 0}) so a whole decode
    step can be captured in one hipGraph and replayed while the lengths change;
  * no global singletons (reference: metaclass Singleton), no exit(), no silent fallbacks: errors raise.

[QUIRK]s of the reference that are NOT reproduced: seen_tokens double counting on flush
(paged_pq_utils.py:208), page pool built from batch 0 / head 0 only (:457), causal mask with one query
row in the fallback (:888), the nested try/except fallbacks.
"""
from __future__ import annotations

import heapq
from typing import Dict, List, Optional, Set

import torch

from . import _lib as L
from . import ops


def l2Ns(l: int) -> int:
    """Split-count heuristic of the reference (pq_utils.py:8-22).  Only selects the binding NAME here:
    the HIP kernel sizes its own split count from the CU count."""
    if l > 2048:
        return 32
    if l > 256:
        return 16
    if l > 128:
        return 4
    if l > 64:
        return 2
    return 1


def scalarTypeToStr(scalar_t) -> str:
    if scalar_t == torch.float16:
        return "f16"
    if scalar_t == torch.float32:
        return "f32"
    raise ValueError(f"Unknown scalar type: {scalar_t}")


def nbits2dtype(nbits: int):
    """pq_utils.py:542-552."""
    if nbits <= 8:
        return torch.uint8
    if nbits <= 16:
        return torch.uint16
    if nbits <= 32:
        return torch.uint32
    if nbits <= 64:
        return torch.uint64
    raise ValueError("nbits must be <= 64")


class KernelRegistry:
    """pq_utils.py:32-96: resolves `bindings.flash_decoding_allocated_buffer_{T}u8_Ns{Ns}Lt{d}d{d}M{M}C{C}`
    by name and owns the (unused by the HIP path, but part of the signature) partial buffers."""

    def __init__(self, *, M=64, d=128, nbits=8, nh=32, scalar_t=torch.float16, device="cuda"):
        self.kernels, self.partial_out_buffers, self.partial_lse_buffers = {}, {}, {}
        self.M, self.d, self.nbits, self.nh, self.scalar_t, self.device = M, d, nbits, nh, scalar_t, device

    def get_kernel(self, l=4096):
        Ns = l2Ns(l)
        if Ns not in self.kernels:
            self.kernels[Ns] = self.get_custom_kernel_with_allocated_buffer(l)
        return self.kernels[Ns]

    def get_custom_kernel_with_allocated_buffer(self, l=4096):
        if self.nbits != 8:
            raise NotImplementedError("Only uint8 code type is supported for now")
        Ns = l2Ns(l)
        fname = (f"flash_decoding_allocated_buffer_{scalarTypeToStr(self.scalar_t)}u8_"
                 f"Ns{Ns}Lt{self.d}d{self.d}M{self.M}C{2 ** self.nbits}")
        func = getattr(__import__("bindings"), fname)
        po = torch.empty(1, self.nh, Ns + 1, self.d, dtype=self.scalar_t, device=self.device)
        pl = torch.empty(1, self.nh, Ns + 1, dtype=self.scalar_t, device=self.device)
        self.partial_out_buffers[Ns], self.partial_lse_buffers[Ns] = po, pl

        def flash_decoding(query, key_codes, value_codes, key_cents, value_cents, key_residuals, value_residuals, r):
            return func(query, key_codes, value_codes, key_cents, value_cents, key_residuals, value_residuals, r, po, pl)

        return flash_decoding


class PageManager:
    """Free-list page allocator with the semantics of dynamic_paged_pq_utils.py:10-321 (allocate_page,
    allocate_pages, free_page, growth by 1.5x capped by max_pages, stats) over ids only — the pools
    themselves are preallocated tensors owned by the cache, so growth never copies pool memory here."""

    def __init__(self, page_size: int = 64, initial_pages: int = 100, max_pages: Optional[int] = None, M: int = 64):
        if max_pages is not None and initial_pages > max_pages:
            initial_pages = max_pages
        self.page_size, self.M = page_size, M
        self.initial_pages, self.max_pages = initial_pages, max_pages
        self.current_active_pages = initial_pages
        self.free_pages: Set[int] = set(range(initial_pages))
        self._heap: List[int] = list(range(initial_pages))     # min-heap over free ids: lowest id first
        self.allocated_pages: Dict[int, Dict] = {}
        self.total_allocations = self.page_reuse_count = self.total_expansions = 0
        self._ever_used: Set[int] = set()

    def _expand_page_pool(self, additional_pages: Optional[int] = None):
        if additional_pages is None:
            additional_pages = max(self.current_active_pages // 2, 50)      # :80
        if self.max_pages is not None:
            room = self.max_pages - self.current_active_pages
            if room <= 0:
                raise RuntimeError(f"Cannot expand page pool: reached max_pages limit {self.max_pages}")
            additional_pages = min(additional_pages, room)
        new = self.current_active_pages + additional_pages
        self.free_pages.update(range(self.current_active_pages, new))
        for i in range(self.current_active_pages, new):
            heapq.heappush(self._heap, i)
        self.current_active_pages = new
        self.total_expansions += 1

    def allocate_page(self) -> int:
        if not self.free_pages:
            try:
                self._expand_page_pool()
            except RuntimeError as e:
                raise RuntimeError(f"No free pages available and cannot expand: {e}")
        pid = heapq.heappop(self._heap)     # deterministic (the reference pops an arbitrary set element)
        self.free_pages.remove(pid)
        if pid in self._ever_used:
            self.page_reuse_count += 1
        self._ever_used.add(pid)
        self.allocated_pages[pid] = {"allocation_count": 1}
        self.total_allocations += 1
        return pid

    def allocate_pages(self, n: int) -> List[int]:
        if n <= 0:
            return []
        if len(self.free_pages) < n:
            self._expand_page_pool(additional_pages=n - len(self.free_pages))
            if len(self.free_pages) < n:
                raise RuntimeError(f"Cannot bulk-allocate {n} pages: max_pages={self.max_pages}")
        return [self.allocate_page() for _ in range(n)]

    def free_page(self, page_id: int):
        if page_id not in self.allocated_pages:
            return
        del self.allocated_pages[page_id]
        self.free_pages.add(page_id)
        heapq.heappush(self._heap, page_id)

    def get_stats(self) -> Dict:
        return {"initial_pages": self.initial_pages, "current_active_pages": self.current_active_pages,
                "max_pages": self.max_pages, "allocated_pages": len(self.allocated_pages),
                "free_pages": len(self.free_pages),
                "utilization": len(self.allocated_pages) / max(self.current_active_pages, 1),
                "page_reuse_count": self.page_reuse_count, "total_allocations": self.total_allocations,
                "total_expansions": self.total_expansions}


class _CacheBase:
    def set_cent(self, key_cent: torch.Tensor, value_cent: torch.Tensor):
        """cent is (M, C, d//M) — one table shared by all layers and heads (pq_utils.py:149-159)."""
        self.key_cent = key_cent.to(self.device, self.scalar_t).contiguous()
        self.value_cent = self.key_cent if value_cent is key_cent else value_cent.to(self.device, self.scalar_t).contiguous()
        if self.key_cent.shape[1] != self.C or self.value_cent.shape[1] != self.C:
            raise ValueError(f"codebooks must have 2**nbits = {self.C} centroids per subspace")
        if self.C <= 256:      # LDS-ready images for the fused decode kernels and the fp32 image for the encoder
            self._kprep = ops.prepare_cents(self.key_cent, cache=False)
            self._vprep = self._kprep if self.value_cent is self.key_cent else ops.prepare_cents(self.value_cent, cache=False)
        else:
            self._kprep = self._vprep = None

    def _prefill_attention(self, q, k, v):
        """Causal attention of the prompt on its own fp16 K/V (reference: repeat_kv + torch SDPA, pq_utils.py:249-260):
        the MFMA flash kernel of this library (csrc/prefill.hip), the G query heads of a kv head sharing its K/V tiles."""
        return ops.prefill_attn(q, k, v, causal=True)


class DynamicPQCache(_CacheBase):
    """Row-major code store + residual window of Lt=d rows; flush ALL Lt rows when full
    (reference pq_utils.py:98-328).  `max_tokens` bounds the preallocated store."""

    def __init__(self, *, bs, nh, num_key_value_heads, M, layer_num, dtype=None, nbits=8, d=128,
                 scalar_t=torch.float16, max_tokens=32768 + 1024, device="cuda"):
        # nbits <= 8: uint8 codes, every path.  nbits 9..16: uint16 codes (nbits2dtype, pq_utils.py:542-552) on the
        # dequantise-then-attend paths (update, prefill); the fused decode kernels are uint8-only, as the reference's
        # are (KernelRegistry raises NotImplementedError, pq_utils.py:50-52).
        if not 1 <= nbits <= 16:
            raise NotImplementedError("nbits must be in 1..16")
        dtype = nbits2dtype(nbits) if dtype is None else dtype
        if dtype != nbits2dtype(nbits):
            raise ValueError(f"dtype {dtype} does not match nbits={nbits} ({nbits2dtype(nbits)})")
        self.bs, self.nh, self.num_key_value_heads, self.M, self.layer_num = bs, nh, num_key_value_heads, M, layer_num
        self.dtype, self.nbits, self.d, self.scalar_t, self.device = dtype, nbits, d, scalar_t, torch.device(device)
        self.C = 2 ** nbits
        self.max_residual_length = d          # "Lt = d", pq_utils.py:110
        self.max_tokens = (max_tokens + d - 1) // d * d
        self.registery = KernelRegistry(M=M, d=d, nbits=nbits, nh=nh, scalar_t=scalar_t, device=device)
        self.init_cache()

    def init_cache(self):
        z = lambda *s, dt: torch.zeros(*s, dtype=dt, device=self.device)
        nk, Lt = self.num_key_value_heads, self.max_residual_length
        self._k_store = [z(self.bs, nk, self.max_tokens, self.M, dt=self.dtype) for _ in range(self.layer_num)]
        self._v_store = [z(self.bs, nk, self.max_tokens, self.M, dt=self.dtype) for _ in range(self.layer_num)]
        self.key_residual_cache = [z(self.bs, nk, Lt, self.d, dt=self.scalar_t) for _ in range(self.layer_num)]
        self.value_residual_cache = [z(self.bs, nk, Lt, self.d, dt=self.scalar_t) for _ in range(self.layer_num)]
        self.seen_tokens = [0] * self.layer_num
        self.residualed_tokens = [0] * self.layer_num
        self._T = [0] * self.layer_num
        self._ws = None

    # views with the reference's shapes
    @property
    def key_cache(self):
        return [s[:, :, :t] for s, t in zip(self._k_store, self._T)]

    @property
    def value_cache(self):
        return [s[:, :, :t] for s, t in zip(self._v_store, self._T)]

    def _append_codes(self, X, layer_idx, n):
        T = self._T[layer_idx]
        if T + n > self.max_tokens:
            raise RuntimeError(f"DynamicPQCache: {T + n} tokens exceed max_tokens={self.max_tokens}")
        ops.pq_encode_into(X[0], self.key_cent, self._k_store[layer_idx], token_start=T, n=n, prepared=self._kprep)
        ops.pq_encode_into(X[1], self.value_cent, self._v_store[layer_idx], token_start=T, n=n, prepared=self._vprep)
        self._T[layer_idx] = T + n

    def prefill(self, query_states, key_states, value_states, layer_idx, distort_recent=False):
        """distort_recent=True attends to the DEQUANTISED prompt (perplexity-style evaluation of the quantiser,
        pq_utils.py:222-260 docstring); leave it False when serving."""
        n = key_states.size(2)
        T0 = self._T[layer_idx]
        self._append_codes((key_states, value_states), layer_idx, n)       # pq_utils.py:235-240
        self.seen_tokens[layer_idx] += n
        if distort_recent:                                                 # :242-246
            key_states = ops.pq_decode(self._k_store[layer_idx][:, :, T0:T0 + n], self.key_cent)
            value_states = ops.pq_decode(self._v_store[layer_idx][:, :, T0:T0 + n], self.value_cent)
        return self._prefill_attention(query_states, key_states, value_states)

    def update(self, key_states, value_states, layer_idx, distort_recent=False):
        """The reference's non-kernel path (pq_utils.py:166-220), DynamicCache-compatible: encode and store the new
        K/V, return the full-length fp16 K/V for a dense attention — the dequantised past followed by the new rows
        as they are (or dequantised too with distort_recent).  No residual window on this path (as in the reference)."""
        n = key_states.size(2)
        T0 = self._T[layer_idx]
        self._append_codes((key_states, value_states), layer_idx, n)
        self.seen_tokens[layer_idx] += n
        upto = T0 + n if distort_recent else T0
        past_k = ops.pq_decode(self._k_store[layer_idx][:, :, :upto], self.key_cent) if upto else None
        past_v = ops.pq_decode(self._v_store[layer_idx][:, :, :upto], self.value_cent) if upto else None
        if distort_recent:
            return past_k, past_v
        if past_k is None:
            return key_states, value_states
        return torch.cat([past_k, key_states.to(past_k.dtype)], dim=2), torch.cat([past_v, value_states.to(past_v.dtype)], dim=2)

    def decoding(self, query_states, key_states, value_states, layer_idx, fused=True):
        """One decode step of one layer (pq_utils.py:281-328).  fused=True (default): append + attention in ONE launch.
        fused=False: the reference's own call sequence - copy the new row into the window (:304-312), then the kernel
        `registery.get_kernel(l=seen_tokens)` resolves by name from `bindings` (:315-325)."""
        if self.nbits > 8:
            raise NotImplementedError("Only uint8 code type is supported by the fused decode kernels (use update())")
        Lt = self.max_residual_length
        if self.residualed_tokens[layer_idx] == Lt:                        # pq_utils.py:288-302
            self._append_codes((self.key_residual_cache[layer_idx], self.value_residual_cache[layer_idx]), layer_idx, Lt)
            self.residualed_tokens[layer_idx] = 0
        r = self.residualed_tokens[layer_idx]
        self.residualed_tokens[layer_idx] = r + 1
        self.seen_tokens[layer_idx] += 1
        if not fused:
            ops.residual_append(key_states, value_states, self.key_residual_cache[layer_idx],
                                self.value_residual_cache[layer_idx], r)
            kernel = self.registery.get_kernel(l=self.seen_tokens[layer_idx])
            T = self._T[layer_idx]
            return kernel(query_states, self._k_store[layer_idx][:, :, :T], self._v_store[layer_idx][:, :, :T],
                          self.key_cent, self.value_cent, self.key_residual_cache[layer_idx],
                          self.value_residual_cache[layer_idx], r + 1)
        if self._ws is None:      # sized once for max_tokens (partials + transposed-V scratch of the row-major path)
            desc = ops.make_attn_desc(query_states, self.key_residual_cache[layer_idx], nh_k=self.num_key_value_heads,
                                      M=self.M, C=self.C, n_tokens=self.max_tokens, r=0,
                                      k_codes=self._k_store[layer_idx], v_codes=self._v_store[layer_idx])
            self._ws = torch.zeros(L.load().million_attn_workspace_bytes(desc), dtype=torch.uint8, device=self.device)
        # append (:304-312) + attention (:314-326) in ONE launch
        return ops.pq_decode_attn(query_states, self._k_store[layer_idx], self._v_store[layer_idx], self._kprep,
                                  self._vprep, self.key_residual_cache[layer_idx], self.value_residual_cache[layer_idx],
                                  r, M=self.M, C=self.C, n_tokens=self._T[layer_idx], k_new=key_states,
                                  v_new=value_states, workspace=self._ws)

    @property
    def pq_cache_size(self):
        return sum(2 * self.bs * self.num_key_value_heads * t * self.M * self._k_store[0].element_size() for t in self._T)

    @property
    def residual_cache_size(self):
        return sum(c.numel() * c.element_size() for c in self.key_residual_cache + self.value_residual_cache)


class _PerLayer:
    """`cache._T[l]`-style access to a (layers, requests) host array: reading gives request 0's value (all requests move in
    lockstep unless slots are recycled, see PagedPQCache.release), writing sets every request of the layer."""

    def __init__(self, arr):
        self._a = arr

    def __getitem__(self, l):
        return int(self._a[l, 0])

    def __setitem__(self, l, v):
        self._a[l, :] = v

    def __len__(self):
        return self._a.shape[0]

    def __iter__(self):
        return (int(x) for x in self._a[:, 0])

    def __eq__(self, other):
        return list(self) == list(other)


class PagedPQCache(_CacheBase):
    """Paged code store: K pages (page_size, M) row-major, V pages (M, page_size) transposed, a residual
    ring of `extended_residual_size` rows; when r reaches it the OLDEST page_size rows are flushed
    (reference paged_pq_utils.py:10-397; policy :141-153, :359-361).

    One global pool per side serves all layers; page ids are handed out by a PageManager.  With
    `preallocate=True` (default) the page table of every (layer, b, hk) is filled for `max_tokens` at
    init, so decode steps touch no host state and are hipGraph-capturable.

    Lengths are kept per (layer, request) - on the host (numpy, (layers, bs)) and on the device (`lengths[l]`, int32
    (bs, 4)) - so that a batch slot can finish and be recycled while the others keep decoding: `release(b)` returns the
    slot's pages and zeroes its lengths, `prefill_request(b, ...)` encodes a new prompt into it.  Requests of different
    lengths share every launch through the device-resident lengths (`use_dev_lengths=True`)."""

    def __init__(self, *, bs, nh, num_key_value_heads, M, layer_num, dtype=torch.uint8, nbits=8, d=128,
                 scalar_t=torch.float16, page_size=64, extended_residual_size=128, max_pages_per_layer=None,
                 max_tokens=32768 + 1024, preallocate=True, device="cuda"):
        if nbits != 8 or dtype != torch.uint8:
            raise NotImplementedError("Only uint8 code type is supported for now")
        if page_size not in (32, 64, 128):
            raise ValueError("page_size must be 32, 64 or 128")
        if extended_residual_size < page_size:
            raise ValueError("extended_residual_size must be >= page_size")
        self.bs, self.nh, self.num_key_value_heads, self.M, self.layer_num = bs, nh, num_key_value_heads, M, layer_num
        self.dtype, self.nbits, self.d, self.scalar_t, self.device = dtype, nbits, d, scalar_t, torch.device(device)
        self.C = 2 ** nbits
        self.page_size, self.extended_residual_size = page_size, extended_residual_size
        self.max_residual_length = extended_residual_size
        self.n_pages_cap = (max_tokens + page_size - 1) // page_size
        if max_pages_per_layer is not None:
            self.n_pages_cap = min(self.n_pages_cap, max_pages_per_layer // max(bs * num_key_value_heads, 1))
        self.max_tokens = self.n_pages_cap * page_size
        self.preallocate = preallocate
        self.registery = KernelRegistry(M=M, d=d, nbits=nbits, nh=nh, scalar_t=scalar_t, device=device)
        self.init_cache()

    def init_cache(self):
        import numpy as np
        nk, cap = self.num_key_value_heads, self.extended_residual_size
        total = self.layer_num * self.bs * nk * self.n_pages_cap
        z = lambda *s, dt: torch.zeros(*s, dtype=dt, device=self.device)
        self.key_page_pool = z(total, self.page_size, self.M, dt=torch.uint8)
        self.value_page_pool = z(total, self.M, self.page_size, dt=torch.uint8)
        self.page_manager = PageManager(self.page_size, initial_pages=total, max_pages=total, M=self.M)
        # the layers' page tables, windows and length rows lie side by side (one allocation each; the per-layer lists are
        # views): one launch can then serve every layer (begin_step: encode-ahead, commit)
        self._page_ids_all = z(self.layer_num, self.bs, nk, self.n_pages_cap, dt=torch.int32)
        self._kres_all = z(self.layer_num, self.bs, nk, cap, self.d, dt=self.scalar_t)
        self._vres_all = z(self.layer_num, self.bs, nk, cap, self.d, dt=self.scalar_t)
        self._lengths_all = z(self.layer_num, self.bs, 4, dt=torch.int32)
        self.page_ids = [self._page_ids_all[l] for l in range(self.layer_num)]
        self.key_residual_cache = [self._kres_all[l] for l in range(self.layer_num)]
        self.value_residual_cache = [self._vres_all[l] for l in range(self.layer_num)]
        self.lengths = [self._lengths_all[l] for l in range(self.layer_num)]     # device mirror
        # host mirrors, (layers, requests)
        zi = lambda: np.zeros((self.layer_num, self.bs), dtype=np.int64)
        self._seen_a, self._r_a, self._T_a, self._rs_a, self._pages_a = zi(), zi(), zi(), zi(), zi()
        self._pre_a = zi()      # 1: the oldest page_size window rows are already encoded at tokens [T, T + page_size) (begin_step)
        self._pre_join = False
        self._host_pids = [[[[] for _ in range(nk)] for _ in range(self.bs)] for _ in range(self.layer_num)]
        self._ws = None
        self._side, self._flush_events, self._ahead = None, {}, None
        if self.preallocate:
            for l in range(self.layer_num):
                for b in range(self.bs):
                    self._assign_pages(l, self.n_pages_cap, b)

    # per-layer views with the reference's names (request 0's value; assignment sets every request)
    seen_tokens = property(lambda self: _PerLayer(self._seen_a), lambda self, v: self._set_rows(self._seen_a, v))
    residualed_tokens = property(lambda self: _PerLayer(self._r_a), lambda self, v: self._set_rows(self._r_a, v))
    _T = property(lambda self: _PerLayer(self._T_a), lambda self, v: self._set_rows(self._T_a, v))
    _rstart = property(lambda self: _PerLayer(self._rs_a), lambda self, v: self._set_rows(self._rs_a, v))
    _pages_assigned = property(lambda self: _PerLayer(self._pages_a))

    @staticmethod
    def _set_rows(arr, v):
        import numpy as np
        v = np.asarray([list(x) if hasattr(x, "__len__") else x for x in v] if not isinstance(v, np.ndarray) else v)
        arr[...] = v if v.ndim == 2 else v[:, None]

    def _lockstep(self, layer_idx) -> bool:
        a = (self._T_a[layer_idx], self._r_a[layer_idx], self._rs_a[layer_idx])
        return all((x == x[0]).all() for x in a)

    def _assign_pages(self, layer_idx, upto_pages, b=None):
        """Page ids for pages [have, upto_pages) of every kv head of request b (None: every request)."""
        for bb in (range(self.bs) if b is None else (b,)):
            have = int(self._pages_a[layer_idx, bb])
            if upto_pages <= have:
                continue
            if upto_pages > self.n_pages_cap:
                raise RuntimeError(f"PagedPQCache: {upto_pages} pages exceed capacity {self.n_pages_cap} (max_tokens)")
            n_new = upto_pages - have
            nk = self.num_key_value_heads
            ids = self.page_manager.allocate_pages(n_new * nk)
            for h in range(nk):
                self._host_pids[layer_idx][bb][h].extend(ids[h * n_new:(h + 1) * n_new])
            t = torch.tensor(ids, dtype=torch.int32).reshape(nk, n_new)
            self.page_ids[layer_idx][bb, :, have:upto_pages] = t.to(self.device)
            self._pages_a[layer_idx, bb] = upto_pages

    def _sync_lengths(self, layer_idx):
        import numpy as np
        rows = np.stack([self._T_a[layer_idx], self._r_a[layer_idx], self._rs_a[layer_idx], np.zeros(self.bs, dtype=np.int64)], axis=1)
        self.lengths[layer_idx].copy_(torch.from_numpy(rows.astype(np.int32)))

    def _encode_to_pages(self, K, V, layer_idx, n, *, b=None):
        """Encode n rows of K, V (bs or 1, nh_k, n, d) behind the T quantised tokens of every request (b = None: requests
        in lockstep) or of request b."""
        sel = slice(None) if b is None else slice(b, b + 1)
        T = int(self._T_a[layer_idx, 0 if b is None else b])
        if not self.preallocate:
            self._assign_pages(layer_idx, (T + n + self.page_size - 1) // self.page_size, b)
        elif T + n > self.max_tokens:
            raise RuntimeError(f"PagedPQCache: {T + n} tokens exceed max_tokens={self.max_tokens}")
        kw = dict(token_start=T, n=n, page_ids=self.page_ids[layer_idx][sel], page_size=self.page_size)
        ops.pq_encode_into(K, self.key_cent, self.key_page_pool, layout=L.MILLION_CODES_KPAGES, prepared=self._kprep, **kw)
        ops.pq_encode_into(V, self.value_cent, self.value_page_pool, layout=L.MILLION_CODES_VPAGES, prepared=self._vprep, **kw)

    def prefill(self, query_states, key_states, value_states, layer_idx, distort_recent=False):
        """Bulk encode of the prompt straight into pages (reference paged_pq_utils.py:216-320: encode,
        torch.cat, per-page permute+contiguous); the residual window stays empty."""
        if not self._lockstep(layer_idx):
            raise RuntimeError("PagedPQCache.prefill: requests are at different lengths; use prefill_request(b, ...)")
        n = key_states.size(2)
        self._encode_to_pages(key_states, value_states, layer_idx, n)
        self._T_a[layer_idx] += n
        self._seen_a[layer_idx] += n
        self._sync_lengths(layer_idx)
        return self._prefill_attention(query_states, key_states, value_states)

    def prefill_request(self, b, query_states, key_states, value_states, layer_idx):
        """The prompt of ONE request (tensors of batch 1) into batch slot b, the other slots untouched - a recycled slot
        (release) starts its next request while the rest of the batch keeps decoding."""
        if key_states.size(0) != 1 or self._T_a[layer_idx, b] or self._r_a[layer_idx, b]:
            raise RuntimeError("prefill_request: tensors of batch 1 into an empty slot (release it first)")
        n = key_states.size(2)
        self._encode_to_pages(key_states, value_states, layer_idx, n, b=b)
        self._T_a[layer_idx, b] += n
        self._seen_a[layer_idx, b] += n
        self._sync_lengths(layer_idx)
        return self._prefill_attention(query_states, key_states, value_states)

    def release(self, b):
        """Request b has finished: its pages go back to the PageManager (on-demand paging; a preallocated table keeps its
        fixed ids for the slot's next request), its host and device lengths return to zero.  The other slots, and any
        captured graph (lengths are read on the device), are not touched."""
        for l in range(self.layer_num):
            if not self.preallocate:
                for h in range(self.num_key_value_heads):
                    for pid in self._host_pids[l][b][h]:
                        self.page_manager.free_page(pid)
                    self._host_pids[l][b][h] = []
                self._pages_a[l, b] = 0
            self._T_a[l, b] = self._r_a[l, b] = self._rs_a[l, b] = self._seen_a[l, b] = self._pre_a[l, b] = 0
            self.lengths[l][b].zero_()

    reset_request = release

    def cleanup(self):
        """Reference PagedPQCache.cleanup (paged_pq_utils.py:1082-1118): drop every request's codes and window; the pools
        and (preallocated) page tables stay."""
        for b in range(self.bs):
            self.release(b)
        for l in range(self.layer_num):
            self.key_residual_cache[l].zero_()
            self.value_residual_cache[l].zero_()

    def _codes_rowmajor(self, layer_idx, T):
        """(bs, nh_k, T, M) row-major K and V codes gathered from the pages (dequantise-then-attend path only)."""
        npg = (T + self.page_size - 1) // self.page_size
        ids = self.page_ids[layer_idx][:, :, :npg].long()
        kc = self.key_page_pool[ids].reshape(self.bs, self.num_key_value_heads, npg * self.page_size, self.M)[:, :, :T]
        vc = self.value_page_pool[ids].permute(0, 1, 2, 4, 3).reshape(self.bs, self.num_key_value_heads, npg * self.page_size, self.M)[:, :, :T]
        return kc.contiguous(), vc.contiguous()

    def update(self, key_states, value_states, layer_idx, distort_recent=False):
        """The reference's non-kernel path on the paged store (paged_pq_utils.py:322-339 -> pq_utils.py:166-220): encode and
        store the new K/V rows into pages, return the full-length fp16 K/V for a dense attention - the dequantised past
        followed by the new rows as they are (or dequantised too with distort_recent).  No residual window on this path."""
        if not self._lockstep(layer_idx) or self._r_a[layer_idx, 0]:
            raise RuntimeError("PagedPQCache.update: lockstep requests and an empty residual window expected")
        n = key_states.size(2)
        T0 = int(self._T_a[layer_idx, 0])
        self._encode_to_pages(key_states, value_states, layer_idx, n)
        self._T_a[layer_idx] += n
        self._seen_a[layer_idx] += n
        self._sync_lengths(layer_idx)
        upto = T0 + n if distort_recent else T0
        if upto:
            kc, vc = self._codes_rowmajor(layer_idx, upto)
            past_k, past_v = ops.pq_decode(kc, self.key_cent), ops.pq_decode(vc, self.value_cent)
        else:
            past_k = past_v = None
        if distort_recent:
            return past_k, past_v
        if past_k is None:
            return key_states, value_states
        return torch.cat([past_k, key_states.to(past_k.dtype)], dim=2), torch.cat([past_v, value_states.to(past_v.dtype)], dim=2)

    def flush_to_pages(self, layer_idx: int, use_dev_lengths=False):
        """Encode the oldest page_size residual rows into a new K page and V page and move the window on
        (paged_pq_utils.py:130-210) - ONE launch (million_pq_flush): K encode, V encode and, with device-resident
        lengths, their advance.  Requests at different lengths (recycled slots): only those whose window is full flush -
        the launch reads every request's lengths on the device and skips the others (min_r)."""
        cap, ps = self.extended_residual_size, self.page_size
        full = self._r_a[layer_idx] >= cap
        lock = self._lockstep(layer_idx)
        # lockstep: the host flushes at r >= page_size, the device (min_r = cap with device lengths) only full windows - the
        # host mirror must not move when the device will not (a direct call with page_size <= r < cap)
        if lock and self._r_a[layer_idx, 0] < (cap if use_dev_lengths else ps):
            return
        if not lock and not full.any():
            return
        if not lock and not use_dev_lengths:
            raise RuntimeError("PagedPQCache: requests at different lengths need use_dev_lengths=True")
        who = range(self.bs) if lock else [b for b in range(self.bs) if full[b]]
        for b in who:
            T = int(self._T_a[layer_idx, b])
            if not self.preallocate:
                self._assign_pages(layer_idx, (T + ps + ps - 1) // ps, b)
            elif T + ps > self.max_tokens:
                raise RuntimeError(f"PagedPQCache: {T + ps} tokens exceed max_tokens={self.max_tokens}")
        ops.pq_flush(self.key_residual_cache[layer_idx], self.value_residual_cache[layer_idx], self.key_cent,
                     self.value_cent, self.key_page_pool, self.value_page_pool, self.page_ids[layer_idx],
                     n=ps, page_size=ps, token_start=int(self._T_a[layer_idx, 0]), x_row_start=int(self._rs_a[layer_idx, 0]),
                     dev_lengths=self.lengths[layer_idx] if use_dev_lengths else None,
                     min_r=cap if use_dev_lengths else 0)      # device lengths: only full windows flush (a captured step serves ragged batches too)
        for b in who:
            self._T_a[layer_idx, b] += ps
            self._r_a[layer_idx, b] -= ps
            self._rs_a[layer_idx, b] = (self._rs_a[layer_idx, b] + ps) % cap
            self._pre_a[layer_idx, b] = 0      # (rows encoded ahead were encoded again: same codes, same pages)

    # ---- encode-ahead: the flush without a flush step ---------------------------------------------------------------
    # The reference flushes the oldest page_size window rows when the window is full (paged_pq_utils.py:359-361).  Those
    # rows are complete page_size steps after the previous flush - extended_residual_size - page_size steps BEFORE they
    # are flushed - and never change again, and a PQ code depends on nothing but its row and the codebook.  begin_step()
    # therefore encodes them early: ONE launch for all layers (million_pq_flush_layers, advance = 0) on a side stream,
    # in a step that needs nothing from it, next to that step's attention launches (the kernel's 4-wave, <= 32-register
    # workgroups fit on a CU beside an attention workgroup; nobody waits for it before the end of the step).  The flush
    # step itself is then a commit: the lengths move (host mirror; device mirror: one million_lengths_advance launch for
    # all layers) and no code is computed.  Codes, pages and lengths after the commit are exactly those of the in-line
    # flush; whatever cannot be served this way (requests at different lengths, a step that was skipped) falls back to it.
    def next_step_kind(self) -> str:
        """'plain' | 'pre' (encode-ahead rides along) | 'commit' (flush step, codes already there) | 'flush' (flush step)."""
        cap = self.extended_residual_size
        full = self._r_a >= cap
        if full.any():
            return "commit" if full.all() and self._pre_a.all() else "flush"
        if (not self._pre_a.all() and self._r_a[0, 0] >= self.encode_ahead_at() and self._r_a[0, 0] >= self.page_size
                and self._all_lockstep()):
            groups = self._ea_groups()
            for gi, (g0, g1) in enumerate(groups):
                if not self._pre_a[g0:g1].all():
                    return "pre" if len(groups) == 1 else f"pre{gi}"
        return "plain"

    def _ea_groups(self):
        """Layer ranges encoded ahead per step: all layers in one step up to 32 (layer, request) pairs, else spread over
        consecutive steps (at 8 requests the attention launches leave few idle issue slots: all layers in one step made
        that step 1.6 x a plain one; 4 layers per step over 8 steps: see DESIGN.md 4.4)."""
        n = getattr(self, "encode_ahead_steps", None)
        if n is None:
            n = -(-self.bs * self.layer_num // 32)
        n = max(1, min(int(n), self.layer_num, max(1, self.page_size - 16)))
        per = -(-self.layer_num // n)
        return [(g0, min(g0 + per, self.layer_num)) for g0 in range(0, self.layer_num, per)]

    def encode_ahead_at(self) -> int:
        """Window fill at which the oldest page is encoded ahead (a few steps after the previous flush)."""
        return self.extended_residual_size - self.page_size + min(8, max(self.page_size // 8, 1))

    def _all_lockstep(self) -> bool:
        return all((a == a[0, 0]).all() for a in (self._T_a, self._r_a, self._rs_a))

    def begin_step(self, use_dev_lengths=False) -> str:
        """Call at the start of every decode step, before layer 0 (then decode every layer, in order).  Returns the kind of
        the step (next_step_kind).  Without it decoding_with_pages flushes in line, as the reference does."""
        kind = self.next_step_kind()
        cap, ps = self.extended_residual_size, self.page_size
        if kind == "commit":
            if use_dev_lengths:
                ops.lengths_advance(self._lengths_all.view(-1, 4), ps, cap)      # every layer, every request: one launch
            self._T_a += ps
            self._r_a -= ps
            self._rs_a[:] = (self._rs_a + ps) % cap
            self._pre_a[:] = 0
        elif kind == "flush":
            self.flush_ahead(use_dev_lengths=use_dev_lengths)
        elif kind.startswith("pre"):
            groups = self._ea_groups()
            G0, G1 = groups[int(kind[3:] or 0)]
            T = int(self._T_a[0, 0])
            if not self.preallocate:
                for l in range(G0, G1):
                    for b in range(self.bs):
                        self._assign_pages(l, (T + ps + ps - 1) // ps, b)
            elif T + ps > self.max_tokens:
                return "plain"      # the flush step will raise, as before
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            self._side.wait_stream(main)      # behind the appends of the previous step (the rows themselves are older)
            with torch.cuda.stream(self._side):
                # a grid of ONE layer's size whose workgroups walk the layers (one flush workgroup per CU at a time: what fits
                # beside an attention workgroup), in `encode_ahead_launches` launches
                nl_ = getattr(self, "encode_ahead_launches", None)
                if nl_ is None:
                    # replayed from a hipGraph: one launch per layer (the work trickles through the whole step: measured
                    # 1.04 x a plain step against 1.09 x for one long launch); eager: one launch (the host pays per launch)
                    nl_ = self.layer_num if torch.cuda.is_current_stream_capturing() else 1
                nl_ = max(1, min(int(nl_), G1 - G0))
                per = (G1 - G0 + nl_ - 1) // nl_
                for g0 in range(G0, G1, per):
                    g1 = min(g0 + per, G1)
                    ops.pq_flush(self._kres_all[g0:g1], self._vres_all[g0:g1], self.key_cent, self.value_cent, self.key_page_pool,
                                 self.value_page_pool, self._page_ids_all[g0:g1], n=ps, page_size=ps, token_start=T,
                                 x_row_start=int(self._rs_a[0, 0]),
                                 dev_lengths=self._lengths_all[g0:g1] if use_dev_lengths else None,
                                 min_r=ps if use_dev_lengths else 0, advance=False)
            self._pre_a[G0:G1] = 1
            self._pre_join = True      # joined behind the last layer's attention (decoding_with_pages)
        return kind

    def flush_ahead(self, use_dev_lengths=False, depth=2):
        """Flush the full windows of this step on a side stream instead of in front of each layer's attention: a layer's
        window rows and lengths do not depend on the current step's computation, so its flush launch (a 4-wave, 21-register
        kernel that fits on a CU beside an attention workgroup) runs under the attention launches of the layers before
        it; decoding_with_pages(layer) only waits for its own layer's flush event.  Call it at the start of a decode step
        (next_step_flushes() says when); every layer must then be decoded in this step, in order (the side stream is
        joined layer by layer - inside a hipGraph capture this becomes a fork / join of graph branches).
        The launches are issued `depth` layers ahead of the attention that needs them, not all at once: the host (and a
        replayed hipGraph, which submits its nodes in capture order) otherwise spends 32 flush submissions before the
        first attention launch gets out - measured: the main branch of the replayed step started ~300 us late.
        Results are identical to the in-line flush."""
        cap = self.extended_residual_size
        todo = [l for l in range(self.layer_num) if (self._r_a[l] >= cap).any()]
        if not todo:
            return
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)      # (a high-priority side stream changes nothing: measured)
            self._flush_events = {}
        self._side.wait_stream(main)      # fork: behind everything this stream has queued (the previous step's appends)
        self._ahead = (todo, use_dev_lengths, max(1, int(depth)))
        self._ahead_issue(upto_layer=-1)

    def _ahead_issue(self, upto_layer):
        """Issue the pending side-stream flushes of the layers <= upto_layer + depth (in layer order)."""
        todo, use_dl, depth = self._ahead
        n = 0
        while n < len(todo) and todo[n] <= upto_layer + depth:
            n += 1
        if n:
            if upto_layer >= 0:
                # behind the attention launches issued so far (layers < upto_layer): a graph replay submits its nodes in
                # an order of the runtime's choosing - without this edge ROCm submitted the whole flush chain before the
                # first attention node (main branch ~300 us late); with it any topological order interleaves the two
                # chains.  The flush then has the attention launches of layers upto_layer .. upto_layer + depth - 1 to hide under.
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                for l in todo[:n]:
                    self.flush_to_pages(l, use_dev_lengths=use_dl)
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    self._flush_events[l] = ev
            del todo[:n]
        if not todo:
            self._ahead = None

    def decoding_with_pages(self, query_states, key_states, value_states, layer_idx, out=None, use_dev_lengths=False):
        """One decode step of one layer (paged_pq_utils.py:341-386): flush if the window is full, append the
        new token's K/V row, fused attention over pages + window.  With use_dev_lengths=True every length
        is read on the device (the host mirror is still advanced), which makes the call graph-capturable and lets
        requests of different lengths share the launch."""
        cap = self.extended_residual_size
        if getattr(self, "_ahead", None) is not None:
            self._ahead_issue(upto_layer=layer_idx)                                      # keep the side stream `depth` layers ahead
        ev = self._flush_events.pop(layer_idx, None) if getattr(self, "_side", None) is not None else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)                                   # flushed ahead on the side stream
        elif (self._r_a[layer_idx] >= cap).any():                                        # :359-361
            self.flush_to_pages(layer_idx, use_dev_lengths=use_dev_lengths)
        lock = self._lockstep(layer_idx)
        if not lock and not use_dev_lengths:
            raise RuntimeError("PagedPQCache: requests at different lengths need use_dev_lengths=True")
        r, rs = int(self._r_a[layer_idx, 0]), int(self._rs_a[layer_idx, 0])
        dl = self.lengths[layer_idx] if use_dev_lengths else None
        if self._ws is None:      # one workspace per cache (calls of one cache are stream-ordered)
            desc = ops.make_attn_desc(query_states, self.key_residual_cache[layer_idx], nh_k=self.num_key_value_heads,
                                      M=self.M, C=self.C, n_tokens=0, r=0)
            self._ws = torch.zeros(L.load().million_attn_workspace_bytes(desc), dtype=torch.uint8, device=self.device)
        self._r_a[layer_idx] += 1
        self._seen_a[layer_idx] += 1
        # append (:377-380) + attention (:386) in ONE launch: the new row is attended to and parked in the window
        res = ops.pq_decode_attn(query_states, self.key_page_pool, self.value_page_pool, self._kprep, self._vprep,
                                 self.key_residual_cache[layer_idx], self.value_residual_cache[layer_idx], r,
                                 k_new=key_states, v_new=value_states,
                                 M=self.M, C=self.C, n_tokens=self.max_tokens if use_dev_lengths else int(self._T_a[layer_idx, 0]),
                                 resid_start=rs, k_page_ids=self.page_ids[layer_idx],
                                 v_page_ids=self.page_ids[layer_idx], page_size=self.page_size, out=out,
                                 dev_lengths=dl, workspace=self._ws)
        if self._pre_join and layer_idx == self.layer_num - 1:
            torch.cuda.current_stream().wait_stream(self._side)      # the step's encode-ahead launch ends with the step
            self._pre_join = False
        return res

    # the reference's non-paged entry point keeps working on the paged store
    decoding = decoding_with_pages

    # ---- host mirror management for captured (hipGraph) decode steps ------------------------------
    # During stream capture the Python above runs once (advancing the host mirror) but no kernel
    # executes; under replay the kernels run but no Python does.  The harness therefore restores the
    # mirror after capture and calls note_replayed_step() after every replay.
    def host_state(self):
        """(seen, r, T, resid_start), each (layers, requests) - what set_host_state takes back (it also takes one value per
        layer for all requests)."""
        return (self._seen_a.copy(), self._r_a.copy(), self._T_a.copy(), self._rs_a.copy(), self._pre_a.copy())

    def set_host_state(self, st):
        """(seen, r, T, resid_start[, encoded-ahead flags]); four entries: the flags are cleared."""
        for arr, v in zip((self._seen_a, self._r_a, self._T_a, self._rs_a), st):
            self._set_rows(arr, v)
        self._pre_a[:] = 0
        if len(st) > 4:
            self._set_rows(self._pre_a, st[4])

    def capture_states(self, st=None):
        """[(kind, host state)] - one host state per kind of step, for capturing one hipGraph per kind with device-resident
        lengths (begin_step + all layers).  st: the state to derive them from (default: now)."""
        st = self.host_state() if st is None else st
        L_, cap = self.layer_num, self.extended_residual_size
        r_now = int(st[1][0][0])
        r_plain = r_now if r_now < cap else 0
        groups = self._ea_groups()
        pre = [("pre" if len(groups) == 1 else f"pre{gi}",
                (st[0], [self.encode_ahead_at() + gi] * L_, st[2], st[3], [1 if l < g0 else 0 for l in range(L_)]))
               for gi, (g0, g1) in enumerate(groups)]
        return [("plain", (st[0], [r_plain] * L_, st[2], st[3], [1] * L_))] + pre + [
                ("commit", (st[0], [cap] * L_, st[2], st[3], [1] * L_)),
                ("flush", (st[0], [cap] * L_, st[2], st[3], [0] * L_))]

    def next_step_flushes(self, layer_idx: int = 0) -> bool:
        return bool((self._r_a[layer_idx] >= self.extended_residual_size).any())

    def note_replayed_step(self, kind=None):
        """The host-mirror transitions of one replayed step of the given kind (default: next_step_kind())."""
        kind = self.next_step_kind() if kind is None else kind
        cap, ps = self.extended_residual_size, self.page_size
        if kind in ("commit", "flush"):
            full = self._r_a >= cap
            self._T_a[full] += ps
            self._r_a[full] -= ps
            self._rs_a[full] = (self._rs_a[full] + ps) % cap
            self._pre_a[full] = 0
        elif kind.startswith("pre"):
            g0, g1 = self._ea_groups()[int(kind[3:] or 0)]
            if not (self.preallocate and int(self._T_a[0, 0]) + ps > self.max_tokens):      # begin_step encoded nothing then
                self._pre_a[g0:g1] = 1
        self._r_a += 1
        self._seen_a += 1

    def get_cache_stats(self) -> Dict:
        """paged_pq_utils.py:898-939 (same keys) + the page-manager counters."""
        layer_stats = [{"layer_idx": l, "seen_tokens": self.seen_tokens[l], "residual_tokens": self.residualed_tokens[l],
                        "key_cache_tokens": self._T[l], "value_cache_tokens": self._T[l]} for l in range(self.layer_num)]
        mb = 1024 * 1024
        cache_mb = float(2 * self.num_key_value_heads * self.M * self._T_a.sum()) / mb
        resid_mb = sum(c.numel() * c.element_size() for c in self.key_residual_cache + self.value_residual_cache) / mb
        return {"layer_stats": layer_stats, "total_memory_usage_mb": cache_mb + resid_mb,
                "memory_breakdown": {"cache_memory_mb": cache_mb, "residual_memory_mb": resid_mb,
                                     "prefill_residual_memory_mb": 0.0, "total_memory_mb": cache_mb + resid_mb},
                "page_manager": self.page_manager.get_stats()}
