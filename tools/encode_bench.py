#!/usr/bin/env python3
"""Timing of the PQ encode kernel: bulk (prefill-sized) and flush-sized calls (HIP events, median of repeats)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda", 0)
M, C, d = 64, 256, 128
cents = torch.randn(M, C, d // M, device=dev).half()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for bs, nhk, n in ((1, 8, 32768), (1, 8, 4096), (1, 32, 4096), (1, 8, 64), (16, 8, 64)):
    X = torch.randn(bs, nhk, n, d, device=dev).half()
    for layout, name in ((L.MILLION_CODES_ROWMAJOR, "rowmajor"), (L.MILLION_CODES_KPAGES, "kpages"), (L.MILLION_CODES_VPAGES, "vpages")):
        ps = 64
        npg = (n + ps - 1) // ps
        if layout == L.MILLION_CODES_ROWMAJOR:
            dst = torch.zeros(bs, nhk, n, M, dtype=torch.uint8, device=dev)
            kw = {}
        else:
            dst = torch.zeros(bs * nhk * npg, ps, M, dtype=torch.uint8, device=dev)
            ids = torch.arange(bs * nhk * npg, device=dev, dtype=torch.int32).reshape(bs, nhk, npg)
            kw = dict(page_ids=ids, page_size=ps)
        us = timeit(lambda: ops.pq_encode_into(X, cents, dst, layout=layout, token_start=0, n=n, **kw))
        rows = bs * nhk * n
        # fp32 direct-form work: per (row, m, c): d_m sub + d_m mul + (d_m-1) add + cmp/select
        print(f"bs={bs} nh_k={nhk} n={n:6d} {name:9s}: {us:9.1f} us  {rows / us:8.2f} rows/us  "
              f"{rows * M * C / us / 1e6:7.2f} T(centroid tests)/s")

# ---- a window flush (64 rows x 8 kv heads, K and V): one fused launch vs two encodes + the length advance ----
bs, nhk, cap, ps = 1, 8, 128, 64
kw_ = torch.randn(bs, nhk, cap, d, device=dev).half()
vw_ = torch.randn(bs, nhk, cap, d, device=dev).half()
ids = torch.arange(bs * nhk * 8, device=dev, dtype=torch.int32).reshape(bs, nhk, 8)
kpool = torch.zeros(bs * nhk * 8, ps, M, dtype=torch.uint8, device=dev)
vpool = torch.zeros(bs * nhk * 8, M, ps, dtype=torch.uint8, device=dev)
dl = torch.tensor([[128, 128, 0, 0]] * bs, dtype=torch.int32, device=dev)


def flush_fused():
    dl.copy_(torch.tensor([[128, 128, 0, 0]] * bs, dtype=torch.int32), non_blocking=True)
    ops.pq_flush(kw_, vw_, cents, cents, kpool, vpool, ids, n=ps, page_size=ps, dev_lengths=dl)


def flush_separate():
    dl.copy_(torch.tensor([[128, 128, 0, 0]] * bs, dtype=torch.int32), non_blocking=True)
    kwargs = dict(n=ps, page_ids=ids, page_size=ps, x_row_mod=cap, dev_lengths=dl)
    ops.pq_encode_into(kw_, cents, kpool, layout=L.MILLION_CODES_KPAGES, **kwargs)
    ops.pq_encode_into(vw_, cents, vpool, layout=L.MILLION_CODES_VPAGES, **kwargs)
    ops.lengths_advance(dl, ps, cap)


def graph_time(fn, n=32):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (10 * n)


dl_host = torch.tensor([[128, 128, 0, 0]] * bs, dtype=torch.int32).pin_memory()


def flush_fused_g():
    ops.pq_flush(kw_, vw_, cents, cents, kpool, vpool, ids, n=ps, page_size=ps, token_start=128)


def flush_separate_g():
    kwargs = dict(token_start=128, n=ps, page_ids=ids, page_size=ps, x_row_mod=cap)
    ops.pq_encode_into(kw_, cents, kpool, layout=L.MILLION_CODES_KPAGES, **kwargs)
    ops.pq_encode_into(vw_, cents, vpool, layout=L.MILLION_CODES_VPAGES, **kwargs)
    ops.lengths_advance(dl, ps, cap)


print(f"window flush of one layer (64 rows x {nhk} kv heads, K + V) inside a hipGraph of 32 flushes: "
      f"one launch (million_pq_flush) {graph_time(flush_fused_g):6.2f} us | two encodes + lengths_advance (round 1) "
      f"{graph_time(flush_separate_g):6.2f} us")
