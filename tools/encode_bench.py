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
