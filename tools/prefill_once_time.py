#!/usr/bin/env python3
"""Median time of the prompt-attention kernel at one shape (development A/B with MILLION_HIP_LIB)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
same = len(sys.argv) > 2 and sys.argv[2] == "same-rows"      # every key / value row is row 0 (stride 0): all tile traffic hits in cache
dev = torch.device("cuda", 0)
q = torch.randn(1, 32, n, 128, device=dev).half()
k = torch.randn(1, 8, n, 128, device=dev).half()
v = torch.randn(1, 8, n, 128, device=dev).half()
if same:
    k = k[:, :, :64, :].repeat(1, 1, n // 64, 1)[:, :, :64, :].expand(1, 8, 64, 128)
    k = torch.as_strided(k.contiguous(), (1, 8, n, 128), (8 * 64 * 128, 64 * 128, 0, 1))
    v = torch.as_strided(v[:, :, :64, :].contiguous(), (1, 8, n, 128), (8 * 64 * 128, 64 * 128, 0, 1))
ts = []
for i in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.prefill_attn(q, k, v)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts = sorted(ts[2:])
ms = ts[len(ts) // 2]
print(f"n={n}{' same-rows' if same else ''}: {ms:.3f} ms  {2.0 * 128 * 32 * n * (n + 1) / ms / 1e9:.0f} TFLOP/s")
