#!/usr/bin/env python3
"""Median time of the prompt-attention kernel at one shape (development A/B with MILLION_HIP_LIB)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda", 0)
q = torch.randn(1, 32, n, 128, device=dev).half()
k = torch.randn(1, 8, n, 128, device=dev).half()
v = torch.randn(1, 8, n, 128, device=dev).half()
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
print(f"n={n}: {ms:.3f} ms  {2.0 * 128 * 32 * n * (n + 1) / ms / 1e9:.0f} TFLOP/s")
