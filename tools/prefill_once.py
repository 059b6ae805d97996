#!/usr/bin/env python3
"""A few launches of the prompt-attention kernel at one shape (for rocprofv3 --pmc / --kernel-trace passes)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nhk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
q = torch.randn(1, 32, n, 128, device=dev).half()
k = torch.randn(1, nhk, n, 128, device=dev).half()
v = torch.randn(1, nhk, n, 128, device=dev).half()
for _ in range(4):
    o = ops.prefill_attn(q, k, v)
torch.cuda.synchronize()
print("ok", float(o.float().abs().mean()))
