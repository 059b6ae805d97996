#!/usr/bin/env python3
"""Development helper: time the prompt-attention kernel of the library MILLION_HIP_LIB names (tools/ab_build.py variants) on one shape.
Ablation builds give wrong outputs by construction: only the time is read.  usage: prefill_var.py [n] [nh_k] [policy]"""
import os
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nhk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pol = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
nh, d = 32, 128
q = torch.randn(1, nh, n, d, device=dev).half(); k = torch.randn(1, nhk, n, d, device=dev).half(); v = torch.randn(1, nhk, n, d, device=dev).half()
ops.set_force_generic(pol)
ts = []
for i in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.prefill_attn(q, k, v); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts = sorted(ts[2:])
flops = 2.0 * d * nh * n * (n + 1)
print(f"{os.path.basename(os.environ.get('MILLION_HIP_LIB', 'in-tree')):28s} policy {pol:2d} n={n} nh_k={nhk}: median {ts[len(ts)//2]:8.3f} ms = {flops / ts[len(ts)//2] * 1e-9:7.1f} TFLOP/s  (min {ts[0]:.3f})")
