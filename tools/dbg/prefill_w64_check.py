#!/usr/bin/env python3
"""Development helper: the experimental 4-wave x 64-row prompt-attention kernel (dev builds, million_set_force_generic(128)) against the plain
loop (64) on a few shapes, and its time at 32K."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from million_amd import ops  # noqa: E402
dev = torch.device("cuda", 0)
torch.manual_seed(1)
for (bs, nh, nhk, nq, nkv, pos0, causal) in ((1, 8, 2, 1, 1, 0, True), (1, 8, 2, 33, 33, 0, True), (2, 4, 4, 100, 100, 0, True), (1, 8, 2, 257, 257, 0, True), (1, 8, 2, 64, 200, 136, True),
                                            (1, 8, 2, 70, 333, 0, False), (1, 32, 8, 1000, 1000, 0, True), (1, 32, 8, 4096, 4096, 0, True)):
    q = torch.randn(bs, nh, nq, 128, device=dev).half(); k = torch.randn(bs, nhk, nkv, 128, device=dev).half(); v = torch.randn(bs, nhk, nkv, 128, device=dev).half()
    ops.set_force_generic(64); a = ops.prefill_attn(q, k, v, causal=causal, q_pos0=pos0).float()
    ops.set_force_generic(128); b = ops.prefill_attn(q, k, v, causal=causal, q_pos0=pos0).float()
    torch.cuda.synchronize()
    rel = ((a - b).norm() / a.norm()).item()
    print(f"shape {(bs, nh, nhk, nq, nkv, pos0, causal)}: rel diff w64 vs plain {rel:.2e}  finite {bool(torch.isfinite(b).all())}")
ops.set_force_generic(0)
