#!/usr/bin/env python3
"""Timing of the prompt-attention kernel (csrc/prefill.hip) against torch's scaled_dot_product_attention on the same
fp16 tensors (the reference's recipe: repeat_kv + SDPA, pq_utils.py:249-260), HIP events, median of repeats.
FLOPs = 4 d nh x (number of unmasked (query, key) pairs) = 2 d nh n (n + 1) for a causal prompt of n tokens."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for (bs, nh, nhk, n) in ((1, 32, 8, 4096), (1, 32, 8, 32768), (1, 32, 32, 4096), (1, 32, 32, 32768), (1, 32, 8, 131072)):
    d = 128
    q = torch.randn(bs, nh, n, d, device=dev).half()
    k = torch.randn(bs, nhk, n, d, device=dev).half()
    v = torch.randn(bs, nhk, n, d, device=dev).half()
    flops = 2.0 * d * nh * bs * n * (n + 1)
    ms = timeit(lambda: ops.prefill_attn(q, k, v))
    line = f"bs={bs} nh={nh} nh_k={nhk} n={n:6d}: prefill.hip {ms:9.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s ({flops / ms / 1e9 / 2500 * 100:4.1f}% of 2.5 PF dense fp16)"
    if n <= 32768:
        G = nh // nhk

        def sdpa():
            kk, vv = (k.repeat_interleave(G, dim=1), v.repeat_interleave(G, dim=1)) if G > 1 else (k, v)
            return torch.nn.functional.scaled_dot_product_attention(q, kk, vv, is_causal=True)
        ms2 = timeit(sdpa, reps=4)
        line += f" | repeat_kv + torch SDPA {ms2:9.3f} ms  {flops / ms2 / 1e9:7.1f} TFLOP/s"
    print(line, flush=True)
    del q, k, v
    torch.cuda.empty_cache()
