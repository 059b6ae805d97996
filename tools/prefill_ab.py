#!/usr/bin/env python3
"""Development helper: the pipelined prompt-attention kernel (default at d = 128) against the plain one (million_set_force_generic(64)),
same tensors, interleaved rounds in one process; plus a correctness check of both against torch's fp32 SDPA on sampled rows."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for (bs, nh, nhk, n) in ((1, 32, 8, 4096), (1, 32, 8, 32768), (1, 32, 32, 32768), (1, 32, 8, 131072)):
    d = 128
    q = torch.randn(bs, nh, n, d, device=dev).half(); k = torch.randn(bs, nhk, n, d, device=dev).half(); v = torch.randn(bs, nhk, n, d, device=dev).half()
    flops = 2.0 * d * nh * bs * n * (n + 1)
    res = {}
    for rnd in range(3 if n <= 32768 else 2):
        for pol, name in ((0, "pipelined"), (64, "plain")):
            ops.set_force_generic(pol)
            res.setdefault(name, []).append(timed(lambda: ops.prefill_attn(q, k, v), reps=5 if n <= 32768 else 3))
    ops.set_force_generic(0)
    a = ops.prefill_attn(q, k, v)
    ops.set_force_generic(64)
    b_ = ops.prefill_attn(q, k, v)
    ops.set_force_generic(0)
    rows = torch.tensor(sorted(set([0, 1, 31, 32, 63, 64, 65, 255, 256, n // 2 - 1, n // 2, n - 65, n - 64, n - 33, n - 2, n - 1])), device=dev)
    G = nh // nhk
    kk, vv = k.float().repeat_interleave(G, dim=1), v.float().repeat_interleave(G, dim=1)
    qs = q.float()[:, :, rows]
    sc = torch.einsum("bhqd,bhkd->bhqk", qs, kk) / d ** 0.5
    msk = torch.arange(n, device=dev)[None, :] <= rows[:, None]
    sc = sc.masked_fill(~msk[None, None], float("-inf"))
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(sc, -1), vv)
    ea = ((a.float()[:, :, rows] - ref).norm() / ref.norm()).item(); eb = ((b_.float()[:, :, rows] - ref).norm() / ref.norm()).item()
    line = f"n={n:6d} nh_k={nhk:2d}: " + "  ".join(f"{name} {min(t):8.3f} ms = {flops / min(t) / 1e9:7.1f} TFLOP/s (all: {' '.join('%.3f' % x for x in t)})" for name, t in res.items())
    print(line + f"  rel-L2 vs fp32 on {len(rows)} rows: {ea:.1e} / {eb:.1e}", flush=True)
    del q, k, v, kk, vv
    torch.cuda.empty_cache()
