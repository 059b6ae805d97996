#!/usr/bin/env python3
"""Host cost of the drop-in call: N eager calls of the reference's 10-argument binding (and of the 13-argument paged-V one)
on a tiny problem (T = 64: the kernel itself takes a few microseconds), wall clock per call with the GPU kept busy behind
the host (no synchronise inside the loop), and the GPU time per call from HIP events over the same calls.  Wall per call is
the host's cost whenever it exceeds the GPU time per call (it does).  Reference boundary: a compiled pybind11 module
(scripts/modeldb/bindings/bindings.template.cpp:11-63); call pattern: scripts/utils/pq_utils.py:61-94."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bindings  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
bs, nh, nhk, d, M, C, T, r, ps = 1, 32, 8, 128, 64, 256, 64, 17, 64
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn(bs, nh, 1, d, device=dev, generator=g).half()
kc = torch.randint(0, C, (bs, nhk, T, M), device=dev, dtype=torch.uint8, generator=g)
vc = torch.randint(0, C, (bs, nhk, T, M), device=dev, dtype=torch.uint8, generator=g)
kcent = torch.randn(M, C, d // M, device=dev, generator=g).half()
vcent = torch.randn(M, C, d // M, device=dev, generator=g).half()
kres = torch.randn(bs, nhk, d, d, device=dev, generator=g).half()
vres = torch.randn(bs, nhk, d, d, device=dev, generator=g).half()
po = torch.empty(bs, nh, 33, d, device=dev, dtype=torch.float16)
pl = torch.empty(bs, nh, 33, device=dev, dtype=torch.float16)
vpool = vc.permute(0, 1, 3, 2).reshape(bs * nhk, M, ps).contiguous()
vids = torch.arange(bs * nhk, device=dev, dtype=torch.int64).reshape(bs, nhk, 1)
f10 = getattr(bindings, f"flash_decoding_allocated_buffer_f16u8_Ns2Lt{d}d{d}M{M}C{C}")
f13 = getattr(bindings, f"flash_decoding_paged_v_f16u8_Ns2Lt{d}d{d}M{M}C{C}")
calls = {
    "10-arg (row-major K and V, the same tensors every call)": lambda: f10(q, kc, vc, kcent, vcent, kres, vres, r, po, pl),
    "10-arg, codes passed as fresh views ([:, :, :T]) every call": lambda: f10(q, kc[:, :, :T], vc[:, :, :T], kcent, vcent, kres, vres, r, po, pl),
    "13-arg paged V (row-major K, int64 page ids)": lambda: f13(q, kc, kcent, kres, vids, vpool, vcent, vres, r, 1, ps, po, pl),
}
for name, fn in calls.items():
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    wall = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize()
    span = e0.elapsed_time(e1) * 1e3 / N
    # GPU time of the kernel alone: the same call replayed from a hipGraph (no host in the loop)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(64):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    gpu = e0.elapsed_time(e1) * 1e3 / (20 * 64)
    print(f"{name}: host wall {wall:6.2f} us/call over {N} eager calls (event span {span:6.2f} us/call); kernel alone {gpu:5.2f} us/call (hipGraph replay); "
          f"host overhead = wall - kernel = {wall - gpu:6.2f} us", flush=True)
