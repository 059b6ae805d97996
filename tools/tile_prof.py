#!/usr/bin/env python3
"""Development helper: per-phase shader-clock totals of the tile kernel (library built with -DMILLION_TILE_PROF).
    MILLION_HIP_LIB=build/ab/libmillion_tileprof.so python tools/tile_prof.py bs,T,M,d"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import _lib as L, ops  # noqa: E402

bs, T, M, d = (int(x) for x in sys.argv[1].split(","))
dev = torch.device("cuda", 0)
lib = L.load()
nh, nhk, C, ps, r = 32, 8, 256, 64, 100
n_pages = (T + ps - 1) // ps
kpool = torch.randint(0, 256, (bs * nhk * n_pages, ps, M), dtype=torch.uint8, device=dev)
vpool = torch.randint(0, 256, (bs * nhk * n_pages, M, ps), dtype=torch.uint8, device=dev)
ids = torch.arange(bs * nhk * n_pages, device=dev).to(torch.int32).reshape(bs, nhk, n_pages)
kc = ops.prepare_cents(torch.randn(M, C, d // M, device=dev).half(), cache=False)
vc = ops.prepare_cents(torch.randn(M, C, d // M, device=dev).half(), cache=False)
q = torch.randn(bs, nh, 1, d, device=dev).half()
kr = torch.randn(bs, nhk, 128, d, device=dev).half()
vr = torch.randn(bs, nhk, 128, d, device=dev).half()
buf = torch.zeros(66 * bs * nhk * 16 * 8, dtype=torch.int64, device=dev)      # (splits + 1) x (b, kv head) x up to 16 waves x 8 words
for it in range(3):
    buf.zero_()
    lib.million_debug_set_stamp_buffer(buf.data_ptr())
    ops.pq_decode_attn(q, kpool, vpool, kc, vc, kr, vr, r, M=M, C=C, n_tokens=T, k_page_ids=ids, v_page_ids=ids, page_size=ps)
    torch.cuda.synchronize()
    lib.million_debug_set_stamp_buffer(0)
a = buf.cpu().numpy().reshape(-1, 8)
a = a[a[:, 4] > 0]
tiles = a[:, 4].astype(float)
names = ["K gathers -> score MFMAs, V gathers issued", "softmax", "packs + value MFMAs", "(unused)"]      # round 4: direct operand gathers
print(f"waves with tiles: {len(a)}; tiles per wave: mean {tiles.mean():.1f}; wave lifetime mean {a[:, 5].mean():.0f} clocks")
import numpy as np
for i, n in enumerate(names):      # (the merge's own diagnostic stamps land in a few of these words: median, not mean)
    print(f"  {n:45s} {np.median(a[:, i] / tiles):8.1f} clocks per tile")
print(f"  lifetime / tiles = {np.median(a[:, 5] / tiles):.1f} clocks")
