#!/usr/bin/env python3
"""Diagnostic: per-phase timing of the fused decode-attention kernel from in-kernel realtime stamps
(million_debug_set_stamp_buffer; lane 0 of wave 0 of every workgroup).  Shares = where a workgroup's
wall time goes; not a benchmark."""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import _lib as L, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=32768)
ap.add_argument("--nh", type=int, default=32)
ap.add_argument("--nhk", type=int, default=8)
ap.add_argument("--bs", type=int, default=1)
ap.add_argument("--r", type=int, default=100)
ap.add_argument("--layers", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda", 0)
M, C, d, ps = 64, 256, 128, 64
bs, nh, nhk, T = args.bs, args.nh, args.nhk, args.T
n_pages = (T + ps - 1) // ps
lib = L.load()
states = []
for l in range(args.layers):
    kpool = torch.randint(0, 256, (bs * nhk * n_pages, ps, M), dtype=torch.uint8, device=dev)
    vpool = torch.randint(0, 256, (bs * nhk * n_pages, M, ps), dtype=torch.uint8, device=dev)
    ids = torch.randperm(bs * nhk * n_pages, device=dev).to(torch.int32).reshape(bs, nhk, n_pages)
    states.append((kpool, vpool, ids))
kc = ops.prepare_cents(torch.randn(M, C, 2, device=dev).half())
vc = ops.prepare_cents(torch.randn(M, C, 2, device=dev).half())
q = torch.randn(bs, nh, 1, d, device=dev).half()
kr = torch.randn(bs, nhk, 128, d, device=dev).half()
vr = torch.randn(bs, nhk, 128, d, device=dev).half()
stamps = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)


def run(l):
    kp, vp, ids = states[l % args.layers]
    return ops.pq_decode_attn(q, kp, vp, kc, vc, kr, vr, args.r, M=M, C=C, n_tokens=T, k_page_ids=ids, v_page_ids=ids,
                              page_size=ps)


for i in range(2 * args.layers):
    run(i)
torch.cuda.synchronize()
lib.million_debug_set_stamp_buffer(stamps.data_ptr())
spans, last = [], None
for i in range(args.layers):
    stamps.zero_()
    run(i)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 16)
    s = s[s[:, 0] != 0]
    spans.append((s[:, 6].max() - s[:, 0].min()) / 100.0)
    last = s
lib.million_debug_set_stamp_buffer(None)
# stamp ids in program order, and what ends at each
order = [(0, "kernel start"), (7, "q + K codebook + code unit 0 requested"), (8, "K codebook written to LDS"),
         (1, "barrier 1 (K codebook)"), (9, "rest requested: residual rows, units 1-3, V codebook"),
         (2, "accumulators initialised"), (3, "all groups done (score pass, V-codebook barrier, value pass)"),
         (4, "wave-merge barrier"), (5, "wave merge done"), (10, "partial published + ticket"), (6, "end (last arriver: merge done)")]
s = last
t0 = s[:, 0].min()
print(f"workgroups {s.shape[0]}; kernel span (first start -> last end) per launch [us]: {[round(float(x), 2) for x in spans]}")
print("start skew of workgroups [us]: max %.2f" % ((s[:, 0].max() - t0) / 100.0))
prev = 0
for sid, name in order[1:]:
    d = (s[:, sid] - s[:, prev]) / 100.0
    print(f"  +{d.mean():6.2f} us (min {d.min():5.2f} max {d.max():5.2f})  -> {name}")
    prev = sid
la = s[s[:, 11] != 0]
if la.shape[0]:
    print("  last arrivers (%d): ticket -> weights ready %.2f | -> out written %.2f us" % (
        la.shape[0], ((la[:, 11] - la[:, 10]) / 100.0).mean(), ((la[:, 6] - la[:, 11]) / 100.0).mean()))
print("  end relative to first start: mean %.2f max %.2f us" % (((s[:, 6] - t0) / 100.0).mean(), ((s[:, 6] - t0) / 100.0).max()))
