#!/usr/bin/env python3
"""Diagnostic: per-phase timing of the fused decode-attention kernel from in-kernel realtime stamps
(million_debug_set_stamp_buffer).  Shares = where a workgroup's wall time goes; not a benchmark."""
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
acc = []
for i in range(args.layers):
    stamps.zero_()
    run(i)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 16)
    s = s[s[:, 0] != 0]
    acc.append(s)
lib.million_debug_set_stamp_buffer(None)
names = ["prologue: loads issued, tables -> LDS, barrier", "residual-window partial (waves 0,1)",
         "code units (MFMA loop)", "wave merge barrier", "wave merge compute", "publish (+ last-arriver merge)"]
tot = []
for s in acc:
    t0 = s[:, 0].min()
    tot.append(((s[:, 6].max() - t0) / 100.0, s.shape[0]))
    rel = (s - t0) / 100.0      # us since the first workgroup started
    d = np.diff(s[:, :7].astype(np.float64), axis=1) / 100.0
print(f"workgroups {tot[-1][1]}; kernel span (first start -> last end) per launch [us]:", [round(t[0], 2) for t in tot])
print("last launch: start skew of workgroups [us]: min %.2f max %.2f" % (rel[:, 0].min(), rel[:, 0].max()))
for i, n in enumerate(names):
    print(f"  phase {i} {n:45s} mean {d[:, i].mean():6.2f}  min {d[:, i].min():6.2f}  max {d[:, i].max():6.2f} us")
print("  end of phase 5 rel. to first start: mean %.2f max %.2f us" % (rel[:, 6].mean(), rel[:, 6].max()))
if s[:, 10].any():
    la = s[s[:, 11] != 0]
    print("  publish: stores+ticket %.2f us (mean, all WGs); last arrivers (%d): ticket->weights ready %.2f | ->out written %.2f us" % (
        ((s[:, 10] - s[:, 5]) / 100.0).mean(), la.shape[0], ((la[:, 11] - la[:, 10]) / 100.0).mean(), ((la[:, 6] - la[:, 11]) / 100.0).mean()))
if s[:, 7].any():
    if s[:, 9].any():
        print("  wave 0: start->all loads issued %.2f | residual partial (incl. wait for its rows) %.2f us" % (
            ((s[:, 9] - s[:, 0]) / 100.0).mean(), ((s[:, 7] - s[:, 9]) / 100.0).mean()))
    a = (s[:, 7] - s[:, 0]) / 100.0
    b_ = (s[:, 8] - s[:, 7]) / 100.0
    c = (s[:, 1] - s[:, 8]) / 100.0
    print("  prologue detail: start->loads issued %.2f | issued->tables in LDS %.2f | ->barrier done %.2f us (means)" % (a.mean(), b_.mean(), c.mean()))
