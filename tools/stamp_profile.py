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
ap.add_argument("--grouped", action="store_true", help="labels of the grouped kernel")
ap.add_argument("--raw", action="store_true", help="print every stamp id that was written, ordered by mean time")
ap.add_argument("--M", type=int, default=64)
ap.add_argument("--dev-lengths", action="store_true")
ap.add_argument("--policy", type=int, default=0, help="million_set_force_generic value (2 = grouped only, 3 = prefer pipelined)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
M, C, d, ps = args.M, 256, 128, 64
bs, nh, nhk, T = args.bs, args.nh, args.nhk, args.T
n_pages = (T + ps - 1) // ps
lib = L.load()
lib.million_set_force_generic(args.policy)
states = []
for l in range(args.layers):
    kpool = torch.randint(0, 256, (bs * nhk * n_pages, ps, M), dtype=torch.uint8, device=dev)
    vpool = torch.randint(0, 256, (bs * nhk * n_pages, M, ps), dtype=torch.uint8, device=dev)
    ids = torch.randperm(bs * nhk * n_pages, device=dev).to(torch.int32).reshape(bs, nhk, n_pages)
    states.append((kpool, vpool, ids))
kc = ops.prepare_cents(torch.randn(M, C, d // M, device=dev).half())
vc = ops.prepare_cents(torch.randn(M, C, d // M, device=dev).half())
q = torch.randn(bs, nh, 1, d, device=dev).half()
kr = torch.randn(bs, nhk, 128, d, device=dev).half()
vr = torch.randn(bs, nhk, 128, d, device=dev).half()
dlen = torch.tensor([[T, args.r, 0, 0]] * bs, dtype=torch.int32, device=dev) if args.dev_lengths else None
NW, NS = 8, 32                               # common.h: kStampWaves, kStampSlots
stamps = torch.zeros(bs * nhk * 64 * NW * NS, dtype=torch.int64, device=dev)


def run(l):
    kp, vp, ids = states[l % args.layers]
    return ops.pq_decode_attn(q, kp, vp, kc, vc, kr, vr, args.r, M=M, C=C, n_tokens=T, k_page_ids=ids, v_page_ids=ids,
                              page_size=ps, dev_lengths=dlen)


for i in range(2 * args.layers):
    run(i)
torch.cuda.synchronize()
lib.million_debug_set_stamp_buffer(stamps.data_ptr())
spans, last, crit = [], None, []
for i in range(args.layers):
    stamps.zero_()
    run(i)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, NW, NS)
    s = s[s[:, 0, 0] != 0]
    spans.append((s[:, :, 6].max() - s[:, :, 0].min()) / 100.0)
    # the critical path of this launch: the workgroup that raises its flag last (start offset + its own time to the flag),
    # then the merge behind it
    t0_ = s[:, :, 0].min()
    w = int(s[:, 0, 10].argmax())
    crit.append(((s[w, 0, 0] - t0_) / 100.0, (s[w, 0, 10] - s[w, 0, 0]) / 100.0, (s[:, :, 6].max() - s[w, 0, 10]) / 100.0,
                 (s[:, 0, 0].max() - t0_) / 100.0, ((s[:, 0, 10] - s[:, 0, 0]) / 100.0).mean(), ((s[:, 0, 10] - s[:, 0, 0]) / 100.0).max()))
    last = s
lib.million_debug_set_stamp_buffer(None)
# stamp ids in program order, and what ends at each
if args.grouped:     # labels of the grouped kernel (attn_mfma_kernel)
    order = [(0, "kernel start"), (7, "q + K codebook + K bytes of the ring requested"), (8, "K codebook written to LDS"),
             (1, "barrier 1 (K codebook)"), (2, "accumulators initialised"),
             (16, "score unit 0"), (17, "  issue: V bytes units 0-1"), (18, "score unit 1"),
             (19, "  issue: V bytes units 2-3"), (20, "score unit 2"), (21, "  issue: V codebook"), (22, "score unit 3"),
             (24, "residual tile scores, softmax update"), (12, "append store, V codebook -> LDS"),
             (13, "barrier 2"), (3, "residual tile values, value pass (+ later groups)"),
             (4, "wave-merge barrier"), (5, "wave merge done"), (10, "partial stored, drained, barrier"), (6, "flag raised; end (mergers: poll + merge of one head)")]
else:                # labels of the streaming kernel (attn_stream_kernel)
    order = [(0, "kernel start"), (7, "requested: page ids, q, both codebooks, residual tile, units 0-1"),
             (8, "codebooks written to LDS"), (1, "barrier (codebooks)"), (2, "residual tile done"),
             (16, "prologue: scores of unit 0 (+ unit 2 requests), softmax"),
             (17, "first whole round of 4 blocks (splits of >= 8 units per wave)"),
             (19, "blocks of the last whole round"),
             (3, "values of the last unit of the round (+ single units beyond the whole rounds)"),
             (4, "wave-merge barrier"), (5, "wave merge done"), (10, "partial stored, drained, barrier"), (6, "flag raised; end (mergers: poll + merge of one head)")]
s = last
t0 = s[:, :, 0].min()
print(f"workgroups {s.shape[0]}; kernel span (first start -> last end) per launch [us]: {[round(float(x), 2) for x in spans]}")
print("start skew of workgroups [us]: max %.2f" % ((s[:, 0, 0].max() - t0) / 100.0))
c_ = np.array(crit)
print("critical path, mean over the %d launches [us]: last flag raiser starts @%.2f, takes %.2f to its flag, the merge behind it ends %.2f later"
      " | start skew max %.2f; start -> flag over all workgroups: mean %.2f max %.2f" % ((len(crit),) + tuple(c_.mean(0))))
for name, sel in (("waves 0-3 (carry a residual tile at r=100, 32 splits)", s[:, :4, :].reshape(-1, NS)), ("waves 4-7", s[:, 4:, :].reshape(-1, NS))):
    print(f"--- {name}: mean time since that wave's start, and step duration [us]")
    prev = 0
    for sid, label in order[1:]:
        if (sel[:, sid] == 0).all():
            continue
        at = (sel[:, sid] - sel[:, 0]) / 100.0
        d = (sel[:, sid] - sel[:, prev]) / 100.0
        print(f"  @{at.mean():6.2f}  +{d.mean():5.2f} (min {d.min():5.2f} max {d.max():5.2f})  {label}")
        prev = sid
if args.raw:
    for name, sel in (("waves 0-3", s[:, :4, :].reshape(-1, NS)), ("waves 4-7", s[:, 4:, :].reshape(-1, NS))):
        rows = []
        for sid in range(1, NS):
            ok = sel[:, sid] != 0
            if ok.any():
                at = (sel[ok, sid] - sel[ok, 0]) / 100.0
                rows.append((at.mean(), sid, at.min(), at.max(), int(ok.sum())))
        print(f"--- raw stamps, {name}: id @mean (min..max) [waves]")
        for m_, sid, lo, hi, n in sorted(rows):
            print(f"   stamp {sid:2d} @{m_:6.2f} ({lo:5.2f}..{hi:5.2f}) [{n}]")
if (s[:, 0, 12] != 0).any():
    print("  workgroups that stored their partial plain (census: every split on their XCD): %d of %d; arrival indices seen: %s" % (
        int((s[:, 0, 12] == 2).sum()), s.shape[0], sorted(set((s[:, 0, 13] - 1).tolist()))[:6]))
la = s[:, 0, :][s[:, 0, 11] != 0]
if la.shape[0]:
    print("  mergers (%d): barrier -> every flag seen %.2f | -> head merged, out written %.2f us" % (
        la.shape[0], ((la[:, 11] - la[:, 10]) / 100.0).mean(), ((la[:, 6] - la[:, 11]) / 100.0).mean()))
    idxs = la[:, 13] - 1
    prim = la[idxs == idxs.max()]
    help_ = la[idxs != idxs.max()]
    for name, rows in (("primary (last arriver)", prim), ("helpers", help_)):
        if rows.shape[0]:
            print("    %-24s (%3d): barrier -> flags seen %.2f | -> end %.2f us (min %.2f max %.2f)" % (
                name, rows.shape[0], ((rows[:, 11] - rows[:, 10]) / 100.0).mean(), ((rows[:, 6] - rows[:, 11]) / 100.0).mean(),
                ((rows[:, 6] - rows[:, 11]) / 100.0).min(), ((rows[:, 6] - rows[:, 11]) / 100.0).max()))
e = (s[:, 0, 6] - t0) / 100.0
print("  end relative to first start: mean %.2f max %.2f us" % (e.mean(), e.max()))
