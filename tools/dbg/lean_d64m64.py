#!/usr/bin/env python3
"""Development helper: the d = 64 / M = 64 lean form against the oracle, a few (T, r) mixes; prints per-head errors."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from million_amd import ops  # noqa: E402
import synth  # noqa: E402
from oracle import oracle  # noqa: E402
oracle.build()

C = 256
for (d, M, T, r, nh, nhk, same) in ((64, 64, 64, 0, 4, 1, 0), (64, 64, 64, 1, 4, 1, 0), (64, 64, 17, 128, 8, 2, 0), (64, 64, 5000, 64, 32, 8, 0), (64, 64, 40000, 100, 32, 8, 0)):
    c = synth.attn_case(4242 + T + r, 1, nh, nhk, d, M, C, T, r, Lt=128)
    if same == 1: c['q'][:, 1:] = c['q'][:, :1]
    if same == 2: c['q'][:, :, :, 1:] = 0; c['q'][:, :, :, 0] = np.arange(1, nh + 1, dtype=np.float16)[None, :, None]
    gold = np.asarray(oracle.decode_attn(**c), dtype=np.float64)
    t = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    kp = ops.prepare_cents(t["k_cents"], cache=False); vp = ops.prepare_cents(t["v_cents"], cache=False)
    for pol in (0, 16):
        ops.set_force_generic(pol)
        out = ops.pq_decode_attn(t["q"], t["k_codes"], t["v_codes"], kp, vp, t["k_res"], t["v_res"], c["r"], M=M, C=C)
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(o - gold) / max(np.linalg.norm(gold), 1e-30)
        print(f"d={d} M={M} T={T} r={r} nh={nh} nhk={nhk} policy {pol}: rel {rel:.3e}  |out| {np.linalg.norm(o):.3f} |gold| {np.linalg.norm(gold):.3f}")
        if pol == 0 and rel > 1e-3:
            err = np.abs(o - gold)[0, :, 0, :]
            print("   per head max err:", np.round(err.max(axis=1), 3))
            print("   head 0 per 8-dim block max err:", np.round(err[0].reshape(8, 8).max(axis=1), 3))
            print("   head 1 per 8-dim block max err:", np.round(err[min(1, nh - 1)].reshape(8, 8).max(axis=1), 3))
            for hh in range(1, min(nh, 4)):
                oh = o[0, hh, 0]
                best = [(float(np.linalg.norm(oh - gold[0, g2, 0]) / np.linalg.norm(gold[0, g2, 0])), g2) for g2 in range(min(nh, 4))]
                print(f"   out head {hh}: rel to gold heads", [(round(a, 3), b) for a, b in best], " out[:6]", np.round(oh[:6], 3), "gold[:6]", np.round(gold[0, hh, 0, :6], 3))
            h = 0
            print("   gold[0,0,0,:8]", np.round(gold[0, h, 0, :8], 4)); print("   out [0,0,0,:8]", np.round(o[0, h, 0, :8], 4))
            # does out hold gold's dims in another order?
            g0, o0 = gold[0, h, 0], o[0, h, 0]
            idx = [int(np.argmin(np.abs(g0 - x))) for x in o0[:16]]
            print("   nearest gold dim of out dims 0..15:", idx)
    ops.set_force_generic(0)
