#!/usr/bin/env python3
"""Development helper: the per-phase clock sums of the pipelined prompt-attention kernel (a -DMILLION_EXP=2048 build, tools/ab_build.py
2048): workgroup 0 (kv head 0, the LAST query block: the longest) writes each wave's eight sums into the wave's own first query row.
    MILLION_HIP_LIB=build/ab/libmillion_exp2048.so python tools/prefill_prof.py [n]"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
policy = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 128: the experimental 4-wave x 64-row kernel (4 waves per workgroup)
NW = 4 if policy == 128 else 8
dev = torch.device("cuda", 0)
nh, nhk, d = 32, 8, 128
G = nh // nhk
hpw = 4 if G % 4 == 0 else 2 if G % 2 == 0 else 1      # heads per workgroup as prefill.hip picks them (largest of 8, 4, 2, 1 dividing G)
wph = NW // hpw
QB = wph * (64 if NW == 4 else 32)
n_qb = (n + QB - 1) // QB
q = torch.randn(1, nh, n, d, device=dev).half(); k = torch.randn(1, nhk, n, d, device=dev).half(); v = torch.randn(1, nhk, n, d, device=dev).half()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.set_force_generic(policy)
a.record(); ops.prefill_attn(q, k, v); b.record(); torch.cuda.synchronize()
print(f"n={n}: {a.elapsed_time(b):.3f} ms (first call, profiled build); workgroup 0 = kv head 0, query block {n_qb - 1} ({(n_qb - 1) * QB}..), {2 * ((n + 63) // 64)} half-steps per wave")
names = ["DMA issue", "phase 1 (QK | exp)", "mask", "phase 2 (PV | max)", "decision", "DMA wait + barrier", "-", "prologue"]
tot = np.zeros(8)
for w in range(NW):
    head = w // wph
    row = (n_qb - 1) * QB + (w % wph) * (64 if NW == 4 else 32)
    vals = q[0, head, row, :16].cpu().numpy().view(np.uint32).astype(np.float64)
    tot += vals
    print(f"  wave {w} (head {head}, rows {row}..): " + "  ".join(f"{int(x):>9d}" for x in vals) + f"   sum {int(vals.sum())}")
print("  mean over waves, share of the wave's time:")
for i, nm in enumerate(names):
    if tot[i]:
        print(f"    {nm:22s} {tot[i] / NW:12.0f} cycles  {100 * tot[i] / tot.sum():5.1f} %   per half-step {tot[i] / NW / (2 * ((n + 63) // 64)):8.1f}")
