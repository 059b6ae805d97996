#!/usr/bin/env python3
"""Development helper: time the fused decode-attention launch of ONE library build at several shapes (back-to-back
launch period over rotating layers, HIP events) and compare its output with the grouped kernel's.
    MILLION_HIP_LIB=build/ab/libmillion_exp1.so python tools/ab_bench.py [--cfg bs,T,M[,d] ...] [--vs-scalar]"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import _lib as L, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cfg", nargs="*", default=["1,32768,64", "2,32768,64", "4,32768,64", "1,131072,64", "1,131072,32", "1,32832,64"])
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--nh", type=int, default=32)
ap.add_argument("--nh-k", type=int, default=8)
ap.add_argument("--C", type=int, default=256)
ap.add_argument("--shuffle-pages", action="store_true", help="random page permutation (default: ids in allocation order, as PagedPQCache hands them out)")
ap.add_argument("--iters", type=int, default=96)
ap.add_argument("--dev-lengths", action="store_true")
ap.add_argument("--vs-scalar", action="store_true", help="also time the scalar fallback kernel (million_set_force_generic(1)) on the same call")
ap.add_argument("--bindings-10arg", action="store_true",
                help="time the reference's 10-argument call (bindings.flash_decoding_allocated_buffer_*: row-major K and V "
                     "codes, the same tensors on every call of a layer) instead of the paged call")
ap.add_argument("--zero-codes", action="store_true", help="diagnostic: all code bytes 0 (every LDS gather is a broadcast: no bank conflicts)")
ap.add_argument("--k-conflict-free", action="store_true", help="diagnostic: K codes chosen so that the 64 lanes of every K gather hit 64 different LDS banks (M = 64 streaming kernel)")
ap.add_argument("--eager-after-sleep", action="store_true", help="rounds 1-3 timing: eager launches enqueued behind a device-side sleep (reads slow: the clocks drop during the sleep)")
ap.add_argument("--policy", type=int, default=0, help="million_set_force_generic value for the TIMED launches (16: the lean kernel's shapes stay on the streaming kernel)")
ap.add_argument("--same-page", action="store_true", help="diagnostic: every page id = 0 (codes come from L2, not HBM)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
lib = L.load()
tag = os.environ.get("MILLION_HIP_LIB", "in-tree")
for cfg in args.cfg:
    f = [int(x) for x in cfg.split(",")]
    bs, T, M = f[:3]
    d = f[3] if len(f) > 3 else 128
    nh, nhk, C, ps, r = args.nh, args.nh_k, args.C, 64, 100
    n_pages = (T + ps - 1) // ps
    states = []
    nl = max(4, min(args.layers, int(6e9 // (2 * bs * nhk * n_pages * ps * M))))      # keep the pools under ~6 GB
    for l in range(nl):
        kpool = torch.randint(0, C, (bs * nhk * n_pages, ps, M), dtype=torch.uint8, device=dev)
        vpool = torch.randint(0, C, (bs * nhk * n_pages, M, ps), dtype=torch.uint8, device=dev)
        ids = (torch.randperm(bs * nhk * n_pages, device=dev) if args.shuffle_pages else torch.arange(bs * nhk * n_pages, device=dev)).to(torch.int32).reshape(bs, nhk, n_pages)
        if args.zero_codes:
            kpool.zero_(); vpool.zero_()
        if args.k_conflict_free:      # code(t, m) = row of t in its score tile + 16 * (lane group of m): bank = code % 64
            t = torch.arange(ps, device=dev)
            row = ((t >> 3) << 2) | (t & 3)
            grp = (torch.arange(M, device=dev) >> 2) & 3
            kpool.copy_((row[:, None] % 16 + 16 * grp[None, :]).to(torch.uint8).expand_as(kpool))
        if args.same_page:
            ids.zero_()
        states.append((kpool, vpool, ids))
    kc = ops.prepare_cents(torch.randn(M, C, d // M, device=dev).half(), cache=False)
    vc = ops.prepare_cents(torch.randn(M, C, d // M, device=dev).half(), cache=False)
    q = torch.randn(bs, nh, 1, d, device=dev).half()
    kr = torch.randn(bs, nhk, 128, d, device=dev).half()
    vr = torch.randn(bs, nhk, 128, d, device=dev).half()
    dl = torch.tensor([[T, r, 0, 0]] * bs, dtype=torch.int32, device=dev) if args.dev_lengths else None

    if args.bindings_10arg:
        import bindings
        fn = getattr(bindings, f"flash_decoding_allocated_buffer_f16u8_Ns32Lt128d128M{M}C{C}")
        kcent, vcent = torch.randn(M, C, d // M, device=dev).half(), torch.randn(M, C, d // M, device=dev).half()
        rm = [(torch.randint(0, 256, (bs, nhk, T, M), dtype=torch.uint8, device=dev),
               torch.randint(0, 256, (bs, nhk, T, M), dtype=torch.uint8, device=dev)) for _ in range(len(states))]
        po = torch.empty(bs, nh, 33, d, dtype=torch.float16, device=dev)
        pl = torch.empty(bs, nh, 33, dtype=torch.float16, device=dev)

    def run(l):
        if args.bindings_10arg:
            kcod, vcod = rm[l % len(rm)]
            return fn(q, kcod, vcod, kcent, vcent, kr, vr, r, po, pl)
        kp, vp, ids = states[l % len(states)]
        return ops.pq_decode_attn(q, kp, vp, kc, vc, kr, vr, r, M=M, C=C, n_tokens=T, k_page_ids=ids, v_page_ids=ids,
                                  page_size=ps, dev_lengths=dl)

    lib.million_set_force_generic(2 if (d == 128 and M in (32, 64)) else 1)        # grouped kernel (scalar kernel off the streaming shapes) as the cross-check
    ref = run(0).float()
    lib.million_set_force_generic(args.policy)
    out = run(0).float()
    err = ((out - ref).norm() / ref.norm()).item()
    if not (err < 1e-2):      # say where: (request, head) pairs that differ, and whether a merge gave up on a flag
        bad = ((out - ref).abs().amax(-1) > 1e-2) | out.isnan().any(-1)
        faults = lib.million_debug_tail_faults() if hasattr(lib, "million_debug_tail_faults") else -1
        print(f"  MISMATCH: {int(bad.sum())} of {bad.numel()} (request, head) rows differ; tail faults {faults}; rows {bad.nonzero().flatten().tolist()[:64]}", flush=True)
    def timed(iters):
        # ONE captured graph of the launches, replayed; the timed replay follows a warm replay directly.  (Through round 3 the
        # region was enqueued eagerly behind a device-side sleep, to keep the host ahead: the chip drops its clocks during the
        # sleep and the ~20 ms region behind it ran 3 % (1 request) to 15 % (8 requests) slow - tools/mode_probe.py,
        # profiles/r04_shapes.txt.  --eager-after-sleep keeps that method for comparison.)
        for i in range(16):
            run(i)
        torch.cuda.synchronize()
        best = 1e9
        if args.eager_after_sleep or args.bindings_10arg:
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(int(5e7))
                e0.record()
                for i in range(iters):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
            return best
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(iters):
                run(i)
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        return best

    kind = lib.million_attn_kernel_kind(__import__('ctypes').byref(ops.make_attn_desc(q, kr, nh_k=nhk, M=M, C=C, n_tokens=T, r=r, k_paged=True, v_paged=True, page_size=ps, n_pages_cap=n_pages)))
    best = timed(args.iters)
    scalar = None
    if args.vs_scalar:
        lib.million_set_force_generic(1)
        scalar = timed(max(4, args.iters // 8))
        lib.million_set_force_generic(args.policy)
    alg = 2 * bs * nhk * T * M + 2 * bs * nhk * r * d * 2 + 2 * M * C * (d // M) * 2 + bs * nh * d * 4
    if args.bindings_10arg:
        tag = tag + " 10-arg"
    extra = f"   scalar fallback {scalar:8.1f} us ({scalar / best:5.1f}x)" if scalar else ""
    if args.policy and "policy" not in tag:
        tag = tag + f" policy {args.policy}"
    print(f"{tag:34s} bs={bs} nh={nh} nh_k={nhk} T={T:6d} d={d} M={M} C={C} kind={kind}: {best:6.2f} us/launch  {alg / best / 1e3:7.1f} GB/s ({alg / best / 8e6 * 100:4.1f}% of 8 TB/s)  rel diff vs cross-check {err:.1e}{extra}", flush=True)
    del states
    torch.cuda.empty_cache()
