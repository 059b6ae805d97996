#!/usr/bin/env python3
"""bench.py — decode tokens/s of the PQ-KV attention hot path at BASELINE.json's headline config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "Llama-3.1-8B-hf (GQA), 32K context, PQ M=64 nbits=8, batch=1"):
one STEP = one decode token through the hot path of all 32 layers of one request per GPU — per layer:
residual-window append of the new K/V row, flush of the oldest 64 rows into a new K page and V page when
the window is full (PQ encode), and the fused decode attention over the PagedPQCache (T ~ 32K quantised
tokens + residual window), nh=32 q heads, nh_k=8 kv heads, d=128.  Synthetic data exactly as the reference's
micro-benchmark (scripts/modeldb/bindings/test_kernel.py:59-65): q/residuals/centroids ~ N(0,1) fp16, codes
~ U{0..255}.  Each layer has its own 33.5 MB of code pages (1.07 GB per step), so the Infinity Cache cannot
hold the working set.  The model's projections / MLP are NOT part of this path (harness glue
is a "next" row) — `value` is hot-path tokens/s, and is labelled so.

A step is replayed from a hipGraph (lengths live on the device), timed over exactly K steps between
barrier + synchronize on both sides; N > 1 shards REQUESTS over ranks (one request per GPU, no data-path
collective: SURVEY.md 8e), value = all ranks' tokens / max-over-ranks time.

Extra objects on the JSON line: `roofline` (dominant kernel = the fused decode-attention launch:
algorithmic bytes per launch / mean launch duration from HIP events recorded on the launch stream) and
`cpu_baseline` (the reference's CPU PyTorch math restated in oracle/, timed on this box's cores on ONE
layer-call and scaled to a step; rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=32768, help="quantised context length T at the start")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--M", type=int, default=64)
    ap.add_argument("--nh", type=int, default=32)
    ap.add_argument("--nh-k", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--force-generic", action="store_true", help="A/B: use the generic LUT kernel")
    ap.add_argument("--kernel-policy", type=int, default=0, help="A/B: 2 = grouped MFMA kernel only, 3 = prefer the pipelined one")
    ap.add_argument("--roofline-launches", type=int, default=256)
    return ap.parse_args()


def algorithmic_bytes(bs, nh, nh_k, T, r, d, M, C):
    """K+V code bytes once per kv head + residual rows + both codebooks + q in / out."""
    dm = d // M
    return 2 * bs * nh_k * T * M + 2 * bs * nh_k * r * d * 2 + 2 * M * C * dm * 2 + bs * nh * d * 2 * 2


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)     # RCCL; only barrier + max-reduce of the elapsed time

    from million_amd import ops, sharding
    from million_amd.pq_cache import PagedPQCache

    ops.set_force_generic(1 if args.force_generic else args.kernel_policy)
    bs, nh, nhk, d, M, C, layers = args.batch_per_gpu, args.nh, args.nh_k, 128, args.M, 256, args.layers
    ps, cap = 64, 128
    T0 = args.ctx // ps * ps
    r0 = 100                                   # window fill at the start: the first flush falls inside the run
    total_steps = args.warmup + args.steps + 4
    g = torch.Generator(device="cpu").manual_seed(42 + rank)
    cache = PagedPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=layers, d=d, page_size=ps,
                         extended_residual_size=cap, max_tokens=T0 + total_steps + 2 * cap, device=dev)
    cents_k = torch.randn(M, C, d // M, generator=g).half().to(dev)
    cents_v = torch.randn(M, C, d // M, generator=g).half().to(dev)
    cache.set_cent(cents_k, cents_v)
    # synthetic state: random codes in every page, random residual rows (same distributions as test_kernel.py)
    cache.key_page_pool.copy_(torch.randint(0, C, cache.key_page_pool.shape, dtype=torch.uint8, device=dev))
    cache.value_page_pool.copy_(torch.randint(0, C, cache.value_page_pool.shape, dtype=torch.uint8, device=dev))
    for l in range(layers):
        cache.key_residual_cache[l].copy_(torch.randn(cache.key_residual_cache[l].shape, device=dev).half())
        cache.value_residual_cache[l].copy_(torch.randn(cache.value_residual_cache[l].shape, device=dev).half())
    cache.set_host_state(([T0 + r0] * layers, [r0] * layers, [T0] * layers, [0] * layers))
    for l in range(layers):
        cache._sync_lengths(l)
    q = [torch.randn(bs, nh, 1, d, device=dev).half() for _ in range(layers)]
    kn = [torch.randn(bs, nhk, 1, d, device=dev).half() for _ in range(layers)]
    vn = [torch.randn(bs, nhk, 1, d, device=dev).half() for _ in range(layers)]
    outs = [torch.empty(bs, nh, 1, d, device=dev, dtype=torch.float16) for _ in range(layers)]

    def step_eager(use_dl):
        for l in range(layers):
            cache.decoding_with_pages(q[l], kn[l], vn[l], l, out=outs[l], use_dev_lengths=use_dl)

    # ---- capture two graphs: a plain step and a step whose layers flush first ----
    graphs = {}
    if not args.no_graph:
        st = cache.host_state()
        dl_backup = [t.clone() for t in cache.lengths]
        step_eager(True)                       # eager once: allocates the workspace, warms caches
        torch.cuda.synchronize()
        for name, r_cap in (("plain", r0), ("flush", cap)):
            cache.set_host_state((st[0], [r_cap] * layers, st[2], st[3]))
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step_eager(True)
            graphs[name] = gr
        cache.set_host_state(st)
        for t, b in zip(cache.lengths, dl_backup):
            t.copy_(b)
        torch.cuda.synchronize()

    def one_step():
        if args.no_graph:
            step_eager(False)
        else:
            graphs["flush" if cache.next_step_flushes() else "plain"].replay()
            cache.note_replayed_step()

    for _ in range(args.warmup):
        one_step()
    n_flush_steps = 0

    def counted_step():
        nonlocal n_flush_steps
        n_flush_steps += int(cache.next_step_flushes())
        one_step()

    # exactly K steps between barrier + synchronize on both sides; MAX over ranks (million_amd/sharding.py)
    elapsed = sharding.timed_steps(counted_step, args.steps, torch.cuda.synchronize, dist if world > 1 else None)
    value, ms_per_step = sharding.aggregate_throughput(bs, args.steps, elapsed, world)

    # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream ----
    # All (event, launch, event) triplets are enqueued behind a long device-side sleep so that the
    # timestamps measure GPU execution, not host enqueue latency.
    T_now, r_now = cache._T[0], cache.residualed_tokens[0]
    nl = max(layers, args.roofline_launches // layers * layers)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nl)]
    kprep, vprep = cache._kprep, cache._vprep
    for rep in range(2):                       # first pass warms, second is measured
        torch.cuda._sleep(int(2.0e8 if rep else 1.0e7))
        for i in range(nl):
            l = i % layers
            evs[i][0].record()
            ops.pq_decode_attn(q[l], cache.key_page_pool, cache.value_page_pool, kprep, vprep,
                               cache.key_residual_cache[l], cache.value_residual_cache[l], r_now, M=M, C=C,
                               n_tokens=T_now, resid_start=cache._rstart[l], k_page_ids=cache.page_ids[l],
                               v_page_ids=cache.page_ids[l], page_size=ps, out=outs[l], workspace=cache._ws)
            evs[i][1].record()
        torch.cuda.synchronize()
    durs = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)          # seconds
    mean_dur = sum(durs) / len(durs)
    med_dur = durs[len(durs) // 2]
    alg = algorithmic_bytes(bs, nh, nhk, T_now, r_now, d, M, C)
    achieved = alg / mean_dur / 1e9
    traffic = None
    pmc = ROOT / "profiles" / "pmc_traffic.json"     # written from rocprofv3 --pmc passes (see profiles/README.md)
    if pmc.exists():
        try:
            rec = json.loads(pmc.read_text())
            if rec.get("ctx") == T0 and rec.get("M") == M and rec.get("batch_per_gpu") == bs:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "decode tokens/sec @32K ctx, Llama-3.1-8B PQ-KV attention hot path (32 layers), 1xMI355X per request",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 in/out, u8 codes, f32 accumulate", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: Llama-3.1-8B shape (32 layers, nh=32, nh_k=8, d=128), "
                                   f"ctx {T0}, PQ M={M} nbits=8, PagedPQCache page 64 / window 128, batch {bs}/GPU; "
                                   "step = 32 x (append + flush-when-full encode + fused decode attention)",
                       "ctx": T0, "layers": layers, "M": M, "batch_per_gpu": bs, "parallelism": f"requests x{world}",
                       "flush_steps_in_timed_region": n_flush_steps,
                       "launch": "eager" if args.no_graph else "hipGraph replay",
                       "kernel": "generic-LUT" if args.force_generic else ("auto" if not args.kernel_policy else f"policy{args.kernel_policy}")},
            "roofline": {"bound": "hbm", "kernel": "fused decode attention (one launch per layer-call)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg, "launch_us_mean": round(mean_dur * 1e6, 2),
                         "launch_us_median": round(med_dur * 1e6, 2), "launches_timed": nl},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(bs, nh, nhk, d, M, C, T0, r0, layers)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(bs, nh, nhk, d, M, C, T, r, layers):
    """The reference's CPU PyTorch math (oracle/oracle.py:decode_attn_torch_cpu) on this box's cores:
    ONE layer-call of the same workload, scaled by the layer count."""
    import torch
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(7)
    q = torch.randn(bs, nh, 1, d, generator=g).half()
    kc = torch.randint(0, C, (bs, nhk, T, M), dtype=torch.uint8, generator=g)
    vc = torch.randint(0, C, (bs, nhk, T, M), dtype=torch.uint8, generator=g)
    ck, cv = torch.randn(M, C, d // M, generator=g).half(), torch.randn(M, C, d // M, generator=g).half()
    kr, vr = torch.randn(bs, nhk, 128, d, generator=g).half(), torch.randn(bs, nhk, 128, d, generator=g).half()
    O.decode_attn_torch_cpu(q, kc[:, :, :1024], vc[:, :, :1024], ck, cv, kr, vr, r)      # warm-up
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or (time.perf_counter() - t0 < 10.0 and reps < 8):
        O.decode_attn_torch_cpu(q, kc, vc, ck, cv, kr, vr, r)
        reps += 1
    per_layer = (time.perf_counter() - t0) / reps
    return {"value": round(bs / (per_layer * layers), 4), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x one layer-call (T={T}, fp32 torch-CPU restatement of sa_decode_4d + cat + SDPA) "
                      f"= {per_layer * 1e3:.1f} ms, scaled x{layers} layers"}


if __name__ == "__main__":
    main()
