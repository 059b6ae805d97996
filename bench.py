#!/usr/bin/env python3
"""bench.py — decode tokens/s of the PQ-KV attention hot path at BASELINE.json's headline config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under torch.distributed.run with N
ranks (as a CHILD process, before anything has touched the GPU) and exits with the child's code, so the one-line
form works too.

Workload (BASELINE.json configs[2], "Llama-3.1-8B-hf (GQA), 32K context, PQ M=64 nbits=8, batch=1"; with N > 1 the
default becomes configs[3]: 2 requests per GPU, 16 requests over 8 GPUs): one STEP = one decode token through the hot
path of all 32 layers of the rank's requests — per layer: residual-window append of the new K/V row, flush of the
oldest 64 rows into a new K page and V page when the window is full (PQ encode), and the fused decode attention over
the PagedPQCache (T ~ 32K quantised tokens + residual window), nh=32 q heads, nh_k=8 kv heads, d=128.  Synthetic data
exactly as the reference's micro-benchmark (scripts/modeldb/bindings/test_kernel.py:59-65): q/residuals/centroids ~
N(0,1) fp16, codes ~ U{0..255}.  Each layer has its own 33.5 MB of code pages (1.07 GB per step), so the Infinity
Cache cannot hold the working set.  The model's projections / MLP are NOT part of this path (harness
glue is a "next" row, timed by tools/e2e_speedtest.py) — `value` is hot-path tokens/s, and is labelled so.

A step is replayed from a hipGraph (lengths live on the device), timed over exactly K steps between barrier +
synchronize on both sides; the window fill at the start is chosen so that ONE flush step (encode of a page per layer)
falls inside the timed region; N > 1 shards REQUESTS over ranks (no data-path collective: SURVEY.md 8e), value = all
ranks' tokens / max-over-ranks time.

Extra objects on the JSON line: `roofline` (dominant kernel = the fused decode-attention launch: algorithmic bytes
per launch / mean launch duration from HIP events recorded on the launch stream), `cpu_baseline` (the reference's CPU
PyTorch math restated in oracle/, timed on this box's cores on ONE layer-call and scaled to a step; rank 0, N = 1
only) and `gpu_fp16_baseline` (one layer-call of the reference's fp16 full-KV recipe — torch.cat + repeat_kv + SDPA,
scripts/modeldb/models/modeling_llama.py:403-443 — and of SDPA on a preallocated cache, same shape, HIP events);
`vs_baseline` = the preallocated-cache SDPA layer-call time / this path's launch time (kernel level; BASELINE.md holds
no published number for the metric itself; `vs_hf_recipe` is the same against the reference's own recipe),
`roofline.residual_tile` (MFMA flops of the residual-window tiles over their in-kernel phase time against the dense fp16
MFMA peak) and - N = 1 only, `--no-e2e` skips it - `e2e`: TPOT / TTFT of a Llama-3.1-8B-shaped random-weight decoder at
32K by the reference's definitions (speedtest.py:92-108) for the reference's fp16 recipe, a preallocated fp16 cache and
this repo's PQ path (hipGraph replay), with `speedup_vs_hf_baseline` = north_star's ">= 2.0x" quantity.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=32768, help="quantised context length T at the start")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="requests per rank (default: 1, or 2 when --gpus > 1 = BASELINE configs[3])")
    ap.add_argument("--M", type=int, default=64)
    ap.add_argument("--nh", type=int, default=32)
    ap.add_argument("--nh-k", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp16-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--force-generic", action="store_true", help="A/B: use the generic LUT kernel")
    ap.add_argument("--kernel-policy", type=int, default=0, help="A/B knob passed to million_set_force_generic")
    ap.add_argument("--roofline-launches", type=int, default=256)
    ap.add_argument("--flush-mode", default="encode-ahead", choices=["encode-ahead", "ahead", "inline"],
                    help="encode-ahead: the oldest window page is encoded (all layers, one launch on a side stream) in an earlier step and the "
                         "flush step only commits; ahead (round 3, A/B): the flush launches of the flush step on a side stream; inline (A/B): in front of each layer's attention")
    ap.add_argument("--flush-depth", type=int, default=2, help="layers the side-stream flush launches run ahead of the attention that needs them (A/B; 99 = all at the start of the step)")
    ap.add_argument("--ea-steps", type=int, default=0, help="A/B: decode steps the encode-ahead of all layers is spread over (0: the cache's choice)")
    ap.add_argument("--ea-launches", type=int, default=0, help="A/B: launches the encode-ahead of all layers is split into (0: the cache's choice)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end TPOT / TTFT record (N = 1 only; ~1-2 min)")
    ap.add_argument("--e2e-cap-s", type=float, default=240.0, help="no end-to-end backend is started later than this many seconds into the e2e leg")
    ap.add_argument("--dry-cpu", action="store_true",
                    help="rehearse the launcher / rank plumbing on CPU (gloo, no kernels): prints the same JSON line with "
                         "a sleep as the step; for the CPU tests, never a measurement")
    args = ap.parse_args(argv)
    if args.batch_per_gpu <= 0:
        args.batch_per_gpu = 2 if args.gpus > 1 else 1
    return args


def algorithmic_bytes(bs, nh, nh_k, T, r, d, M, C):
    """K+V code bytes once per kv head + residual rows + both codebooks + q in / out."""
    dm = d // M
    return 2 * bs * nh_k * T * M + 2 * bs * nh_k * r * d * 2 + 2 * M * C * dm * 2 + bs * nh * d * 2 * 2


def window_fill_at_start(cap, warmup, steps):
    """Residual-window fill r0 such that the first flush (the step that starts with r == cap) is step steps // 2 of the
    timed region.  The window holds cap rows and flushes 64 of them: r cycles through (cap - 64, cap]."""
    period = 64
    r0 = cap - warmup - steps // 2
    while r0 <= cap - period:
        r0 += period
    return max(1, min(cap, r0))


def relaunch_under_torchrun(args):
    """--gpus N > 1 outside a torchrun environment: start N ranks as a child process (this process has not touched the
    GPU and never will) and return its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd)


def device_uuid(torch, idx):
    try:
        return str(torch.cuda.get_device_properties(idx).uuid)
    except Exception:
        return f"cuda:{idx}"


def ranks_seen(dist, rank, ident):
    """All-gather of (rank, device identity) over the job's backend (RCCL on the GPU box): proves how many ranks ran."""
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, (rank, ident))
    return [list(x) for x in sorted(got)]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_cpu:
        return dry_cpu(args, torch, dist, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)     # RCCL; only barrier + max-reduce of the elapsed time
    seen = ranks_seen(dist, rank, device_uuid(torch, local_rank)) if world > 1 else [[0, device_uuid(torch, local_rank)]]

    from million_amd import ops, sharding
    from million_amd.pq_cache import PagedPQCache

    ops.set_force_generic(1 if args.force_generic else args.kernel_policy)
    bs, nh, nhk, d, M, C, layers = args.batch_per_gpu, args.nh, args.nh_k, 128, args.M, 256, args.layers
    ps, cap = 64, 128
    T0 = args.ctx // ps * ps
    # PRE untimed steps come before the caller's W warm-up steps whatever W is: one whole flush period, so that both
    # captured graphs (plain step, flush step) have been replayed once before anything is timed
    PRE = 72
    r0 = window_fill_at_start(cap, args.warmup + PRE, args.steps)
    total_steps = PRE + args.warmup + args.steps + 64 + 8
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

    cache.encode_ahead_launches = args.ea_launches or None
    cache.encode_ahead_steps = args.ea_steps or None

    def step_eager(use_dl):
        if args.flush_mode == "encode-ahead":
            cache.begin_step(use_dev_lengths=use_dl)      # oldest window page encoded ahead of its flush step / committed
        elif args.flush_mode == "ahead" and cache.next_step_flushes():
            cache.flush_ahead(use_dev_lengths=use_dl, depth=args.flush_depth)      # flushes on a side stream, joined layer by layer
        for l in range(layers):
            cache.decoding_with_pages(q[l], kn[l], vn[l], l, out=outs[l], use_dev_lengths=use_dl)

    # ---- capture two graphs: a plain step and a step whose layers flush first ----
    graphs = {}
    if not args.no_graph:
        st = cache.host_state()
        dl_backup = [t.clone() for t in cache.lengths]
        step_eager(True)                       # eager once: allocates the workspace, warms caches
        torch.cuda.synchronize()
        for name, state in cache.capture_states(st):
            if args.flush_mode != "encode-ahead" and (name.startswith("pre") or name == "commit"):
                continue
            cache.set_host_state(state)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step_eager(True)
            graphs[name] = gr
        cache.set_host_state(st)
        for t, b in zip(cache.lengths, dl_backup):
            t.copy_(b)
        torch.cuda.synchronize()

    def step_kind():
        if args.flush_mode == "encode-ahead":
            return cache.next_step_kind()
        return "flush" if cache.next_step_flushes() else "plain"

    def one_step():
        if args.no_graph:
            step_eager(False)
        else:
            kind = step_kind()
            graphs[kind].replay()
            cache.note_replayed_step(kind)

    for _ in range(PRE + args.warmup):
        one_step()
    n_flush_steps = 0

    def counted_step():
        nonlocal n_flush_steps
        n_flush_steps += int(cache.next_step_flushes())
        one_step()

    # exactly K steps between barrier + synchronize on both sides; MAX over ranks (million_amd/sharding.py)
    elapsed, elapsed_ranks = sharding.timed_steps_per_rank(counted_step, args.steps, torch.cuda.synchronize, dist if world > 1 else None)
    value, ms_per_step = sharding.aggregate_throughput(bs, args.steps, elapsed, world)
    flushes_timed = n_flush_steps
    # for the record: one whole flush period (64 steps = 63 plain + 1 flush), the steady-state mix
    elapsed64 = sharding.timed_steps(counted_step, 64, torch.cuda.synchronize, dist if world > 1 else None)
    value64, ms64 = sharding.aggregate_throughput(bs, 64, elapsed64, world)
    # and the two kinds of step apart (HIP events around every step of another period): what a flush step costs
    step_ev, step_is_flush, step_kinds = [], [], []
    for _ in range(64):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        step_is_flush.append(cache.next_step_flushes())
        step_kinds.append(step_kind())
        a.record()
        one_step()
        b.record()
        step_ev.append((a, b))
    torch.cuda.synchronize()
    step_ms = [a.elapsed_time(b) for a, b in step_ev]
    plain_ms = [t for t, f in zip(step_ms, step_is_flush) if not f]
    flush_ms = [t for t, f in zip(step_ms, step_is_flush) if f]
    plain_step_ms = sorted(plain_ms)[len(plain_ms) // 2] if plain_ms else None
    flush_step_ms = sum(flush_ms) / len(flush_ms) if flush_ms else None
    pre_ms = [t for t, k in zip(step_ms, step_kinds) if k.startswith("pre")]
    pre_step_ms = sum(pre_ms) / len(pre_ms) if pre_ms else None
    plain_only = [t for t, k in zip(step_ms, step_kinds) if k == "plain"]
    if plain_only:
        plain_step_ms = sorted(plain_only)[len(plain_only) // 2]

    # ---- roofline of the dominant kernel: HIP events on the launch stream ----
    # `achieved` uses the average launch duration over a region of nl back-to-back launches between ONE pair of events
    # (= what rocprofv3 --kernel-trace reports as the kernel's average, profiles/rNN_kernel_stats.csv, within ~1 %).
    # Event pairs around every single launch are reported too; they read 1.5-2 us higher on a ~20 us kernel (the cost
    # of the two event records).  Everything is enqueued behind a long device-side sleep so that the timestamps
    # measure GPU execution, not host enqueue latency.
    T_now, r_now = cache._T[0], cache.residualed_tokens[0]
    nl = max(layers, args.roofline_launches // layers * layers)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nl)]
    kprep, vprep = cache._kprep, cache._vprep

    def attn_launch(l):
        ops.pq_decode_attn(q[l], cache.key_page_pool, cache.value_page_pool, kprep, vprep,
                           cache.key_residual_cache[l], cache.value_residual_cache[l], r_now, M=M, C=C,
                           n_tokens=T_now, resid_start=cache._rstart[l], k_page_ids=cache.page_ids[l],
                           v_page_ids=cache.page_ids[l], page_size=ps, out=outs[l], workspace=cache._ws)

    for rep in range(2):                       # first pass warms, second is measured
        torch.cuda._sleep(int(2.0e8 if rep else 1.0e7))
        for i in range(nl):
            evs[i][0].record()
            attn_launch(i % layers)
            evs[i][1].record()
        torch.cuda.synchronize()
    durs = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)          # seconds
    mean_dur = sum(durs) / len(durs)
    med_dur = durs[len(durs) // 2]
    # the same launches back to back between ONE pair of events: launch-to-launch period, no per-launch event cost
    periods = []
    for _ in range(3):                         # three regions; the median is reported, all three are in the line
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(2.0e8))
        e0.record()
        for i in range(nl):
            attn_launch(i % layers)
        e1.record()
        torch.cuda.synchronize()
        periods.append(e0.elapsed_time(e1) * 1e-3 / nl)
    period_eager = sorted(periods)[1]
    # the same nl launches as ONE captured hipGraph, replayed: how the steps themselves are launched (round 4; VERDICT r03 item 2:
    # rocprofv3 inside replayed steps and the eager region above disagreed by 15 % at 8 requests - the cause is the device-side
    # sleep in front of the eager region, not the launch mode: tools/mode_probe.py).  `frac` uses this region; the eager region
    # stays in the line as a second field.  tools/trace_phases.py splits a rocprofv3 --kernel-trace of this very command into
    # the same phases (the sleeps are the separators; the three graph regions follow the last sleep-separated phase).
    periods_graph = []
    if not args.no_graph:
        gr_roof = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr_roof):
            for i in range(nl):
                attn_launch(i % layers)
        gr_roof.replay()                       # warm replay; the timed ones follow it DIRECTLY: a device-side sleep in front of a
        torch.cuda.synchronize()               # region lets the chip drop its clocks and the ~20 ms behind it read 3-15 % slow
        for _ in range(3):                     # (tools/mode_probe.py; that is what the eager regions above still do)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr_roof.replay()
            e1.record()
            torch.cuda.synchronize()
            periods_graph.append(e0.elapsed_time(e1) * 1e-3 / nl)
    period = sorted(periods_graph)[1] if periods_graph else period_eager
    alg = algorithmic_bytes(bs, nh, nhk, T_now, r_now, d, M, C)
    achieved = alg / period / 1e9
    # every rank's own launch period (the line's roofline is rank 0's; a slow GPU shows here and in per_rank_ms_per_step)
    period_ranks = sharding.gather_floats(period, dist if world > 1 else None)
    # a split merge that gave up on a partial writes NaN heads and counts itself (million_hip.h): must be 0 after the timed regions
    tail_faults = sharding.gather_floats(float(ops.tail_faults()), dist if world > 1 else None)
    traffic = None
    pmc = ROOT / "profiles" / "pmc_traffic.json"     # written from rocprofv3 --pmc passes (see profiles/README.md)
    if pmc.exists():
        try:
            rec = json.loads(pmc.read_text())
            if rec.get("ctx") == T0 and rec.get("M") == M and rec.get("batch_per_gpu") == bs:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    fp16 = None
    if not args.no_fp16_baseline and not args.force_generic:
        fp16 = gpu_fp16_baseline(torch, dev, bs, nh, nhk, d, T_now + r_now)
    res_tile = None
    if not args.force_generic and M in (32, 64):
        res_tile = residual_tile_record(torch, attn_launch, layers, bs, nh, nhk, d, r_now, T_now)

    if rank == 0:
        cfg_name = "configs[2]" if (world == 1 and bs == 1) else ("configs[3]" if bs == 2 else "configs[2] shape")
        line = {
            "metric": "decode tokens/sec @32K ctx, Llama-3.1-8B PQ-KV attention hot path (32 layers), 1xMI355X per request",
            "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(fp16["preallocated_sdpa_us"] / (period * 1e6), 2) if fp16 else None),
            "dtype": "f16 in/out, u8 codes, f32 accumulate", "data": "synthetic",
            "per_rank_ms_per_step": [round(1e3 * e / args.steps, 4) for e in elapsed_ranks],
            "tail_faults": int(sum(tail_faults)),
            "config": {"workload": f"BASELINE {cfg_name}: Llama-3.1-8B shape (32 layers, nh={nh}, nh_k={nhk}, d=128), "
                                   f"ctx {T0}, PQ M={M} nbits=8, PagedPQCache page 64 / window 128, batch {bs}/GPU; "
                                   "step = 32 x (append + flush-when-full encode + fused decode attention)",
                       "ctx": T0, "layers": layers, "M": M, "batch_per_gpu": bs, "parallelism": f"requests x{world}",
                       "ranks_seen": seen, "window_fill_at_start": r0,
                       "weak_scaling_base": ("this line" if world == 1 else
                                             f"per-GPU batch {bs} (BASELINE configs[3] = 16 requests over 8 GPUs): the like-for-like "
                                             f"1-GPU base is `bench.py --gpus 1 --batch-per-gpu {bs}`, not the default configs[2] line"),
                       "flush_steps_in_timed_region": flushes_timed,
                       "flush": {"encode-ahead": "encode-ahead: the oldest window page of ALL layers is encoded by one launch on a side stream "
                                                 f"when the window holds {cache.encode_ahead_at()} rows (beside that step's attention launches); the flush "
                                                 "step only moves the lengths (one launch for all layers)",
                                 "ahead": f"ahead: flush launches of the flush step on a side stream, issued {args.flush_depth} layers ahead; each layer's attention waits for its own",
                                 "inline": "in-line"}[args.flush_mode],
                       "steady_state_64_steps": {"value": round(value64, 2), "ms_per_step": round(ms64, 4),
                                                 "note": "one whole flush period (64 steps: plain steps, one encode-ahead step, one flush / commit step)"},
                       "plain_step_ms": round(plain_step_ms, 4) if plain_step_ms else None,
                       "flush_step_ms": round(flush_step_ms, 4) if flush_step_ms else None,
                       "flush_step_over_plain_step": round(flush_step_ms / plain_step_ms, 3) if plain_step_ms and flush_step_ms else None,
                       "encode_ahead_steps_per_period": len(pre_ms),
                       "encode_ahead_step_ms": round(pre_step_ms, 4) if pre_step_ms else None,
                       "encode_ahead_step_over_plain_step": round(pre_step_ms / plain_step_ms, 3) if plain_step_ms and pre_step_ms else None,
                       "launch": "eager" if args.no_graph else "hipGraph replay",
                       "kernel": "generic-LUT" if args.force_generic else ("auto" if not args.kernel_policy else f"policy{args.kernel_policy}")},
            "roofline": {"bound": "hbm", "kernel": "fused decode attention (one launch per layer-call)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command)" if traffic else None,
                         "algorithmic_bytes_per_launch": alg, "launch_us_mean": round(period * 1e6, 2),
                         "launches_timed": nl,
                         "per_rank": {"launch_us_mean": [round(x * 1e6, 2) for x in period_ranks],
                                      "frac": [round(alg / x / 1e9 / HBM_PEAK_GBS, 4) for x in period_ranks]},
                         "timing": ("HIP events around ONE replay of a captured hipGraph of the launches / launches (how the steps are launched); "
                                    "median of three regions" if periods_graph else
                                    "HIP events around a region of back-to-back eager launches / launches; median of three regions"),
                         "launch_us_regions": [round(x * 1e6, 2) for x in (periods_graph or periods)],
                         "eager": {"launch_us_mean": round(period_eager * 1e6, 2), "launch_us_regions": [round(x * 1e6, 2) for x in periods],
                                   "frac": round(alg / period_eager / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "rounds 1-3 method: the same launches enqueued one by one behind a device-side sleep (keeps the host ahead, but the "
                                           "chip lowers its clocks during the sleep: reads 3 % slow at 1 request, 15 % at 8; profiles/r04_shapes.txt)"},
                         "event_pair_per_launch_us_mean": round(mean_dur * 1e6, 2),
                         "event_pair_per_launch_us_median": round(med_dur * 1e6, 2)},
        }
        if fp16:
            line["vs_baseline_note"] = ("kernel level, no published number in BASELINE.md: layer-call time of torch SDPA over a "
                                        "PREALLOCATED fp16 KV cache (no cat, no repeat_kv) / fused PQ decode-attention launch time; "
                                        "vs_hf_recipe = the same against the reference's own baseline recipe (torch.cat + repeat_kv + SDPA, "
                                        "modeling_llama.py:403-443); the end-to-end ratio by the reference's TPOT definition is e2e.speedup_vs_hf_baseline")
            line["vs_hf_recipe"] = round(fp16["hf_recipe_us"] / (period * 1e6), 2)
            line["gpu_fp16_baseline"] = fp16
        if res_tile:
            line["roofline"]["residual_tile"] = res_tile
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(bs, nh, nhk, d, M, C, T0, r0, layers)
        if world == 1 and not args.no_e2e and not args.force_generic and bs == 1 and M == 64 and layers == 32:
            # free the hot-path state first: the end-to-end model brings 16 GB of weights and its own caches
            del cache, graphs, q, kn, vn, outs
            torch.cuda.empty_cache()
            line["e2e"] = e2e_record(T0, args.e2e_cap_s)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if sum(tail_faults):
        raise SystemExit(f"bench.py: {int(sum(tail_faults))} split merges gave up waiting for a partial (NaN heads were written): "
                         "the numbers above are not a measurement; zero the workspace (million_workspace_init) and look for a dead workgroup")


def gpu_fp16_baseline(torch, dev, bs, nh, nhk, d, T):
    """One layer-call of the fp16 full-KV attention at the same shape, HIP-event timed: (a) the reference's baseline
    recipe (modeling_llama.py:403-443: DynamicCache.update = torch.cat of the new row, repeat_kv, SDPA) and (b) SDPA on a
    preallocated cache with the G query heads of a kv head as query rows (no cat, no repeat).  Four rotating layers of
    K/V so that the 134 MB per layer do not sit in the Infinity Cache."""
    import torch.nn.functional as F
    G = nh // nhk
    nrot = 4
    Ks = [torch.randn(bs, nhk, T, d, device=dev, dtype=torch.float16) for _ in range(nrot)]
    Vs = [torch.randn(bs, nhk, T, d, device=dev, dtype=torch.float16) for _ in range(nrot)]
    q = torch.randn(bs, nh, 1, d, device=dev, dtype=torch.float16)
    kn = torch.randn(bs, nhk, 1, d, device=dev, dtype=torch.float16)
    vn = torch.randn(bs, nhk, 1, d, device=dev, dtype=torch.float16)

    def repeat_kv(x):
        b, h, t, dd = x.shape
        return x[:, :, None, :, :].expand(b, h, G, t, dd).reshape(b, h * G, t, dd)

    def hf(i):
        k = torch.cat([Ks[i % nrot], kn], dim=2)
        v = torch.cat([Vs[i % nrot], vn], dim=2)
        return F.scaled_dot_product_attention(q, repeat_kv(k), repeat_kv(v))

    def static(i):
        qq = q.view(bs, nhk, G, d)
        return F.scaled_dot_product_attention(qq, Ks[i % nrot], Vs[i % nrot]).view(bs, nh, 1, d)

    res = {}
    for name, fn, n in (("hf_recipe_us", hf, 24), ("preallocated_sdpa_us", static, 48)):
        for _ in range(4):
            fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) * 1e3 / n, 2)
    res["shape"] = f"bs={bs} nh={nh} nh_k={nhk} d={d} T={T} fp16 K/V ({2 * bs * nhk * T * d * 2 / 1e6:.1f} MB per layer)"
    res["fp16_kv_bytes_per_layer_call"] = 2 * bs * nhk * T * d * 2
    return res


def residual_tile_record(torch, attn_launch, layers, bs, nh, nhk, d, r, T):
    """north_star: "MFMA utilisation on the residual tile against chip peak".  The window's rows are 16-row MFMA tiles
    inside the fused launch (one per wave that owns rows); the time of that phase comes from the kernel's diagnostic
    stamps (lane 0 of each wave, 100 MHz counter: stamp 1 = codebook barrier passed, stamp 2 = residual tile done), taken
    over `layers` launches with the stamp buffer set.  Two flop counts: what the MFMAs ISSUE (whole 16 x 16 x 32 / 32 x 32
    x 16 tiles: 8 head columns and 16 rows whatever G and the row count) and what is USEFUL (r rows x d x G heads x 2 for
    q.K^T and again for p.V per kv head).  Peak: 2.5 PFLOP/s dense fp16 (MI355X_MICROARCH.md).  It is tiny by
    construction - the tile is latency-, not throughput-bound - and is reported for the record."""
    from million_amd import _lib as L
    lib = L.load()
    NW, NS = 8, 32
    n_wg = 256 * 2
    stamps = torch.zeros(n_wg * NW * NS, dtype=torch.int64, device="cuda")
    us, waves_with_tile, n_wgs = [], 0, 0
    try:
        lib.million_debug_set_stamp_buffer(stamps.data_ptr())
        for l in range(min(layers, 8)):
            stamps.zero_()
            attn_launch(l)
            torch.cuda.synchronize()
            s = stamps.cpu().numpy().reshape(-1, NW, NS)
            s = s[s[:, 0, 0] != 0]
            if not s.shape[0]:
                continue
            d12 = (s[:, :, 2] - s[:, :, 1]) / 100.0                   # us per wave
            us.append(float(d12[:, 0].mean()))                        # wave 0 carries a split's rows (runs of 16: one tile at r ~ 100 over 32 splits)
            n_wgs = s.shape[0]
    finally:
        lib.million_debug_set_stamp_buffer(None)
    if not us:
        return None
    G = nh // nhk
    nsplit = max(1, n_wgs // (bs * nhk))
    rows_per_split = -(-r // nsplit)
    waves_with_tile = min(8, -(-rows_per_split // 16)) * n_wgs         # a split's rows go to its waves in runs of 16 (attn_mfma.hip: kResRows)
    issued = waves_with_tile * (4 * 16 * 16 * 32 * 2 + 4 * 32 * 32 * 16 * 2)
    useful = bs * nhk * r * d * G * 2 * 2
    t = sum(us) / len(us)
    peak = 2.5e15
    return {"phase_us_per_wave": round(t, 3), "waves_with_a_tile": waves_with_tile,
            "mfma_flops_issued_per_launch": issued, "useful_flops_per_launch": useful,
            "tflops_issued": round(issued / (t * 1e-6) / 1e12, 2), "tflops_useful": round(useful / (t * 1e-6) / 1e12, 3),
            "frac_of_dense_fp16_mfma_peak_issued": round(issued / (t * 1e-6) / peak, 5),
            "frac_of_dense_fp16_mfma_peak_useful": round(useful / (t * 1e-6) / peak, 6), "peak_tflops": 2500.0,
            "note": "phase = stamp 1 -> 2 of the waves that own window rows (all tiles of the launch run concurrently); "
                    "latency-bound by construction: 8 MFMAs per wave behind the loads of 16 fp16 rows"}


def e2e_record(ctx, cap_s):
    """North-star target 1 (>= 2.0x decode tokens/s over the fp16 full-KV baseline at 32K, Llama-3.1-8B shape) by the
    reference's own definitions (scripts/benchmarks/speedtest.py:92-108: 10 generated tokens = 9 timed decode intervals,
    1 warm-up + 5 timed runs; TTFT = the prompt pass): million_amd.harness.speedtest on a random-fp16-weight decoder
    (the reference's `_synthetic` mode), a real 32K-token prompt pass for TTFT.  hf_baseline = the reference's baseline
    recipe (torch.cat + repeat_kv + SDPA, modeling_llama.py:403-443); static_fp16 = SDPA on a preallocated cache;
    pq_graph = this repo's PagedPQCache path, the decode step replayed from hipGraphs."""
    import traceback
    from million_amd import harness as H
    t0 = time.time()
    try:
        res = H.speedtest(ctx=ctx, decode=10, niter=5, bs=1, model="llama31_8b",
                          backends=("hf_baseline", "pq_graph", "static_fp16", "pq_eager"), prefill=True, ttft_iters=1,
                          deadline=t0 + cap_s)
    except Exception as e:      # the hot-path line must survive a failure of the glue around it
        return {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-600:]}
    out = {"definition": "speedtest.py:92-108 (TPOT: mean of 9 decode intervals over 5 runs after 1 warm-up; TTFT: prompt pass to first token on the host)",
           "model": "Llama-3.1-8B shape, 32 layers, random fp16 weights, batch 1", "ctx": ctx, "wall_s": round(time.time() - t0, 1)}
    for k in ("hf_baseline", "static_fp16", "pq_graph", "pq_eager"):
        if k in res:
            out[k] = res[k]
    pq = res.get("pq_graph", {})
    out["speedup_vs_hf_baseline"] = pq.get("speedup_vs_hf_baseline")
    out["speedup_vs_static_fp16"] = pq.get("speedup_vs_static_fp16")
    out["ttft_vs_hf_baseline"] = pq.get("ttft_vs_hf_baseline")
    return out


def dry_cpu(args, torch, dist, world, rank):
    """Launcher rehearsal on CPU (tests/test_host_logic.py): same rank plumbing, barrier, max-over-ranks timing and JSON
    line, with a sleep as the step and gloo as the backend.  Not a measurement; nothing under oracle/ or the HIP library
    is involved."""
    from million_amd import sharding
    if world > 1:
        dist.init_process_group("gloo")
    seen = ranks_seen(dist, rank, f"cpu-rank{rank}") if world > 1 else [[0, "cpu-rank0"]]
    bs = args.batch_per_gpu

    def step():
        time.sleep(0.002 * (1 + rank))
    elapsed, elapsed_ranks = sharding.timed_steps_per_rank(step, args.steps, lambda: None, dist if world > 1 else None)
    value, ms = sharding.aggregate_throughput(bs, args.steps, elapsed, world)
    period_ranks = sharding.gather_floats(20e-6 * (1 + rank), dist if world > 1 else None)      # stands for each rank's launch period
    n = dist.get_world_size() if world > 1 else 1
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (cpu, gloo): launcher rehearsal, not a measurement", "value": round(value, 2),
                          "unit": "tokens/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "none",
                          "per_rank_ms_per_step": [round(1e3 * e / args.steps, 4) for e in elapsed_ranks],
                          "roofline": {"per_rank": {"launch_us_mean": [round(x * 1e6, 2) for x in period_ranks],
                                                    "frac": [round(34.0e6 / x / 1e9 / HBM_PEAK_GBS, 4) for x in period_ranks]}},
                          "config": {"workload": "dry-cpu", "batch_per_gpu": bs, "parallelism": f"requests x{n}",
                                     "ranks_seen": seen,
                                     "window_fill_at_start": window_fill_at_start(128, args.warmup, args.steps)}}),
              flush=True)
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
                      f"= {per_layer * 1e3:.1f} ms, scaled x{layers} layers",
            "encode": cpu_baseline_encode(bs, nhk, d, M, C, cores)}


def cpu_baseline_encode(bs, nhk, d, M, C, cores):
    """the encode leg on the host's cores - the reference's CPU-runnable sa_encode_4d (torch.cdist + argmin,
    pq_utils.py:410-449; oracle.pq_encode_cdist_torch) and the direct form of its GPU encoder (pq_utils.py:483-494;
    oracle.pq_encode_direct_torch), fp32, on (i) one layer's flush page (page_size 64 rows x nh_k heads x requests, K and V
    sides) and (ii) a 1024-token prompt slice of one request (K and V sides).  Bounded (~10-20 s in all): at most 32 torch
    threads - on the 256-core GPU box these small dense ops ran 20-200 x SLOWER with one thread per core (85 s for one 4096-token
    slice) than on 8 cores - and `cores` says how many were used."""
    import torch
    from oracle import oracle as O
    used = max(1, min(cores, 32))
    torch.set_num_threads(used)
    g = torch.Generator().manual_seed(11)
    cents = torch.randn(M, C, d // M, generator=g).half().float().numpy()
    n_prompt = 1024
    out = {"unit": "ms per layer, K and V sides", "cores": used, "dtype": "f32"}
    for name, n, nb in (("flush_page_64_rows", 64, bs), (f"prompt_{n_prompt}_tokens", n_prompt, 1)):
        X = torch.randn(nb, nhk, n, d, generator=g).half().float().numpy()
        cell = {}
        for form, fn in (("cdist", O.pq_encode_cdist_torch), ("direct", O.pq_encode_direct_torch)):
            fn(X[:, :, :min(n, 64)], cents)      # warm-up
            t0, reps = time.perf_counter(), 0
            while reps < 1 or (time.perf_counter() - t0 < 1.5 and reps < 10):
                fn(X, cents)
                reps += 1
            cell[form] = round(2.0 * (time.perf_counter() - t0) / reps * 1e3, 3)      # x 2: K side and V side
        out[name] = cell
    out["rows_per_s_direct_prompt"] = round(2 * nhk * n_prompt / (out[f"prompt_{n_prompt}_tokens"]["direct"] * 1e-3), 1)
    torch.set_num_threads(cores)
    return out


if __name__ == "__main__":
    main()
