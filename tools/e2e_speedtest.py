#!/usr/bin/env python3
"""End-to-end decode TPOT on a Llama-shaped random-weight model: fp16 full-KV baselines vs the PQ path.

Restates the reference's speed test (scripts/benchmarks/speedtest.py:85-108): one warm-up generation, then
`niter` timed ones; every generated token is handed to the host (the reference's streamer), the wall-clock
interval between consecutive tokens is recorded, and TPOT = sum(intervals[1:]) / (dl - 1).  The prompt is
not run: caches are filled synthetically at `--ctx` tokens (the reference's `_synthetic` loader does the same
for weights), so intervals[0] here is simply the first decode step and is dropped like the reference drops
the prefill interval.

    python tools/e2e_speedtest.py --ctx 32768 --decode 64 --out gpurun_out/e2e.json
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def tpot_from(intervals, dl):
    return sum(intervals[1:]) / (dl - 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", type=int, default=32768)
    ap.add_argument("--decode", type=int, default=64, help="decoding_length (tokens generated per run)")
    ap.add_argument("--niter", type=int, default=2)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--model", default="llama31_8b", choices=["llama31_8b", "llama2_7b"])
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--backends", default="hf_baseline,static_fp16,pq_eager,pq_graph")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    from million_amd import harness as H

    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    dev = torch.device("cuda", 0)
    shape = getattr(H.LlamaShape, args.model)()
    if args.layers:
        shape.n_layers = args.layers
    model = H.LlamaShapeDecoder(shape, dev)
    bs, dl = args.bs, args.decode
    max_new = (args.niter + 1) * dl + 8
    results = {"config": {"model": args.model, "ctx": args.ctx, "decoding_length": dl, "niter": args.niter, "bs": bs,
                          "layers": shape.n_layers, "weights": "random fp16", "tpot": "speedtest.py:104 definition"}}

    def run_generation(step_fn):
        """dl tokens; returns the inter-token wall-clock intervals in ms (host receives every token)."""
        torch.cuda.synchronize()
        ivals, t_prev = [], time.perf_counter()
        for _ in range(dl):
            step_fn().cpu()                    # streamer.put(token) -> host
            t = time.perf_counter()
            ivals.append((t - t_prev) * 1e3)
            t_prev = t
        return ivals

    def measure(step_fn):
        run_generation(step_fn)                # warm-up generation (speedtest.py:92)
        tp = [tpot_from(run_generation(step_fn), dl) for _ in range(args.niter)]
        return sum(tp) / len(tp)

    for name in args.backends.split(","):
        torch.cuda.empty_cache()
        tokens = torch.zeros(bs, dtype=torch.long, device=dev)
        pos = torch.full((bs,), args.ctx, dtype=torch.long, device=dev)
        if name == "hf_baseline":
            be = H.HFBaselineCache(shape, bs, args.ctx, dev)
        elif name == "static_fp16":
            be = H.StaticFP16Cache(shape, bs, args.ctx, max_new, dev)
        elif name in ("pq_eager", "pq_graph"):
            be = H.PQBackend(shape, bs, args.ctx, max_new, dev)
        else:
            raise SystemExit(f"unknown backend {name}")

        def eager_step():
            nxt = model.step(tokens, pos, be)
            tokens.copy_(nxt)
            pos.add_(1)
            return tokens

        if name != "pq_graph":
            step_fn = eager_step
        else:
            graphed = H.GraphedPQDecoder(model, be, tokens, pos)
            step_fn = graphed.step

        tp = measure(step_fn)
        results[name] = {"tpot_ms": round(tp, 4), "tokens_per_s": round(bs * 1e3 / tp, 2)}
        print(name, results[name], flush=True)
        del be, step_fn
        graphed = None

    base = results.get("hf_baseline", {}).get("tpot_ms")
    stat = results.get("static_fp16", {}).get("tpot_ms")
    for k in ("pq_eager", "pq_graph"):
        if k in results:
            if base:
                results[k]["speedup_vs_hf_baseline"] = round(base / results[k]["tpot_ms"], 3)
            if stat:
                results[k]["speedup_vs_static_fp16"] = round(stat / results[k]["tpot_ms"], 3)
    line = json.dumps(results)
    print(line)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
