#!/usr/bin/env python3
"""End-to-end TTFT / TPOT / per-section breakdown on a Llama-shaped random-weight model: fp16 full-KV baselines vs
the PQ path.

Restates the reference's speed test (scripts/benchmarks/speedtest.py:85-117): one warm-up generation, then `niter`
timed ones; every generated token is handed to the host (the reference's streamer), the wall-clock interval between
consecutive tokens is recorded, TPOT = sum(intervals[1:]) / (dl - 1) (:104) and TTFT = intervals[0] (:105), the prompt
pass.  Without --prefill the prompt is not run: caches are filled synthetically at `--ctx` tokens (the reference's
`_synthetic` loader does the same for weights) and intervals[0] is just the first decode step.  With --prefill a
random prompt of `--ctx` tokens really goes through the model (q/k/v projections, RoPE, the backend's prefill: for PQ
the bulk encode of all layers into pages + causal SDPA on the fp16 prompt, pq_utils.py:222-260; MLP), which is what
the reference's time_to_first_token measures.  --breakdown adds the reference's per-section timers (Timer.py;
speedtest.py:110-117): cumulative seconds per section over all attention calls of the timed decode steps, each
section closed by a device synchronise.

    python tools/e2e_speedtest.py --ctx 32768 --decode 64 --prefill --breakdown --out gpurun_out/e2e.json
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
    ap.add_argument("--prefill", action="store_true", help="run a real prompt of --ctx tokens through the model: TTFT")
    ap.add_argument("--breakdown", action="store_true", help="per-section cumulative timers over one extra generation")
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
    max_new = (args.niter + 2) * dl + 16
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

    def make_backend(name, filled):
        if name == "hf_baseline":
            return H.HFBaselineCache(shape, bs, args.ctx if filled else 0, dev)
        if name == "static_fp16":
            return H.StaticFP16Cache(shape, bs, args.ctx, max_new, dev)
        if name in ("pq_eager", "pq_graph"):
            return H.PQBackend(shape, bs, args.ctx, max_new, dev, synthetic_fill=filled)
        raise SystemExit(f"unknown backend {name}")

    prompt = torch.randint(0, shape.vocab, (bs, args.ctx), device=dev) if args.prefill else None
    for name in args.backends.split(","):
        torch.cuda.empty_cache()
        tokens = torch.zeros(bs, dtype=torch.long, device=dev)
        pos = torch.full((bs,), args.ctx, dtype=torch.long, device=dev)
        rec = {}
        use_prefill = args.prefill and name != "static_fp16"      # (the preallocated baseline has no prompt pass of its own)
        if use_prefill:
            ttft = []
            for it in range(args.niter + 1):                       # first one is the warm-up (speedtest.py:92)
                be = None
                torch.cuda.empty_cache()
                be = make_backend(name, filled=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                first = model.prefill(prompt, be)
                first.cpu()                                        # the first token reaches the host: intervals[0]
                if it:
                    ttft.append((time.perf_counter() - t0) * 1e3)
            rec["ttft_ms"] = round(sum(ttft) / len(ttft), 2)
            tokens.copy_(first)
        else:
            be = make_backend(name, filled=True)

        def eager_step():
            nxt = model.step(tokens, pos, be)
            tokens.copy_(nxt)
            pos.add_(1)
            return tokens

        if args.breakdown and name != "pq_graph":                 # one generation with the section timers on (eager only)
            tm = H.SectionTimers()
            model.timers = be.timers = tm
            for _ in range(dl):
                eager_step().cpu()
            model.timers = be.timers = H.NO_TIMERS
            rec["breakdown_times_s"] = {k: round(v, 5) for k, v in sorted(tm.seconds.items())}
            rec["breakdown_calls"] = dict(sorted(tm.calls.items()))

        if name != "pq_graph":
            step_fn = eager_step
        else:
            graphed = H.GraphedPQDecoder(model, be, tokens, pos)
            step_fn = graphed.step

        tp = measure(step_fn)
        rec.update({"tpot_ms": round(tp, 4), "tokens_per_s": round(bs * 1e3 / tp, 2)})
        results[name] = rec
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
            bt = results.get("hf_baseline", {}).get("ttft_ms")
            if bt and "ttft_ms" in results[k]:
                results[k]["ttft_vs_hf_baseline"] = round(results[k]["ttft_ms"] / bt, 3)
    line = json.dumps(results)
    print(line)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
