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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctx", type=int, default=32768)
    ap.add_argument("--decode", type=int, default=64, help="decoding_length (tokens generated per run)")
    ap.add_argument("--niter", type=int, default=2)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--model", default="llama31_8b", choices=["llama31_8b", "llama2_7b"])
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--M", type=int, default=64, help="PQ subspaces (BASELINE configs[4]: 32)")
    ap.add_argument("--backends", default="hf_baseline,static_fp16,pq_eager,pq_graph")
    ap.add_argument("--prefill", action="store_true", help="run a real prompt of --ctx tokens through the model: TTFT")
    ap.add_argument("--breakdown", action="store_true", help="per-section cumulative timers over one extra generation")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    from million_amd import harness as H

    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    results = H.speedtest(ctx=args.ctx, decode=args.decode, niter=args.niter, bs=args.bs, model=args.model, layers=args.layers,
                          backends=tuple(args.backends.split(",")), prefill=args.prefill, breakdown=args.breakdown,
                          ttft_iters=args.niter, M=args.M, log=lambda m: print(m, flush=True))
    line = json.dumps(results)
    print(line)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
