#!/usr/bin/env python3
"""Development probe: why do the attention launches inside bench.py's replayed steps take ~60 us at 8
requests while the same shape timed as plain back-to-back launches takes ~67-70 us?  Times ONE captured graph of 32 x R launches
(HIP events around a replay, after a warm replay) for the call forms that differ between the two:
    plain          million_pq_decode_attn, host lengths, n_tokens = T                      (bench.py's roofline region, ab_bench)
    append         ..._append (new K/V row attended to and stored), host lengths
    devlen         plain + device-resident lengths, n_tokens = the cache's bound
    append+devlen  the step's own form (PagedPQCache.decoding_with_pages); r is restored inside the graph
    plain-bound    plain with n_tokens = T but the PAGE TABLE sliced / not sliced ... (see code)
    python tools/mode_probe.py --bs 8
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from million_amd import ops  # noqa: E402
from million_amd.pq_cache import PagedPQCache  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=8)
ap.add_argument("--ctx", type=int, default=32768)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--r", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda", 0)
bs, nh, nhk, d, M, C, ps, cap, layers = a.bs, 32, 8, 128, 64, 256, 64, 128, a.layers
T0 = a.ctx
cache = PagedPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=layers, d=d, page_size=ps, extended_residual_size=cap,
                     max_tokens=T0 + 600, device=dev)
g = torch.Generator(device="cpu").manual_seed(1)
cache.set_cent(torch.randn(M, C, d // M, generator=g).half().to(dev), torch.randn(M, C, d // M, generator=g).half().to(dev))
cache.key_page_pool.copy_(torch.randint(0, C, cache.key_page_pool.shape, dtype=torch.uint8, device=dev))
cache.value_page_pool.copy_(torch.randint(0, C, cache.value_page_pool.shape, dtype=torch.uint8, device=dev))
for l in range(layers):
    cache.key_residual_cache[l].copy_(torch.randn(cache.key_residual_cache[l].shape, device=dev).half())
    cache.value_residual_cache[l].copy_(torch.randn(cache.value_residual_cache[l].shape, device=dev).half())
cache.set_host_state(([T0 + a.r] * layers, [a.r] * layers, [T0] * layers, [0] * layers))
for l in range(layers):
    cache._sync_lengths(l)
q = [torch.randn(bs, nh, 1, d, device=dev).half() for _ in range(layers)]
kn = [torch.randn(bs, nhk, 1, d, device=dev).half() for _ in range(layers)]
vn = [torch.randn(bs, nhk, 1, d, device=dev).half() for _ in range(layers)]
outs = [torch.empty(bs, nh, 1, d, device=dev, dtype=torch.float16) for _ in range(layers)]
cache.decoding_with_pages(q[0], kn[0], vn[0], 0, out=outs[0], use_dev_lengths=False)      # allocates the workspace
cache.set_host_state(([T0 + a.r] * layers, [a.r] * layers, [T0] * layers, [0] * layers))
for l in range(layers):
    cache._sync_lengths(l)
saved = [t.clone() for t in cache.lengths]
kp, vp = cache._kprep, cache._vprep


def call(l, append, devlen, bound=None):
    ops.pq_decode_attn(q[l], cache.key_page_pool, cache.value_page_pool, kp, vp, cache.key_residual_cache[l],
                       cache.value_residual_cache[l], a.r, M=M, C=C,
                       n_tokens=(bound or (cache.max_tokens if devlen else T0)), resid_start=0,
                       k_page_ids=cache.page_ids[l], v_page_ids=cache.page_ids[l], page_size=ps, out=outs[l],
                       dev_lengths=cache.lengths[l] if devlen else None, workspace=cache._ws,
                       **(dict(k_new=kn[l], v_new=vn[l]) if append else {}))


forms = {
    "plain": lambda l: call(l, False, False),
    "append": lambda l: call(l, True, False),
    "devlen": lambda l: call(l, False, True),
    "plain, n_tokens = bound": lambda l: call(l, False, False, bound=cache.max_tokens),
    "append+devlen (step form)": lambda l: call(l, True, True),
}
graphs = {}
for name, fn in forms.items():
    for l in range(layers):
        fn(l)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for rep in range(a.reps):
            for l in range(layers):
                fn(l)
            if "devlen" in name and "append" in name:
                for t, s in zip(cache.lengths, saved):      # r += 1 per launch on the device: put it back
                    t.copy_(s)
    graphs[name] = gr
for t, s in zip(cache.lengths, saved):
    t.copy_(s)
torch.cuda.synchronize()
n = a.reps * layers
print(f"bs={bs} T={T0} r={a.r}: one graph of {n} launches per form, us per launch (three rounds, forms interleaved)")
res = {k: [] for k in graphs}
for rnd in range(4):
    for name, gr in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            res[name].append(e0.elapsed_time(e1) * 1e3 / n)
for name, v in res.items():
    print(f"  {name:28s} " + "  ".join(f"{x:7.2f}" for x in v))
# the plain form launched EAGERLY (one dispatch packet per launch, enqueued behind a device sleep so that the host is ahead)
ev = []
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(1.0e8))
    e0.record()
    for rep in range(a.reps):
        for l in range(layers):
            forms["plain"](l)
    e1.record()
    torch.cuda.synchronize()
    ev.append(e0.elapsed_time(e1) * 1e3 / n)
print(f"  {'plain, eager launches':28s} " + "  ".join(f"{x:7.2f}" for x in ev))
ev = []
for rnd in range(3):      # and the graph form behind the same sleep (does the idle period before a region matter?)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(1.0e8))
    e0.record()
    graphs["plain"].replay()
    e1.record()
    torch.cuda.synchronize()
    ev.append(e0.elapsed_time(e1) * 1e3 / n)
print(f"  {'plain, graph after a sleep':28s} " + "  ".join(f"{x:7.2f}" for x in ev))
