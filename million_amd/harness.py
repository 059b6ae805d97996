"""Self-contained Llama-shaped decoder for end-to-end decode timing (SURVEY.md 8f-2 "harness glue").

The reference patches HF's LlamaSdpaAttention.forward (scripts/modeldb/models/modeling_llama.py:455-554
attn_forward_custom_kernel, :345-453 baseline_forward) inside `model.generate`; the installed transformers
no longer has that class, and no weights exist offline, so this is a plain-torch restatement of the decode
step of a Llama block with random fp16 weights (the reference's `_synthetic` mode, main_pq.py:252-255,
speedtest.py:31-33).  Everything outside attention (RMSNorm, q/k/v/o projections, RoPE, SwiGLU MLP,
lm_head, greedy argmax) is torch / hipBLASLt — plumbing, identical for every attention backend:

  * "hf_baseline"  — the reference's fp16 full-KV baseline recipe: DynamicCache-style torch.cat append,
                     repeat_kv, scaled_dot_product_attention (modeling_llama.py:403-443);
  * "static_fp16"  — the same attention on a preallocated cache, GQA without materialising repeat_kv
                     (a stronger fp16 baseline, for the record);
  * "pq"           — PagedPQCache.decoding_with_pages: flush-if-full + fused append + PQ attention (this repo).
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass

import torch
import torch.nn.functional as F


class SectionTimers:
    """The reference's --breakdown mode (scripts/utils/Timer.py:69-75, pq_utils.py:285-326, modeling_llama.py:403-443):
    named wall-clock sections inside the attention forward, each closed by a device synchronise, accumulated over all
    calls.  `with timers("sdpa"): ...`; timers.seconds -> {name: cumulative s}, timers.calls -> {name: count}."""

    def __init__(self, sync=None):
        self.seconds, self.calls = {}, {}
        self._sync = sync or (torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None))

    def __call__(self, name):
        return _Section(self, name)

    def reset(self):
        self.seconds.clear()
        self.calls.clear()


class _Section:
    def __init__(self, owner, name):
        self.o, self.name = owner, name

    def __enter__(self):
        self.o._sync()
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        self.o._sync()
        self.o.seconds[self.name] = self.o.seconds.get(self.name, 0.0) + time.perf_counter() - self.t0
        self.o.calls[self.name] = self.o.calls.get(self.name, 0) + 1
        return False


class _NoTimers:
    class _S:
        def __enter__(self):
            return None

        def __exit__(self, *exc):
            return False

    _s = _S()

    def __call__(self, name):
        return self._s


NO_TIMERS = _NoTimers()


@dataclass
class LlamaShape:
    hidden: int = 4096
    n_layers: int = 32
    nh: int = 32
    nh_k: int = 8
    d: int = 128
    inter: int = 14336
    vocab: int = 128256
    rope_theta: float = 500000.0
    eps: float = 1e-5

    @staticmethod
    def llama31_8b():
        return LlamaShape()

    @staticmethod
    def llama2_7b():
        return LlamaShape(nh_k=32, inter=11008, vocab=32000, rope_theta=10000.0)


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    """transformers.models.llama.modeling_llama.repeat_kv: (b, nh_k, T, d) -> (b, nh_k*n_rep, T, d), materialised."""
    b, h, t, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None, :, :].expand(b, h, n_rep, t, d).reshape(b, h * n_rep, t, d)


class HFBaselineCache:
    """fp16 full-KV cache grown by torch.cat (HF DynamicCache.update), + repeat_kv + SDPA."""

    def __init__(self, shape: LlamaShape, bs: int, ctx: int, device):
        """ctx > 0: caches filled synthetically with ctx tokens; ctx = 0: empty, to be filled by prefill()."""
        self.G = shape.nh // shape.nh_k
        mk = lambda: torch.randn(bs, shape.nh_k, ctx, shape.d, device=device, dtype=torch.float16)
        self.k = [mk() for _ in range(shape.n_layers)]
        self.v = [mk() for _ in range(shape.n_layers)]
        self.timers = NO_TIMERS

    def prefill(self, layer, q, k, v):
        """baseline_forward with q_len > 1 (modeling_llama.py:403-443): DynamicCache.update, repeat_kv, causal SDPA."""
        with self.timers("cat"):
            self.k[layer] = torch.cat([self.k[layer], k], dim=2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=2)
        with self.timers("repeat_kv"):
            kk, vv = repeat_kv(self.k[layer], self.G), repeat_kv(self.v[layer], self.G)
        with self.timers("sdpa"):
            return F.scaled_dot_product_attention(q, kk, vv, is_causal=True)

    def attend(self, layer, q, k, v):
        with self.timers("cat"):
            self.k[layer] = torch.cat([self.k[layer], k], dim=2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=2)
        with self.timers("repeat_kv"):
            kk, vv = repeat_kv(self.k[layer], self.G), repeat_kv(self.v[layer], self.G)
        with self.timers("sdpa"):
            return F.scaled_dot_product_attention(q, kk, vv)


class StaticFP16Cache:
    """Preallocated fp16 cache; GQA handled by viewing the G query heads of a kv head as G query rows."""

    def __init__(self, shape: LlamaShape, bs: int, ctx: int, max_new: int, device):
        self.shape, self.T = shape, ctx
        mk = lambda: torch.randn(bs, shape.nh_k, ctx + max_new, shape.d, device=device, dtype=torch.float16)
        self.k = [mk() for _ in range(shape.n_layers)]
        self.v = [mk() for _ in range(shape.n_layers)]
        self.timers = NO_TIMERS

    def attend(self, layer, q, k, v):
        s = self.shape
        T = self.T
        with self.timers("copy"):
            self.k[layer][:, :, T:T + 1] = k
            self.v[layer][:, :, T:T + 1] = v
        bs = q.shape[0]
        qg = q.view(bs, s.nh_k, s.nh // s.nh_k, s.d)
        with self.timers("sdpa"):
            out = F.scaled_dot_product_attention(qg, self.k[layer][:, :, :T + 1], self.v[layer][:, :, :T + 1])
        if layer == s.n_layers - 1:
            self.T += 1
        return out.view(bs, s.nh, 1, s.d)


class PQBackend:
    def __init__(self, shape: LlamaShape, bs: int, ctx: int, max_new: int, device, M=64, synthetic_fill=True):
        """synthetic_fill: the code pages hold ctx random tokens (decode-only timing); otherwise the cache starts empty and
        prefill() encodes a real prompt into it (attn_forward_custom_kernel's q_len > 1 branch, modeling_llama.py:540-543)."""
        from .pq_cache import PagedPQCache
        ps, cap = 64, 128
        T0 = ctx // ps * ps if synthetic_fill else 0
        self.cache = PagedPQCache(bs=bs, nh=shape.nh, num_key_value_heads=shape.nh_k, M=M, layer_num=shape.n_layers,
                                  d=shape.d, page_size=ps, extended_residual_size=cap, max_tokens=ctx + max_new + 2 * cap,
                                  device=device)
        g = torch.Generator(device="cpu").manual_seed(5)
        self.cache.set_cent(torch.randn(M, 256, shape.d // M, generator=g).half().to(device),
                            torch.randn(M, 256, shape.d // M, generator=g).half().to(device))
        if synthetic_fill:
            gd = torch.Generator(device=device).manual_seed(11)
            for pool in (self.cache.key_page_pool, self.cache.value_page_pool):
                pool.copy_(torch.randint(0, 256, pool.shape, dtype=torch.uint8, device=device, generator=gd))
            L = shape.n_layers
            self.cache.set_host_state(([T0] * L, [0] * L, [T0] * L, [0] * L))      # window empty after prefill
            for l in range(L):
                self.cache._sync_lengths(l)
        self.use_dev_lengths = False
        self.flush_ahead = True
        self.timers = NO_TIMERS

    def prefill(self, layer, q, k, v):
        with self.timers("prefill_encode+sdpa"):
            return self.cache.prefill(q, k, v, layer)

    def begin_step(self):
        """Start of a decode step: the cache encodes the oldest window page ahead of its flush step / commits it
        (PagedPQCache.begin_step; not with the section timers on: they want the flush as a section of its own)."""
        if self.flush_ahead and self.timers is NO_TIMERS:
            self.cache.begin_step(use_dev_lengths=self.use_dev_lengths)

    def attend(self, layer, q, k, v):
        if self.timers is not NO_TIMERS and self.cache.residualed_tokens[layer] >= self.cache.extended_residual_size:
            with self.timers("flush_encode"):      # the flush that decoding_with_pages would do first (pq_utils.py:288-302)
                self.cache.flush_to_pages(layer, use_dev_lengths=self.use_dev_lengths)
        with self.timers("kernel"):                # window append + fused PQ attention (pq_utils.py:304-326): one launch
            return self.cache.decoding_with_pages(q, k, v, layer, use_dev_lengths=self.use_dev_lengths)


class LlamaShapeDecoder:
    """Random-weight Llama decoder; `step(tokens, pos, backend)` runs one decode step and returns argmax ids."""

    def __init__(self, shape: LlamaShape, device, seed=0):
        self.s, self.device = shape, device
        g = torch.Generator(device=device).manual_seed(seed)
        h, s = shape.hidden, shape

        def w(n_out, n_in):
            return (torch.randn(n_out, n_in, generator=g, device=device, dtype=torch.float16) / math.sqrt(n_in))

        self.embed = torch.randn(s.vocab, h, generator=g, device=device, dtype=torch.float16)
        self.layers = []
        for _ in range(s.n_layers):
            self.layers.append(dict(wqkv=w((s.nh + 2 * s.nh_k) * s.d, h), wo=w(h, s.nh * s.d), wgu=w(2 * s.inter, h),
                                    wd=w(h, s.inter), n1=torch.ones(h, device=device, dtype=torch.float16),
                                    n2=torch.ones(h, device=device, dtype=torch.float16)))
        self.norm = torch.ones(h, device=device, dtype=torch.float16)
        self.lm_head = w(s.vocab, h)
        inv = 1.0 / (s.rope_theta ** (torch.arange(0, s.d, 2, device=device, dtype=torch.float32) / s.d))
        self.inv_freq = inv
        self.timers = NO_TIMERS

    def _rms(self, x, w):
        # one fused kernel (fp32 accumulation inside) instead of the eight of the spelled-out form: the decode step of this
        # harness was ~60 small launches per layer, more GPU time than its weight GEMVs (rocprofv3, round 3)
        return F.rms_norm(x, (x.shape[-1],), w, self.s.eps)

    def _rope_tables(self, pos):
        """cos and sign-folded sin of one decode step (bs, 1, 1, d), computed once per step, not per layer and tensor."""
        ang = pos.float()[:, None] * self.inv_freq[None, :]            # (bs, d/2)
        cos, sin = ang.cos(), ang.sin()
        return (torch.cat([cos, cos], -1).half()[:, None, None, :], torch.cat([-sin, sin], -1).half()[:, None, None, :])

    def _rope(self, x, tables):
        # x (bs, heads, 1, d); HF rotate_half convention: rotate_half(x) = cat(-x2, x1) = roll(x, d/2) with the sign of the
        # first half folded into the sin table
        cos, sin_signed = tables
        return torch.addcmul(x * cos, torch.roll(x, self.s.d // 2, -1), sin_signed)

    def _rope_seq_tables(self, n, pos0, device):
        """cos / sign-folded sin of positions pos0 .. pos0 + n - 1, shaped (1, n, 1, d) for token-major (bs, n, heads, d)."""
        ang = (pos0 + torch.arange(n, device=device)).float()[:, None] * self.inv_freq[None, :]      # (n, d/2)
        cos, sin = ang.cos(), ang.sin()
        return (torch.cat([cos, cos], -1).half()[None, :, None, :], torch.cat([-sin, sin], -1).half()[None, :, None, :])

    def step(self, tokens, pos, backend):
        s, tm = self.s, self.timers
        bs = tokens.shape[0]
        x = self.embed[tokens]                                           # (bs, hidden)
        if hasattr(backend, "begin_step"):
            backend.begin_step()
        rope = self._rope_tables(pos)
        for l, L in enumerate(self.layers):
            hN = self._rms(x, L["n1"])
            with tm("qkv_proj"):
                qkv = F.linear(hN, L["wqkv"])
                v = qkv[:, (s.nh + s.nh_k) * s.d:].view(bs, s.nh_k, 1, s.d)
            with tm("rotary"):
                # q and k heads lie side by side in the fused projection: one rotary pass over both
                qk = self._rope(qkv[:, : (s.nh + s.nh_k) * s.d].view(bs, s.nh + s.nh_k, 1, s.d), rope)
                q, k = qk[:, : s.nh], qk[:, s.nh:]
                if bs > 1:
                    q, k = q.contiguous(), k.contiguous()
                v = v.contiguous()
            a = backend.attend(l, q, k, v)
            with tm("o_proj"):
                x = torch.addmm(x, a.reshape(bs, s.nh * s.d), L["wo"].t())      # residual add in the GEMV's epilogue
            hN = self._rms(x, L["n2"])
            gu = F.linear(hN, L["wgu"])
            x = torch.addmm(x, F.silu(gu[:, : s.inter]) * gu[:, s.inter:], L["wd"].t())
        self.last_logits = F.linear(self._rms(x, self.norm), self.lm_head)
        return self.last_logits.argmax(-1)

    def prefill(self, tokens, backend, chunk=8192):
        """The prompt pass (q_len > 1): tokens (bs, n) -> first generated token (bs,).  Per layer the whole prompt goes
        through backend.prefill (PQ: bulk encode into pages + causal SDPA on the fp16 prompt, pq_utils.py:222-260); the
        MLP runs in row chunks to bound the (n, 2*inter) intermediate."""
        s, tm = self.s, self.timers
        bs, n = tokens.shape
        x = self.embed[tokens]                                           # (bs, n, hidden)
        rope = self._rope_seq_tables(n, 0, x.device)
        for l, L in enumerate(self.layers):
            hN = self._rms(x, L["n1"])
            with tm("qkv_proj"):
                qkv = F.linear(hN, L["wqkv"])
            with tm("rotary"):
                # token-major all the way: one rotary pass over the q and k heads where the projection left them; the
                # (bs, heads, n, d) tensors handed to the backend are strided views (the attention and encode kernels of
                # this library take row strides; torch's SDPA and cat do too)
                qk = self._rope(qkv[..., : (s.nh + s.nh_k) * s.d].view(bs, n, s.nh + s.nh_k, s.d), rope)
                q, k = qk[:, :, : s.nh].transpose(1, 2), qk[:, :, s.nh:].transpose(1, 2)
                v = qkv[..., (s.nh + s.nh_k) * s.d:].view(bs, n, s.nh_k, s.d).transpose(1, 2)
            a = backend.prefill(l, q, k, v)                              # (bs, nh, n, d)
            with tm("o_proj"):
                x.view(bs * n, s.hidden).addmm_(a.transpose(1, 2).reshape(bs * n, s.nh * s.d), L["wo"].t())
            del qkv, qk, q, k, v, a
            for c0 in range(0, n, chunk):
                xc = x[:, c0:c0 + chunk]
                hN = self._rms(xc, L["n2"])
                gu = F.linear(hN, L["wgu"])
                xc += F.linear(F.silu(gu[..., : s.inter]) * gu[..., s.inter:], L["wd"])
        self.last_logits = F.linear(self._rms(x[:, -1], self.norm), self.lm_head)
        return self.last_logits.argmax(-1)


class GraphedPQDecoder:
    """The whole decode step (all layers + lm_head + argmax + token feedback) captured in one hipGraph per kind of step
    (PagedPQCache.next_step_kind: plain, encode-ahead riding along, commit, in-line flush); lengths live on the device
    (PagedPQCache.lengths), so a replay touches no host state."""

    def __init__(self, model: LlamaShapeDecoder, backend: PQBackend, tokens: torch.Tensor, pos: torch.Tensor):
        self.model, self.be, self.tokens, self.pos = model, backend, tokens, pos
        cache, L, cap = backend.cache, model.s.n_layers, backend.cache.extended_residual_size
        backend.use_dev_lengths = True
        st = cache.host_state()
        dl_backup = [t.clone() for t in cache.lengths]
        tok0, pos0 = tokens.clone(), pos.clone()
        # the warm-up step below really runs: keep the window rows it overwrites
        self._eager_step()                     # allocates workspaces, warms hipBLASLt heuristics
        torch.cuda.synchronize()
        self.graphs = {}
        for name, state in cache.capture_states(st):
            cache.set_host_state(state)
            if cache.next_step_kind() != name:      # e.g. requests at different lengths: no encode-ahead, hence no 'pre' / 'commit'
                continue
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                self._eager_step()
            self.graphs[name] = gr
        cache.set_host_state(st)
        for t, b in zip(cache.lengths, dl_backup):
            t.copy_(b)
        tokens.copy_(tok0)
        pos.copy_(pos0)
        torch.cuda.synchronize()

    def _eager_step(self):
        nxt = self.model.step(self.tokens, self.pos, self.be)
        self.tokens.copy_(nxt)
        self.pos.add_(1)

    def step(self):
        cache = self.be.cache
        kind = cache.next_step_kind()
        self.graphs[kind].replay()
        cache.note_replayed_step(kind)
        return self.tokens


def speedtest(*, ctx=32768, decode=10, niter=5, bs=1, model="llama31_8b", layers=None,
              backends=("hf_baseline", "static_fp16", "pq_eager", "pq_graph"), prefill=True, breakdown=False,
              ttft_iters=2, device=None, log=None, deadline=None, M=64):
    """The reference's speed test (scripts/benchmarks/speedtest.py:85-117) on a Llama-shaped random-weight model: one
    warm-up generation, then `niter` timed ones of `decode` tokens; every generated token is handed to the host (the
    reference's streamer), the wall-clock interval between consecutive tokens is recorded, TPOT = sum(intervals[1:]) /
    (decode - 1) (:104) and, with `prefill`, TTFT = the prompt pass of `ctx` random tokens up to the first token on the
    host (:105) - through the model's q/k/v projections, RoPE, the backend's prefill (PQ: bulk encode of every layer into
    pages + causal attention on the fp16 prompt, pq_utils.py:222-260) and MLP.  Without `prefill` the caches are filled
    synthetically at `ctx` tokens.  `breakdown` adds the reference's per-section timers (Timer.py; speedtest.py:110-117).
    `deadline` (time.time() value): a backend that would start after it is skipped and says so.
    Returns a dict: config, one record per backend, speed-ups of the PQ backends over the two fp16 baselines."""
    device = device or torch.device("cuda", 0)
    shape = getattr(LlamaShape, model)()
    if layers:
        shape.n_layers = layers
    net = LlamaShapeDecoder(shape, device)
    dl = decode
    max_new = (niter + 2) * dl + 16
    results = {"config": {"model": model, "ctx": ctx, "decoding_length": dl, "niter": niter, "bs": bs, "pq_M": M,
                          "layers": shape.n_layers, "weights": "random fp16", "prompt": "random ids" if prefill else "synthetic cache fill",
                          "tpot": "speedtest.py:104 definition: mean inter-token interval, the first (prompt) interval excluded"}}

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
        tp = [sum(iv[1:]) / (dl - 1) for iv in (run_generation(step_fn) for _ in range(niter))]
        return sum(tp) / len(tp)

    def make_backend(name, filled):
        if name == "hf_baseline":
            return HFBaselineCache(shape, bs, ctx if filled else 0, device)
        if name == "static_fp16":
            return StaticFP16Cache(shape, bs, ctx, max_new, device)
        if name in ("pq_eager", "pq_graph"):
            return PQBackend(shape, bs, ctx, max_new, device, M=M, synthetic_fill=filled)
        raise ValueError(f"unknown backend {name}")

    prompt = torch.randint(0, shape.vocab, (bs, ctx), device=device) if prefill else None
    for name in backends:
        if deadline is not None and time.time() > deadline:
            results[name] = {"skipped": "time cap of the caller reached"}
            continue
        torch.cuda.empty_cache()
        tokens = torch.zeros(bs, dtype=torch.long, device=device)
        pos = torch.full((bs,), ctx, dtype=torch.long, device=device)
        rec = {}
        be = None
        if prefill and name != "static_fp16":      # (the preallocated baseline has no prompt pass of its own)
            ttft = []
            for it in range(ttft_iters + 1):       # first one is the warm-up (speedtest.py:92)
                be = None
                torch.cuda.empty_cache()
                be = make_backend(name, filled=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                first = net.prefill(prompt, be)
                first.cpu()                        # the first token reaches the host: intervals[0]
                if it:
                    ttft.append((time.perf_counter() - t0) * 1e3)
            rec["ttft_ms"] = round(sum(ttft) / len(ttft), 2)
            tokens.copy_(first)
        else:
            be = make_backend(name, filled=True)

        def eager_step():
            nxt = net.step(tokens, pos, be)
            tokens.copy_(nxt)
            pos.add_(1)
            return tokens

        if breakdown and name != "pq_graph":       # one generation with the section timers on (eager only)
            tm = SectionTimers()
            net.timers = be.timers = tm
            for _ in range(dl):
                eager_step().cpu()
            net.timers = be.timers = NO_TIMERS
            rec["breakdown_times_s"] = {k: round(v, 5) for k, v in sorted(tm.seconds.items())}
            rec["breakdown_calls"] = dict(sorted(tm.calls.items()))

        graphed = None
        if name != "pq_graph":
            step_fn = eager_step
        else:
            graphed = GraphedPQDecoder(net, be, tokens, pos)
            step_fn = graphed.step
        tp = measure(step_fn)
        rec.update({"tpot_ms": round(tp, 4), "tokens_per_s": round(bs * 1e3 / tp, 2)})
        results[name] = rec
        if log:
            log(f"{name} {rec}")
        del be, step_fn, graphed

    base = results.get("hf_baseline", {}).get("tpot_ms")
    stat = results.get("static_fp16", {}).get("tpot_ms")
    for k in ("pq_eager", "pq_graph"):
        if k in results and "tpot_ms" in results[k]:
            if base:
                results[k]["speedup_vs_hf_baseline"] = round(base / results[k]["tpot_ms"], 3)
            if stat:
                results[k]["speedup_vs_static_fp16"] = round(stat / results[k]["tpot_ms"], 3)
            bt = results.get("hf_baseline", {}).get("ttft_ms")
            if bt and "ttft_ms" in results[k]:
                results[k]["ttft_vs_hf_baseline"] = round(results[k]["ttft_ms"] / bt, 3)
    return results
