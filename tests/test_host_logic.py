"""CPU tests of host-side logic: page manager, split/name heuristics, request sharding, and the
multi-process timing path of bench.py on the gloo backend (world size 2)."""
import json
import os
import socket
from pathlib import Path
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from million_amd import sharding
from million_amd.pq_cache import PageManager, l2Ns, nbits2dtype, scalarTypeToStr


def test_l2ns_and_names_match_reference_table(golden_dir):
    man = json.loads((golden_dir / "manifest.json").read_text())
    for l, ns in man["l2Ns"].items():
        assert l2Ns(int(l)) == ns
    for n, dt in man["nbits2dtype"].items():
        assert str(nbits2dtype(int(n))) == dt
    assert scalarTypeToStr(torch.float16) == "f16"
    with pytest.raises(ValueError):
        scalarTypeToStr(torch.bfloat16)


def test_page_manager_semantics():
    """dynamic_paged_pq_utils.py:137-241: allocate / free / reuse / growth capped by max_pages."""
    pm = PageManager(page_size=64, initial_pages=4, max_pages=10, M=64)
    ids = [pm.allocate_page() for _ in range(4)]
    assert ids == [0, 1, 2, 3] and pm.get_stats()["free_pages"] == 0
    pm.free_page(2)
    assert pm.allocate_page() == 2 and pm.page_reuse_count == 1
    more = pm.allocate_pages(5)                     # forces growth by exactly the missing amount
    assert len(set(ids + more)) == 9 and pm.current_active_pages == 9
    assert pm.allocate_page() == 9                  # grows to the cap
    with pytest.raises(RuntimeError):
        pm.allocate_page()                          # max_pages reached
    pm.free_page(12345)                             # unknown id: ignored, as in the reference (:234-236)
    st = pm.get_stats()
    assert st["allocated_pages"] == 10 and st["max_pages"] == 10 and st["total_expansions"] >= 2
    unlimited = PageManager(initial_pages=2, max_pages=None)
    assert len(unlimited.allocate_pages(200)) == 200


def test_shard_requests():
    # BASELINE configs[3]: 16 requests over 8 GPUs -> 2 per rank
    assert [len(sharding.shard_requests(16, 8, r)) for r in range(8)] == [2] * 8
    got = sum((sharding.shard_requests(10, 4, r) for r in range(4)), [])
    assert got == list(range(10))
    assert [len(sharding.shard_requests(10, 4, r)) for r in range(4)] == [3, 3, 2, 2]
    assert sharding.shard_requests(1, 4, 3) == []
    with pytest.raises(ValueError):
        sharding.shard_requests(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    done = []

    def step():
        time.sleep(0.01 * (1 + 2 * rank))       # rank 1 is 3x slower: the job time must be rank 1's
        done.append(1)

    elapsed, per_rank = sharding.timed_steps_per_rank(step, 5, lambda: None, dist)
    value, ms = sharding.aggregate_throughput(1, 5, elapsed, world)
    out.put((rank, len(done), elapsed, value, ms, sharding.shard_requests(5, world, rank), per_rank))
    dist.barrier()
    dist.destroy_process_group()


def test_multirank_timing_gloo():
    """bench.py --gpus N contract: exactly K steps per rank, barrier on both sides, MAX over ranks,
    value = all ranks' units / that time.  World size 2 on CPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, n0, e0, v0, ms0, s0, pr0), (r1, n1, e1, v1, ms1, s1, pr1) = res
    assert n0 == n1 == 5
    # every rank holds the same list of the ranks' OWN times, indexed by rank: the slow rank is visible (rank 1 sleeps 3x as long)
    assert pr0 == pr1 and len(pr0) == 2
    assert pr0[1] >= 5 * 0.03 * 0.9 and pr0[0] < 0.7 * pr0[1] and max(pr0) <= e0
    assert e0 == e1 and e0 >= 5 * 0.03 * 0.9            # both ranks report rank 1's (max) time
    assert v0 == v1 == pytest.approx(10 / e0)
    assert s0 + s1 == list(range(5))


def test_bench_launcher_spawns_ranks_dry_cpu():
    """`python bench.py --gpus 2 --dry-cpu`: with no WORLD_SIZE in the environment bench.py re-launches itself under
    torch.distributed.run with 2 ranks (gloo here, RCCL on the GPU box) and rank 0 prints ONE JSON line whose n_gpus
    comes from dist.get_world_size() and whose ranks_seen comes from an all-gather over the job's backend; with --gpus
    > 1 the default is BASELINE configs[3]'s 2 requests per GPU."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--dry-cpu", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=str(root))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["scaling"] == "weak"
    assert [x[0] for x in rec["config"]["ranks_seen"]] == [0, 1]
    assert rec["config"]["batch_per_gpu"] == 2
    # a slow rank is visible in the line: each rank's own ms per step and launch period / roofline fraction, indexed by rank
    pr = rec["per_rank_ms_per_step"]
    assert len(pr) == 2 and pr[1] > 1.5 * pr[0] and max(pr) <= rec["ms_per_step"] * 1.001
    assert len(rec["roofline"]["per_rank"]["frac"]) == 2 and len(rec["roofline"]["per_rank"]["launch_us_mean"]) == 2
    # rank 1's step sleeps twice as long: the job time is rank 1's, the value counts both ranks' requests
    assert rec["ms_per_step"] >= 4.0 * 0.9 and rec["value"] == pytest.approx(2 * 2 * 1000.0 / rec["ms_per_step"], rel=1e-3)


def test_bench_window_fill_puts_a_flush_in_the_timed_region():
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parents[1] / "bench.py")
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    for warmup, steps in ((5, 20), (8, 64), (0, 1), (3, 200), (100, 10)):
        r = B.window_fill_at_start(128, warmup, steps)
        assert 1 <= r <= 128
        # simulate the window: flush (r -= 64) at the start of a step that finds r >= 128, then append
        flushes = []
        for i in range(warmup + steps):
            if r >= 128:
                r -= 64
                flushes.append(i - warmup)
            r += 1
        assert any(0 <= f < steps for f in flushes), (warmup, steps, flushes)
    assert B.parse(["--gpus", "8"]).batch_per_gpu == 2 and B.parse([]).batch_per_gpu == 1
    assert B.parse(["--gpus", "8", "--batch-per-gpu", "1"]).batch_per_gpu == 1


def test_harness_fp16_backends_agree_cpu():
    """million_amd/harness.py: the HF-recipe baseline (torch.cat + repeat_kv + SDPA, modeling_llama.py:403-443)
    and the preallocated GQA-view baseline are the same attention; tiny shape, CPU, fp32 weights."""
    from million_amd import harness as H
    shape = H.LlamaShape(hidden=64, n_layers=2, nh=4, nh_k=2, d=16, inter=96, vocab=50)
    torch.manual_seed(0)
    model = H.LlamaShapeDecoder(shape, torch.device("cpu"))
    ctx, bs = 24, 2
    hf = H.HFBaselineCache(shape, bs, ctx, torch.device("cpu"))
    st = H.StaticFP16Cache(shape, bs, ctx, 8, torch.device("cpu"))
    for l in range(shape.n_layers):
        st.k[l][:, :, :ctx] = hf.k[l]
        st.v[l][:, :, :ctx] = hf.v[l]
    tok_a = torch.zeros(bs, dtype=torch.long)
    tok_b = tok_a.clone()
    pos = torch.full((bs,), ctx, dtype=torch.long)
    for i in range(4):
        tok_a = model.step(tok_a, pos + i, hf)
        tok_b = model.step(tok_b, pos + i, st)
        assert torch.equal(tok_a, tok_b)
    assert hf.k[0].shape[2] == ctx + 4 and st.T == ctx + 4
    # repeat_kv layout: head h of the output is kv head h // G
    x = torch.arange(2 * 2 * 3 * 1, dtype=torch.float32).view(2, 2, 3, 1)
    r = H.repeat_kv(x, 2)
    assert r.shape == (2, 4, 3, 1) and torch.equal(r[:, 1], x[:, 0]) and torch.equal(r[:, 2], x[:, 1])


def test_harness_prefill_equals_stepwise_and_timers_cpu():
    """harness.LlamaShapeDecoder.prefill (the q_len > 1 pass that TTFT times, speedtest.py:105) gives the logits that
    prefilling one token less and then decoding the last token gives; SectionTimers accumulates the reference's
    breakdown sections (speedtest.py:110-117)."""
    from million_amd import harness as H
    shape = H.LlamaShape(hidden=64, n_layers=2, nh=4, nh_k=2, d=16, inter=96, vocab=50)
    torch.manual_seed(1)
    model = H.LlamaShapeDecoder(shape, torch.device("cpu"))
    for L in model.layers:
        for k in L:
            L[k] = L[k].float()
    model.embed, model.lm_head, model.norm = model.embed.float(), model.lm_head.float(), model.norm.float()

    def empty():
        be = H.HFBaselineCache(shape, 2, 0, torch.device("cpu"))
        be.k, be.v = [t.float() for t in be.k], [t.float() for t in be.v]
        return be
    toks = torch.randint(0, 50, (2, 24))
    model.prefill(toks, empty(), chunk=7)
    full = model.last_logits.clone()
    be = empty()
    tm = H.SectionTimers(sync=lambda: None)
    model.timers = be.timers = tm
    model.prefill(toks[:, :23], be)
    model.step(toks[:, 23], torch.full((2,), 23), be)
    model.timers = H.NO_TIMERS
    assert (full - model.last_logits).abs().max() < 1e-5
    assert be.k[0].shape[2] == 24
    assert set(tm.seconds) == {"qkv_proj", "rotary", "cat", "repeat_kv", "sdpa", "o_proj"}
    assert tm.calls["sdpa"] == 2 * shape.n_layers and all(v >= 0 for v in tm.seconds.values())


def test_fvecs_and_centroid_formats(tmp_path):
    """million_amd/formats.py against the byte layout the reference writes (fvecio.py:23-43; main_pq.py:222-260)."""
    import struct
    import numpy as np
    from million_amd import formats as F
    rs = np.random.RandomState(3)
    a, b = rs.standard_normal((5, 128)).astype(np.float32), rs.standard_normal((3, 128)).astype(np.float32)
    fn = tmp_path / "key_sampled_64_8.fvecs"
    # hand-built bytes in the reference's record format: int32 d, then d float32, per vector
    with open(fn, "wb") as f:
        for v in a:
            f.write(struct.pack("<i", 128) + v.tobytes())
    np.testing.assert_array_equal(F.read_fvecs(fn), a)
    F.write_fvecs(fn, b)                                   # default mode appends
    np.testing.assert_array_equal(F.read_fvecs(fn), np.concatenate([a, b]))
    with open(fn, "rb") as f:                               # and our writer produces those same bytes
        blob = f.read()
    assert blob[5 * 516: 5 * 516 + 4] == struct.pack("<i", 128) and blob[5 * 516 + 4: 6 * 516] == b[0].tobytes()
    fn2 = tmp_path / "new.fvecs"
    F.write_fvecs(fn2, a[0])                                # 1-D input = one vector; 'ab' on a missing file creates it
    assert F.read_fvecs(fn2).shape == (1, 128)
    with open(fn2, "ab") as f:
        f.write(struct.pack("<i", 64) + b"\0" * 256)
    with pytest.raises(ValueError):
        F.read_fvecs(fn2)
    (tmp_path / "empty.fvecs").write_bytes(b"")
    assert F.read_fvecs(tmp_path / "empty.fvecs").shape == (0, 0)

    kc, vc = torch.randn(64, 256, 2), torch.randn(64, 256, 2)
    kp, vp = F.save_centroids(tmp_path / "cents", kc, vc, nbits=8)
    assert kp.name == "key_cent_64_8.pq.pt" and vp.name == "val_cent_64_8.pq.pt"
    assert torch.equal(torch.load(kp, weights_only=True), kc)          # what the reference's loader would see
    k2, v2 = F.load_centroids(tmp_path / "cents", 64, 8, d=128)
    assert k2.dtype == torch.float16 and torch.equal(k2, kc.half()) and torch.equal(v2, vc.half())
    with pytest.raises(FileNotFoundError):
        F.load_centroids(tmp_path / "cents", 32, 8)
    with pytest.raises(ValueError):
        F.load_centroid_file(kp, M=32)
    with pytest.raises(ValueError):
        F.load_centroid_file(kp, nbits=7)
    torch.save({"not": "a tensor"}, tmp_path / "cents" / "key_cent_16_8.pq.pt")
    torch.save(vc, tmp_path / "cents" / "val_cent_16_8.pq.pt")
    with pytest.raises(ValueError):
        F.load_centroids(tmp_path / "cents", 16, 8)


def test_formats_against_reference_written_files(golden_dir):
    """Files written by the REFERENCE's own writers (tools/gen_golden.py: fvecio.py:35-43 write_fvecs in its default
    append mode; main_pq.py:222-226 torch `save` of the fp32 centroid tensor), committed as data: million_amd/formats.py
    must read them back exactly, and its own writer must produce the same bytes."""
    import hashlib
    import json

    import numpy as np
    import torch
    from million_amd import formats as F
    man = json.loads((golden_dir / "manifest.json").read_text())["formats"]
    fv = golden_dir / man["fvecs_file"]
    assert hashlib.sha256(fv.read_bytes()).hexdigest() == man["fvecs_sha256"]
    want = np.asarray(man["fvecs_expected"], dtype=np.float32)
    got = F.read_fvecs(fv)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, want)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        mine = f"{td}/mine.fvecs"
        F.write_fvecs(mine, want[:3])
        F.write_fvecs(mine, want[3:])
        assert open(mine, "rb").read() == fv.read_bytes()
    pq = man["pq_pt"]
    kc, vc = F.load_centroids(golden_dir, pq["M"], pq["nbits"], d=pq["d"], dtype=torch.float32)
    np.testing.assert_array_equal(kc.numpy(), np.asarray(pq["key"], dtype=np.float32))
    np.testing.assert_array_equal(vc.numpy(), np.asarray(pq["val"], dtype=np.float32))
    k16, _ = F.load_centroids(golden_dir, pq["M"], pq["nbits"])      # default: the model dtype the caches take
    assert k16.dtype == torch.float16 and k16.shape == (4, 4, 2)


def test_paged_cache_host_lifecycle_on_cpu():
    """Host half of the request lifecycle (no kernels): on-demand page assignment per request, release() returns exactly the
    released slot's pages to the PageManager and zeroes its lengths, the other slot keeps its pages; the per-layer views
    (cache._T[l] ...) read request 0 and write every request."""
    import numpy as np
    from million_amd.pq_cache import PagedPQCache
    c = PagedPQCache(bs=2, nh=8, num_key_value_heads=2, M=64, layer_num=3, d=128, page_size=64, extended_residual_size=128,
                     max_tokens=1024, preallocate=False, device="cpu")
    assert c.page_manager.get_stats()["allocated_pages"] == 0
    c._assign_pages(1, 5)                       # both requests of layer 1: 5 pages x 2 kv heads each
    c._assign_pages(2, 3, b=1)
    assert c.page_manager.get_stats()["allocated_pages"] == 2 * 2 * 5 + 2 * 3
    ids = c.page_ids[1].numpy()
    assert len(set(ids[:, :, :5].ravel().tolist())) == 20
    c._T_a[1] = [300, 200]
    c._r_a[1] = [5, 128]
    assert c._T[1] == 300 and c.residualed_tokens[1] == 5 and not c._lockstep(1) and c._lockstep(0)
    assert c.next_step_flushes(1) and not c.next_step_flushes(0)
    kept = set(ids[0, :, :5].ravel().tolist())
    c.release(1)
    st = c.page_manager.get_stats()
    assert st["allocated_pages"] == 10 and set(c.page_manager.allocated_pages) == kept
    assert c._T_a[1].tolist() == [300, 0] and c._r_a[1].tolist() == [5, 0] and c._pages_a[:, 1].tolist() == [0, 0, 0]
    c._T[0] = 64                                 # legacy per-layer write: every request
    assert c._T_a[0].tolist() == [64, 64]
    st0 = c.host_state()
    c.set_host_state(([7] * 3, [1, 2, 3], st0[2], st0[3]))
    assert c._seen_a.tolist() == [[7, 7]] * 3 and c._r_a.tolist() == [[1, 1], [2, 2], [3, 3]] and c._T_a[1].tolist() == [300, 0]
    c.set_host_state(st0)
    c._r_a[:] = [[127, 128], [0, 0], [128, 128]]
    c._T_a[:] = 0
    c.note_replayed_step()
    assert c._r_a.tolist() == [[128, 65], [1, 1], [65, 65]] and c._T_a.tolist() == [[0, 64], [0, 0], [64, 64]]
    assert c._rs_a.tolist() == [[0, 64], [0, 0], [64, 64]]


def test_paged_cache_step_kinds_state_machine_on_cpu():
    """PagedPQCache.next_step_kind / note_replayed_step / capture_states (the host side of encode-ahead, pq_cache.py):
    plain steps, the encode-ahead step(s) a few steps after a flush, the commit step when the window is full, the in-line
    flush as the fallback (requests at different lengths, flags lost); one capture state per kind reproduces its kind."""
    from million_amd.pq_cache import PagedPQCache
    c = PagedPQCache(bs=2, nh=8, num_key_value_heads=2, M=64, layer_num=4, d=128, page_size=64, extended_residual_size=128,
                     max_tokens=4096, preallocate=False, device="cpu")
    assert c.encode_ahead_at() == 72 and c._ea_groups() == [(0, 4)]
    kinds = []
    for _ in range(200):                         # empty window after a prefill: commits at steps 128 and 192
        k = c.next_step_kind()
        kinds.append(k)
        c.note_replayed_step(k)
    assert [i for i, k in enumerate(kinds) if k != "plain"] == [72, 128, 136, 192]
    assert [kinds[i] for i in (72, 128, 136, 192)] == ["pre", "commit", "pre", "commit"]
    assert c._T_a.tolist() == [[128, 128]] * 4 and c._r_a.tolist() == [[200 - 128] * 2] * 4
    # spread over two steps (what large batches do by themselves: bs * layers > 32)
    c.encode_ahead_steps = 2
    assert c._ea_groups() == [(0, 2), (2, 4)]
    kinds = []
    for _ in range(64):
        k = c.next_step_kind()
        kinds.append(k)
        c.note_replayed_step(k)
    assert [k for k in kinds if k != "plain"] == ["pre0", "pre1", "commit"]
    big = PagedPQCache(bs=8, nh=8, num_key_value_heads=2, M=64, layer_num=32, d=128, page_size=64, extended_residual_size=128,
                       max_tokens=256, preallocate=False, device="cpu")
    assert len(big._ea_groups()) == 8 and big._ea_groups()[0] == (0, 4)
    # one capture state per kind, and each reproduces its kind
    st = c.host_state()
    names = []
    for name, state in c.capture_states(st):
        c.set_host_state(state)
        assert c.next_step_kind() == name
        names.append(name)
    assert names == ["plain", "pre0", "pre1", "commit", "flush"]
    c.set_host_state(st)
    # fallbacks: a request at another length -> no encode-ahead; its window full without the flags -> in-line flush
    c._r_a[:] = 100
    c._pre_a[:] = 0
    c._r_a[:, 1] = 90
    assert c.next_step_kind() == "plain"
    c._r_a[:, 1] = 128
    assert c.next_step_kind() == "flush"
    c._r_a[:] = 128
    c._pre_a[:] = 1
    c._pre_a[3, 1] = 0
    assert c.next_step_kind() == "flush"
    c.release(1)                                  # a recycled slot loses its flags with its lengths
    assert c._pre_a[:, 1].tolist() == [0] * 4


def test_make_install_pth_resolves_bindings_without_pythonpath(tmp_path):
    """`make install` (tools/install_pth.py, reference makefile:1-4): with the .pth file in a site directory, `import bindings`
    and `import million_amd` resolve from an unrelated working directory with no PYTHONPATH; --uninstall removes it."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    site_dir = tmp_path / "site"
    r = subprocess.run([sys.executable, str(root / "tools" / "install_pth.py"), "--target", str(site_dir)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the .pth names build/site (two symlinks), not the checkout root: tests / tools / oracle do not become importable
    assert (site_dir / "million_hip.pth").read_text().strip() == str(root / "build" / "site")
    assert sorted(p.name for p in (root / "build" / "site").iterdir()) == ["bindings", "million_amd"]
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    code = ("import site, sys; site.addsitedir(%r); import bindings, million_amd; "
            "print(bindings.__file__); "
            "assert %r not in sys.path; "
            "assert hasattr(bindings, 'flash_decoding_allocated_buffer_f16u8_Ns32Lt128d128M64C256')") % (str(site_dir), str(root))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, r.stderr
    assert str(root) in r.stdout
    r = subprocess.run([sys.executable, str(root / "tools" / "install_pth.py"), "--target", str(site_dir), "--uninstall"], capture_output=True, text=True)
    assert r.returncode == 0 and not (site_dir / "million_hip.pth").exists()


def test_split_merge_ticket_word_protocol_every_interleaving():
    """The split merge of the MFMA kernels (attn_mfma.hip "Tail", common.h record word [2]): ONE word per (b, kv head) holds the
    arrival count (a ticket = atomic add of 256) and the give-up bits of the merge helpers (helper k: atomic OR of 1 << k).
    A helper wave out of patience ORs its bit and leaves iff the count the atomic returns is incomplete; the primary - the
    workgroup whose ticket completes the count - merges the heads of every helper whose bit its own ticket returned.  Model of
    exactly that rule, run over EVERY order of the atomics of 3 helpers (each with 2 waves that may or may not run out of
    patience) and the primary's ticket: each head must be merged by somebody - and a head the primary takes over must be one
    whose helper may really have left.  (The atomics of one word are serialised by the memory system, so an order of the
    atomics is all there is to enumerate; the device code is exercised by tests/test_gpu_parity.py::test_attn_merge_*.)"""
    import itertools

    ns, nm = 8, 4                       # 8 splits, 4 mergers: helpers = arrival indices 4, 5, 6; primary = 7
    helpers = range(nm - 1)
    checked = 0
    # every helper wave either sees the flags in time (no atomic) or gives up (one OR somewhere in the order)
    for gives in itertools.product([0, 1, 2], repeat=nm - 1):          # waves of helper k that run out of patience
        events = ["ticket"] + [("or", k, w) for k in helpers for w in range(gives[k])]
        for order in set(itertools.permutations(events)):
            word = (ns - 1) << 8                                       # every ticket but the primary's has been taken
            left = {k: 0 for k in helpers}                             # waves of helper k that left without merging
            stayed = {k: 2 - gives[k] for k in helpers}                # waves that merged their part (saw the flags, or polled on)
            seen_by_primary = 0
            for ev in order:
                if ev == "ticket":
                    seen_by_primary = word & 0xFF                      # the primary's own ticket returns the bits set so far
                    word += 256
                else:
                    _, k, _w = ev
                    old, word = word, word | (1 << k)
                    if (old >> 8) < ns:
                        left[k] += 1                                   # incomplete count: the primary's ticket comes later
                    else:
                        stayed[k] += 1                                 # complete: everybody is resident, poll on and merge
            for k in helpers:
                primary_merges = bool(seen_by_primary >> k & 1)
                # every part of head k is written: by the helper's waves that stayed, or - whole head - by the primary
                assert primary_merges or left[k] == 0, (gives, order)
                # the primary never takes over a head whose helper did not raise the bit before the last ticket
                assert not primary_merges or left[k] > 0, (gives, order)
                assert left[k] + stayed[k] == 2
            checked += 1
    assert checked > 500
