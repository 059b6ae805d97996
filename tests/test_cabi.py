"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/million_hip.h declares (no compute calls without a GPU); argument validation that
happens before any launch; the `bindings` module exports the reference's names."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from million_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from million_amd import _lib
    hdr = (ROOT / "include" / "million_hip.h").read_text()
    declared = set(re.findall(r"\b(million_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"million_stream_t"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.million_version() == 1


def test_struct_sizes_match_header(lib):
    """ctypes mirrors must have the C layout: compile a tiny C program against the header."""
    import subprocess, tempfile
    from million_amd import _lib
    src = '#include "million_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu\\n", sizeof(million_encode_desc), sizeof(million_attn_desc), sizeof(million_prefill_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "a.c").write_text(src)
        subprocess.check_call(["gcc", "-std=c11", "-I", str(ROOT / "include"), "-o", f"{td}/a", f"{td}/a.c"])
        enc, attn, pre = map(int, subprocess.check_output([f"{td}/a"]).split())
    assert enc == ctypes.sizeof(_lib.EncodeDesc)
    assert attn == ctypes.sizeof(_lib.AttnDesc)
    assert pre == ctypes.sizeof(_lib.PrefillDesc)


def test_argument_validation_without_gpu(lib):
    from million_amd import _lib
    d = _lib.AttnDesc()
    d.struct_size = 3
    rc = lib.million_pq_decode_attn(ctypes.byref(d), *([None] * 11), 0, None)
    assert rc == -3 and b"struct_size" in lib.million_last_error()
    d.struct_size = ctypes.sizeof(_lib.AttnDesc)
    d.bs, d.nh, d.nh_k, d.d, d.M, d.C = 1, 32, 8, 128, 64, 256
    d.n_tokens, d.r, d.resid_cap = 10, 200, 128
    assert lib.million_pq_decode_attn(ctypes.byref(d), *([None] * 11), 0, None) == -3
    assert b"r=200" in lib.million_last_error()
    d.r = 17
    assert lib.million_pq_decode_attn(ctypes.byref(d), *([None] * 11), 0, None) == -3   # null pointers
    d.nh = 33
    assert lib.million_pq_decode_attn(ctypes.byref(d), *([None] * 11), 0, None) == -1
    assert lib.million_attn_workspace_bytes(ctypes.byref(d)) == 0
    d.nh = 32
    assert lib.million_attn_workspace_bytes(ctypes.byref(d)) > 8 * 65 * 4 * 130 * 4
    e = _lib.EncodeDesc()
    assert lib.million_pq_encode(ctypes.byref(e), None, None, None, None, None) == -3
    assert lib.million_prepared_cents_bytes(64, 256, 2) == 64 * 256 * 2 * (2 * 2 + 4)   # two fp16 images + one fp32


def test_kernel_kind_policy_without_gpu(lib):
    """million_attn_kernel_kind is host logic (no launch): which kernel a descriptor gets.  Every batch x context of the
    streaming kernel's shapes stays on it (kind 1) - calls with more than 64 rounds per wave get more splits instead of the
    grouped (5) or scalar (0) kernel (Interface.template.cu:45,62-77: the reference has one kernel for any (bs, nh, T)) - and the
    tile kernel (3) takes the rest of the build matrix with up to 16 query heads per kv head in one launch."""
    from million_amd import _lib

    def kind(bs, nh, nh_k, T, d=128, M=64, C=256, paged=True):
        q = _lib.AttnDesc()
        q.struct_size = ctypes.sizeof(_lib.AttnDesc)
        q.bs, q.nh, q.nh_k, q.d, q.M, q.C = bs, nh, nh_k, d, M, C
        q.n_tokens, q.r, q.resid_cap, q.resid_start = T, 17, 128, 0
        q.resid_stride_b, q.resid_stride_h = nh_k * 128 * d, 128 * d
        q.k_layout = q.v_layout = _lib.MILLION_KV_PAGED if paged else _lib.MILLION_KV_ROWMAJOR
        q.page_size, q.n_pages_cap = 64, max(1, (T + 63) // 64)
        q.k_stride_b, q.k_stride_h, q.v_stride_b, q.v_stride_h = nh_k * T * M, T * M, nh_k * T * M, T * M
        return lib.million_attn_kernel_kind(ctypes.byref(q))

    assert kind(1, 32, 8, 32768) == 1
    for bs, nh, nh_k, T, C in ((16, 32, 8, 40000, 256), (8, 32, 32, 32768, 256), (32, 32, 8, 20000, 128), (64, 32, 8, 131072, 256),
                               (1, 32, 8, 1000000, 256)):
        assert kind(bs, nh, nh_k, T, C=C) == 1, (bs, nh, nh_k, T, C)
    assert kind(1, 32, 8, 0) == 5                      # nothing quantised yet: grouped MFMA kernel (window only)
    assert kind(1, 32, 8, 4096, paged=False) == 2      # the reference's 10-argument layout: V transposed first
    for d, M in ((128, 16), (64, 64), (64, 32), (64, 16)):
        # d_m = 8 form of the streaming kernel, and (round 5) the lean kernel's d = 64 forms (d_m = 1 / 2 / 4): up to 4 query heads per kv head
        fast = True      # round 5: d = 64 / M = 64 (d_m = 1) too, as d_m = 2 with every odd dim zero
        assert kind(1, 32, 8, 4096, d=d, M=M) == (1 if fast else 3)
        assert kind(1, 128, 8, 4096, d=d, M=M) == 1    # 16 heads per kv head: 4 virtual kv heads of 4 query heads per real one (round 5; 5 .. 16: ceil(G / 4) parts)
        assert kind(1, 40, 8, 4096, d=d, M=M) == 1     # 5 heads per kv head: parts of 3 + 2
        assert kind(1, 32, 8, 4096, d=d, M=M, paged=False) == (2 if fast else 4)
    assert kind(1, 32, 8, 0, d=64, M=32) == 3          # nothing quantised yet at d = 64: the tile kernel (the lean kernel is not asked)
    assert kind(1, 32, 8, 4096, d=64, M=32, C=128) == 1        # (round 5: 128 centroids on the lean kernel too)
    assert kind(1, 32, 8, 4096, d=128, M=16, C=128) == 1
    assert kind(1, 64, 8, 4096, d=128, M=16) == 1      # 8 heads per kv head: two virtual kv heads of 4 on the streaming kernel (round 5)
    assert kind(64, 256, 32, 4096, d=128, M=16) == 3   # ... unless the virtual pairs outgrow the workspace head (2048 records)
    assert kind(1, 32, 8, 4096, C=64) == 0             # off the build matrix: scalar kernel


def test_bindings_exports_reference_names():
    import bindings
    # the 240 names the reference generates (setup.py:26-54) ...
    for Ns in (2, 4, 8, 16, 32):
        for d in (64, 128):
            for M in (16, 32, 64):
                for C in (128, 256):
                    for fam in ("allocated_buffer", "allocated_paged_buffer", "allocated_paged_split_qkv_buffer",
                                "allocated_paged_lastblock_sync_buffer"):
                        assert callable(getattr(bindings, f"flash_decoding_{fam}_f16u8_Ns{Ns}Lt{d}d{d}M{M}C{C}"))
    # ... plus Ns1 (l2Ns returns 1 for l <= 64), the 13-arg paged call and the encode entry
    assert callable(bindings.flash_decoding_allocated_buffer_f16u8_Ns1Lt128d128M64C256)
    assert callable(bindings.flash_decoding_paged_v_f16u8_Ns32Lt128d128M64C256)
    assert callable(bindings.pq_encode_f16u8_d128M64C256)
    assert "flash_decoding_paged_v_f16u8_Ns32Lt128d128M64C256" in dir(bindings)


def test_ops_refuse_cpu_tensors():
    import torch
    from million_amd import ops
    with pytest.raises(RuntimeError):
        ops.pq_encode(torch.zeros(1, 1, 4, 128, dtype=torch.float16), torch.zeros(64, 256, 2, dtype=torch.float16))


def test_planned_call_checks_store_extents_on_every_call(monkeypatch):
    """the plan signature of the reference's 10-argument call holds nothing T-dependent, so the cached
    fast path must itself refuse a row-major value store that is shorter than the key store (million_transpose_v_codes would
    read T rows of it), a store of another batch, or another M - on the SECOND and later calls of a signature, where
    pq_decode_attn's full validation no longer runs.  CPU tensors: the check must fire before anything touches the library."""
    import torch
    from million_amd import ops
    monkeypatch.setattr(ops, "_stream", lambda: 0)
    bs, nh, nhk, d, M, C, T = 1, 8, 2, 128, 64, 256, 256
    q = torch.zeros(bs, nh, 1, d, dtype=torch.float16)
    kres = torch.zeros(bs, nhk, 128, d, dtype=torch.float16)
    vres = torch.zeros_like(kres)
    kc = torch.zeros(bs, nhk, T, M, dtype=torch.uint8)
    cents = torch.zeros(M, C, d // M, dtype=torch.float16)

    def call(k, v):
        sig = ops._plan_signature(0, q, k, v, kres, vres, None, 0, M, C)
        monkeypatch.setitem(ops._plans, sig, ops._Plan())      # "this signature has been validated once"
        return ops.decode_attn_planned(q, k, v, cents, cents, kres, vres, 17, M=M, C=C)

    same_sig = lambda v: ops._plan_signature(0, q, kc, v, kres, vres, None, 0, M, C) == ops._plan_signature(0, q, kc, kc, kres, vres, None, 0, M, C)
    short_v = torch.zeros(bs, nhk, T - 64, M, dtype=torch.uint8)
    assert same_sig(short_v)                       # the hazard: a shorter V store is the SAME signature
    with pytest.raises(RuntimeError, match="n_tokens exceeds value_codes"):
        call(kc, short_v)
    with pytest.raises(RuntimeError, match="n_tokens exceeds value_codes"):
        call(kc, torch.zeros(bs, nhk + 1, T, M, dtype=torch.uint8))      # other head count, same strides [2:]
    with pytest.raises(RuntimeError, match="n_tokens exceeds value_codes"):
        call(kc, torch.zeros(bs + 1, nhk, T, M, dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="do not match query batch"):
        k2 = torch.zeros(bs + 1, nhk, T, M, dtype=torch.uint8)
        call(k2, k2)
    # a plan evicted by another thread between the lookup and move_to_end is not an error
    sig = ops._plan_signature(0, q, kc, short_v, kres, vres, None, 0, M, C)

    class Evicting(type(ops._plans)):
        def get(self, key, default=None):
            v = super().get(key, default)
            self.pop(key, None)
            return v
    plans = Evicting()
    plans[sig] = ops._Plan()
    monkeypatch.setattr(ops, "_plans", plans)
    with pytest.raises(RuntimeError, match="n_tokens exceeds value_codes"):      # reaches the per-call check, no KeyError
        ops.decode_attn_planned(q, kc, short_v, cents, cents, kres, vres, 17, M=M, C=C)
