"""GPU parity tests (run with -m gpu on an MI355X): HIP path through the C ABI vs the CPU oracle.

Bars (BASELINE.md 2 / SURVEY.md 8c): uint8 codes bit-exact with oracle/pq_oracle.c:pq_encode_direct;
fp16 attention output within 1e-3 relative (rel-L2) and mean-abs < 1e-3 (the reference's own bar,
scripts/utils/pq_utils.py:374-379) of the fp64 oracle.  Nothing here reads /root/reference.
"""
import ctypes
import time
import hashlib
import json

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3       # ||out - gold|| / ||gold||   (north_star: "fp16 attention output within 1e-3 rel")
MEAN_ABS_TOL = 1e-3  # reference bar, pq_utils.py:374-379


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from million_amd import ops, _lib
    _lib.load()   # fails loudly if libmillion_hip.so is missing
    return torch, ops


def _dev(torch, c):
    return {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in c.items()}


def _check(out, gold, what=""):
    out = out.astype(np.float64)
    gold = np.asarray(gold, dtype=np.float64)
    assert np.isfinite(out).all(), what
    rel = np.linalg.norm(out - gold) / max(np.linalg.norm(gold), 1e-30)
    mae = np.abs(out - gold).mean()
    assert rel < REL_TOL and mae < MEAN_ABS_TOL, f"{what}: rel={rel:.3e} mean_abs={mae:.3e}"
    return rel, mae


def _run_rowmajor(torch, ops, c, M, C):
    t = _dev(torch, c)
    kp = ops.prepare_cents(t["k_cents"], cache=False)
    vp = ops.prepare_cents(t["v_cents"], cache=False)
    out = ops.pq_decode_attn(t["q"], t["k_codes"], t["v_codes"], kp, vp, t["k_res"], t["v_res"], c["r"], M=M, C=C)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _run_paged(torch, ops, oracle, c, M, C, ps, k_paged=True, shuffle=True, i64=False, poison_out=False):
    t = _dev(torch, c)
    dst = torch.full_like(t["q"], float("nan")) if poison_out else None      # a head nobody writes must show
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    if shuffle and vpool.shape[0] > 1:      # physical page order must not matter
        perm = np.random.RandomState(5).permutation(vpool.shape[0])
        inv = np.argsort(perm)
        vpool, kpool, ids = vpool[perm], kpool[perm], inv[ids]
    T = c["k_codes"].shape[2]
    ids_t = torch.from_numpy(ids.astype(np.int64 if i64 else np.int32)).cuda()
    kp = ops.prepare_cents(t["k_cents"], cache=False)
    vp = ops.prepare_cents(t["v_cents"], cache=False)
    out = ops.pq_decode_attn(t["q"], torch.from_numpy(kpool).cuda() if k_paged else t["k_codes"],
                             torch.from_numpy(vpool).cuda(), kp, vp, t["k_res"], t["v_res"], c["r"], M=M, C=C,
                             n_tokens=T, k_page_ids=ids_t if k_paged else None, v_page_ids=ids_t, page_size=ps, out=dst)
    torch.cuda.synchronize()
    return out.cpu().numpy()


# ---------------------------------------------------------------- encode: bit-exact -----------------
@pytest.mark.parametrize("case", synth.GOLDEN_ENCODE, ids=[c[0] for c in synth.GOLDEN_ENCODE])
def test_encode_bit_exact_small(case, env, oracle, golden_dir):
    torch, ops = env
    name, seed, bs, nhk, n, d, M, C = case
    c = synth.encode_case(seed, bs, nhk, n, d, M, C)
    codes = ops.pq_encode(torch.from_numpy(c["X"]).cuda(), torch.from_numpy(c["cents"]).cuda()).cpu().numpy()
    np.testing.assert_array_equal(codes, oracle.pq_encode(c["X"], c["cents"]))
    # and directly against the reference's own sa_encode_4d output (cdist form: near-tie flips only)
    ref = np.load(golden_dir / f"encode_{name}.npz")["codes"]
    assert np.count_nonzero(codes != ref) <= max(1, codes.size // 100000)


@pytest.mark.parametrize("d,M", [(128, 32), (128, 16), (64, 64), (128, 64)], ids=["d128M32", "d128M16", "d64M64", "d128M64"])
@pytest.mark.parametrize("use_prepared", [True, False], ids=["prepared-f32tab", "raw-f16tab"])
def test_encode_bulk_kernel_instances(d, M, use_prepared, env, oracle):
    """The BULK encode kernel (pq_encode_kernel<DM, F32TAB, uint8_t>, taken from 1024 waves on: a prompt-sized call) for
    every sub-vector width the BASELINE configs use - d_m = 4 (configs[4]: M = 32), 8, 1 and 2 - with the prepared
    (fp32 image, scalar operands) and the raw fp16 codebook, into all three destinations; n = 4096 tokens x 8 kv heads
    (the flush-sized LDS kernel serves only the small calls of the other encode tests)."""
    torch, ops = env
    from million_amd import _lib as L
    bs, nhk, n, C, ps = 1, 8, 4096 + 37, 256, 64
    c = synth.encode_case(3000 + d + M, bs, nhk, n, d, M, C)
    gold = oracle.pq_encode(c["X"], c["cents"])
    Xd, cd = torch.from_numpy(c["X"]).cuda(), torch.from_numpy(c["cents"]).cuda()
    out = torch.zeros(bs, nhk, n, M, dtype=torch.uint8, device="cuda")
    ops.pq_encode_into(Xd, cd, out, use_prepared=use_prepared)
    np.testing.assert_array_equal(out.cpu().numpy(), gold)
    n_pages = (n + ps - 1) // ps
    ids = torch.randperm(bs * nhk * n_pages).to(torch.int32).reshape(bs, nhk, n_pages).cuda()
    kpool = torch.zeros(bs * nhk * n_pages, ps, M, dtype=torch.uint8, device="cuda")
    vpool = torch.zeros(bs * nhk * n_pages, M, ps, dtype=torch.uint8, device="cuda")
    ops.pq_encode_into(Xd, cd, kpool, layout=L.MILLION_CODES_KPAGES, page_ids=ids, page_size=ps, use_prepared=use_prepared)
    ops.pq_encode_into(Xd, cd, vpool, layout=L.MILLION_CODES_VPAGES, page_ids=ids, page_size=ps, use_prepared=use_prepared)
    idn = ids.cpu().numpy()
    np.testing.assert_array_equal(oracle.pool_to_k_rowmajor(kpool.cpu().numpy(), idn, n), gold)
    np.testing.assert_array_equal(oracle.pool_to_v_rowmajor(vpool.cpu().numpy(), idn, n), gold)


def test_encode_128k_m32_prefill_hash(env, oracle, golden_dir):
    """BASELINE configs[4]'s prefill encode (131072 tokens, M = 32, d_m = 4; one kv head): SHA-256 of the HIP codes ==
    the hash tools/gen_golden.py computed from the oracle, which tests/test_oracle.py ties to the SHA-256 of the
    reference's own sa_encode_4d output through the manifest's flip list (3 near ties of 4 194 304 codes)."""
    torch, ops = env
    man = json.loads((golden_dir / "manifest.json").read_text())["encode_128k_m32"]
    name, seed, bs, nhk, n, d, M, C = synth.GOLDEN_ENCODE_128K
    c = synth.encode_case(seed, bs, nhk, n, d, M, C)
    codes = ops.pq_encode(torch.from_numpy(c["X"]).cuda(), torch.from_numpy(c["cents"]).cuda()).cpu().numpy()
    assert hashlib.sha256(codes.tobytes()).hexdigest() == man["sha256_direct_oracle_codes"]
    ref_like = codes.copy()
    for pos, ref_code in zip(man["diff_positions"], man["diff_reference_codes"]):
        ref_like[tuple(pos)] = ref_code
    assert hashlib.sha256(ref_like.tobytes()).hexdigest() == man["sha256_reference_cdist_codes"]


def test_encode_bit_exact_cfg1_hash(env, oracle, golden_dir):
    """BASELINE config 1 size (1,8,4096,128): SHA-256 of the codes == committed oracle hash."""
    torch, ops = env
    man = json.loads((golden_dir / "manifest.json").read_text())["encode_big"]
    name, seed, bs, nhk, n, d, M, C = synth.GOLDEN_ENCODE_BIG
    c = synth.encode_case(seed, bs, nhk, n, d, M, C)
    codes = ops.pq_encode(torch.from_numpy(c["X"]).cuda(), torch.from_numpy(c["cents"]).cuda()).cpu().numpy()
    assert hashlib.sha256(codes.tobytes()).hexdigest() == man["sha256_direct_oracle_codes"]


def test_encode_ties_and_layouts(env, oracle):
    torch, ops = env
    from million_amd import _lib as L
    rs = np.random.RandomState(11)
    M, C, d = 64, 256, 128
    cents = rs.standard_normal((M, C, 2)).astype(np.float16)
    cents[:, 200] = cents[:, 7]          # exact duplicates: lower index must win
    X = rs.standard_normal((2, 3, 150, d)).astype(np.float16)
    X[0, 0, :10] = cents[:, 7].reshape(-1)      # rows sitting exactly on the duplicated centroid
    gold = oracle.pq_encode(X, cents)
    Xd, cd = torch.from_numpy(X).cuda(), torch.from_numpy(cents).cuda()
    assert (gold[0, 0, :10] == 7).all()
    np.testing.assert_array_equal(ops.pq_encode(Xd, cd).cpu().numpy(), gold)
    # paged destinations: K pages row-major, V pages transposed, written at a token offset
    ps, t0 = 64, 64
    n_pages = (t0 + 150 + ps - 1) // ps
    ids = torch.arange(2 * 3 * n_pages, dtype=torch.int32).reshape(2, 3, n_pages).flip(2).contiguous().cuda()
    kpool = torch.zeros(2 * 3 * n_pages, ps, M, dtype=torch.uint8).cuda()
    vpool = torch.zeros(2 * 3 * n_pages, M, ps, dtype=torch.uint8).cuda()
    ops.pq_encode_into(Xd, cd, kpool, layout=L.MILLION_CODES_KPAGES, token_start=t0, page_ids=ids, page_size=ps)
    ops.pq_encode_into(Xd, cd, vpool, layout=L.MILLION_CODES_VPAGES, token_start=t0, page_ids=ids, page_size=ps)
    kp, vp, idn = kpool.cpu().numpy(), vpool.cpu().numpy(), ids.cpu().numpy()
    for b in range(2):
        for h in range(3):
            for t in range(150):
                tok = t0 + t
                pid = idn[b, h, tok // ps]
                assert (kp[pid, tok % ps] == gold[b, h, t]).all()
                assert (vp[pid, :, tok % ps] == gold[b, h, t]).all()
    # ring-buffer source rows + strided view (residual window flush)
    buf = torch.zeros(2, 3, 128, d, dtype=torch.float16).cuda()
    rows = (np.arange(64) + 100) % 128
    buf[:, :, rows] = Xd[:, :, :64]
    out = torch.zeros(2, 3, 64, M, dtype=torch.uint8).cuda()
    ops.pq_encode_into(buf, cd, out, n=64, x_row_start=100, x_row_mod=128)
    np.testing.assert_array_equal(out.cpu().numpy(), gold[:, :, :64])


# ---------------------------------------------------------------- attention -------------------------
@pytest.mark.parametrize("case", synth.GOLDEN_ATTN, ids=[c[0] for c in synth.GOLDEN_ATTN])
@pytest.mark.parametrize("force_generic", [False, True], ids=["auto", "generic"])
def test_attn_golden_rowmajor(case, force_generic, env, oracle, golden_dir):
    torch, ops = env
    name, seed, bs, nh, nhk, d, M, C, T, r = case
    c = synth.attn_case(seed, bs, nh, nhk, d, M, C, T, r)
    ops.set_force_generic(force_generic)
    try:
        out = _run_rowmajor(torch, ops, c, M, C)
    finally:
        ops.set_force_generic(False)
    gold = oracle.decode_attn(**c)
    _check(out, gold, name)
    fix = np.load(golden_dir / f"attn_{name}.npz")["out"]       # the reference's own formula, torch fp32
    _check(out, fix, name + " vs fixture")


@pytest.mark.parametrize("case", [c for c in synth.GOLDEN_ATTN if c[8] > 0 and c[5] == 128],
                         ids=[c[0] for c in synth.GOLDEN_ATTN if c[8] > 0 and c[5] == 128])
@pytest.mark.parametrize("ps", [32, 64, 128])
def test_attn_golden_paged(case, ps, env, oracle):
    torch, ops = env
    name, seed, bs, nh, nhk, d, M, C, T, r = case
    c = synth.attn_case(seed, bs, nh, nhk, d, M, C, T, r)
    gold = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, M, C, ps), gold, f"{name} paged ps={ps}")
    # the reference's 13-arg layout: row-major K, paged V, int64 page ids
    _check(_run_paged(torch, ops, oracle, c, M, C, ps, k_paged=False, i64=True), gold, f"{name} mixed ps={ps}")


@pytest.mark.parametrize("T,r,nh,nhk,M", [(4096, 17, 8, 8, 64), (4097, 128, 32, 8, 64), (5000, 1, 8, 2, 32),
                                          (33, 0, 4, 4, 64), (0, 128, 32, 8, 64), (2048 + 31, 64, 16, 8, 64)])
def test_attn_random_shapes(T, r, nh, nhk, M, env, oracle):
    torch, ops = env
    c = synth.attn_case(1000 + T + r, 1, nh, nhk, 128, M, 256, T, r)
    gold = oracle.decode_attn(**c)
    _check(_run_rowmajor(torch, ops, c, M, 256), gold, "rowmajor")
    if T:
        _check(_run_paged(torch, ops, oracle, c, M, 256, 64), gold, "paged")


def test_attn_resid_ring_and_repeat_calls(env, oracle):
    """Residual ring buffer (resid_start > 0), reuse of the workspace across calls, batch > 1."""
    torch, ops = env
    c = synth.attn_case(77, 2, 8, 2, 128, 64, 256, 700, 50)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    gold = oracle.decode_attn(**c)
    start = 100
    kr = torch.roll(t["k_res"], start, dims=2).contiguous()
    vr = torch.roll(t["v_res"], start, dims=2).contiguous()
    for _ in range(3):
        out = ops.pq_decode_attn(t["q"], t["k_codes"], t["v_codes"], kp, vp, kr, vr, c["r"], M=64, C=256,
                                 resid_start=start)
    torch.cuda.synchronize()
    _check(out.cpu().numpy(), gold, "ring")


def test_attn_peaked_softmax_forces_rescale(env, oracle):
    """One key matches the query strongly late in the sequence, so the running max jumps mid-stream
    (cdna_hip_programming.md rule 26: a data-dependent rescale branch needs an input that forces it)."""
    torch, ops = env
    c = synth.attn_case(88, 1, 8, 2, 128, 64, 256, 3000, 9)
    # make code 0 of every subspace point along q of head 0 (large positive score), and plant it at token 2500
    kc = c["k_cents"].astype(np.float32)
    q0 = c["q"][0, 0, 0].astype(np.float32).reshape(64, 2)
    kc[:, 0, :] = 3.0 * q0
    c["k_cents"] = kc.astype(np.float16)
    c["k_codes"][0, 0, 2500, :] = 0
    c["k_codes"][0, 0, 10, :32] = 0
    gold = oracle.decode_attn(**c)
    _check(_run_rowmajor(torch, ops, c, 64, 256), gold, "peaked rowmajor")
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 64), gold, "peaked paged")


@pytest.mark.parametrize("G,heads", [(16, (13, 5)), (9, (8,)), (12, (11, 0))], ids=["G16", "G9", "G12"])
def test_attn_peaked_softmax_rescale_above_eight_heads(G, heads, env, oracle):
    """Groups of 9..16 query heads per kv head run in ONE launch of the MFMA kernels (second register row of the value
    tile): a head of that second row (and one of the first) gets a key that matches it strongly late in the sequence, so
    its softmax reference moves mid-stream and the rescale of rows 8..15 is exercised."""
    torch, ops = env
    nhk, T = 2, 3000
    c = synth.attn_case(880 + G, 1, G * nhk, nhk, 128, 64, 256, T, 9)
    kc = c["k_cents"].astype(np.float32)
    for i, h in enumerate(heads):                      # code i of every subspace points along q of head h (kv head 0)
        kc[:, i, :] = 3.0 * c["q"][0, h, 0].astype(np.float32).reshape(64, 2)
    c["k_cents"] = kc.astype(np.float16)
    for i, h in enumerate(heads):
        c["k_codes"][0, 0, 2500 - 700 * i, :] = i
        c["k_codes"][0, 0, 10 + i, :32] = i
    gold = oracle.decode_attn(**c)
    _check(_run_rowmajor(torch, ops, c, 64, 256), gold, f"peaked G={G} rowmajor")
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 64), gold, f"peaked G={G} paged")


def test_bindings_module_dropin(env, oracle):
    """The reference's call sequence (pq_utils.py:61-94; test_kernel.py:45-69) through `bindings`."""
    torch, ops = env
    import bindings
    bs, nh, d, M, C, T, r, Ns = 1, 32, 128, 64, 256, 1000, 17, 16
    c = synth.attn_case(5, bs, nh, nh, d, M, C, T, r)
    t = _dev(torch, c)
    fn = getattr(__import__("bindings"), f"flash_decoding_allocated_buffer_f16u8_Ns{Ns}Lt{d}d{d}M{M}C{C}")
    po = torch.empty(bs, nh, Ns + 1, d, dtype=torch.float16, device="cuda")
    pl = torch.empty(bs, nh, Ns + 1, dtype=torch.float16, device="cuda")
    out = fn(t["q"], t["k_codes"], t["v_codes"], t["k_cents"], t["v_cents"], t["k_res"], t["v_res"], r, po, pl)
    assert out.shape == (bs, nh, 1, d) and out.dtype == torch.float16
    gold = oracle.decode_attn(**c)
    _check(out.cpu().numpy(), gold, "bindings 10-arg")
    # residual views sliced to r rows (paged_pq_utils.py:421)
    out2 = fn(t["q"], t["k_codes"], t["v_codes"], t["k_cents"], t["v_cents"], t["k_res"][:, :, :r], t["v_res"][:, :, :r], r, po, pl)
    _check(out2.cpu().numpy(), gold, "bindings residual views")
    # 13-arg paged call (paged_pq_utils.py:621-635)
    ps = 64
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    pfn = getattr(bindings, f"flash_decoding_paged_v_f16u8_Ns{Ns}Lt128d{d}M{M}C{C}")
    out3 = pfn(t["q"], t["k_codes"], t["k_cents"], t["k_res"][:, :, :r], torch.from_numpy(ids).cuda(),
               torch.from_numpy(vpool).cuda(), t["v_cents"], t["v_res"][:, :, :r], r, ids.shape[2], ps, po, pl)
    _check(out3.cpu().numpy(), gold, "bindings 13-arg")
    enc = getattr(bindings, f"pq_encode_f16u8_d{d}M{M}C{C}")
    e = synth.encode_case(9, 1, 2, 70, d, M, C)
    np.testing.assert_array_equal(enc(torch.from_numpy(e["X"]).cuda(), torch.from_numpy(e["cents"]).cuda()).cpu().numpy(),
                                  oracle.pq_encode(e["X"], e["cents"]))
    with pytest.raises(RuntimeError):
        fn(t["q"].float(), t["k_codes"], t["v_codes"], t["k_cents"], t["v_cents"], t["k_res"], t["v_res"], r, po, pl)


def test_residual_append_and_dev_lengths(env, oracle):
    torch, ops = env
    c = synth.attn_case(31, 2, 8, 2, 128, 64, 256, 300, 20)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    rs = np.random.RandomState(1)
    k_new = rs.standard_normal((2, 2, 1, 128)).astype(np.float16)
    v_new = rs.standard_normal((2, 2, 1, 128)).astype(np.float16)
    lengths = torch.tensor([[300, 20, 0, 0], [300, 20, 0, 0]], dtype=torch.int32, device="cuda")
    kr, vr = t["k_res"].clone(), t["v_res"].clone()
    ops.residual_append(torch.from_numpy(k_new).cuda(), torch.from_numpy(v_new).cuda(), kr, vr, 0, 0, dev_lengths=lengths)
    out = ops.pq_decode_attn(t["q"], t["k_codes"], t["v_codes"], kp, vp, kr, vr, 0, M=64, C=256, dev_lengths=lengths)
    torch.cuda.synchronize()
    assert lengths.cpu().numpy()[:, 1].tolist() == [21, 21]
    c2 = dict(c)
    c2["k_res"] = c["k_res"].copy(); c2["v_res"] = c["v_res"].copy()
    c2["k_res"][:, :, 20] = k_new[:, :, 0]; c2["v_res"][:, :, 20] = v_new[:, :, 0]
    c2["r"] = 21
    _check(out.cpu().numpy(), oracle.decode_attn(**c2), "append + device lengths")


@pytest.mark.parametrize("paged", [False, True], ids=["rowmajor-generic", "paged-mfma"])
@pytest.mark.parametrize("use_dl", [False, True], ids=["host-lengths", "device-lengths"])
def test_fused_append(paged, use_dl, env, oracle):
    """million_pq_decode_attn_append: the new token's row is attended to AND parked in the window
    (replaces pq_utils.py:304-312 + :314-326 in one launch); with device-resident lengths r advances."""
    torch, ops = env
    bs, nh, nhk, T, r0, ps = 2, 8, 2, 700, 37, 64
    c = synth.attn_case(55, bs, nh, nhk, 128, 64, 256, T, r0)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    rs = np.random.RandomState(2)
    start = 100 if paged else 0
    kr = torch.roll(t["k_res"], start, dims=2).contiguous()
    vr = torch.roll(t["v_res"], start, dims=2).contiguous()
    kw = {}
    if paged:
        vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
        kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
        ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
        kc, vc = torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda()
        kw = dict(k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps, n_tokens=T)
    else:
        kc, vc = t["k_codes"], t["v_codes"]
    lengths = torch.tensor([[T, r0, start, 0]] * bs, dtype=torch.int32, device="cuda") if use_dl else None
    k_hist, v_hist = c["k_res"].copy(), c["v_res"].copy()
    r = r0
    for step in range(3):
        k_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        v_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, kr, vr, 0 if use_dl else r, M=64, C=256, resid_start=start,
                                 dev_lengths=lengths, k_new=torch.from_numpy(k_new).cuda(),
                                 v_new=torch.from_numpy(v_new).cuda(), **kw)
        torch.cuda.synchronize()
        k_hist[:, :, r], v_hist[:, :, r] = k_new[:, :, 0], v_new[:, :, 0]
        r += 1
        c2 = dict(c, k_res=k_hist, v_res=v_hist, r=r)
        _check(out.cpu().numpy(), oracle.decode_attn(**c2), f"fused append step {step}")
    # the rows were parked in the ring at (start + r0 + i) % cap
    got = torch.roll(kr, -start, dims=2).cpu().numpy()
    np.testing.assert_array_equal(got[:, :, r0:r], k_hist[:, :, r0:r])
    if use_dl:
        assert lengths.cpu().numpy()[:, 1].tolist() == [r] * bs


@pytest.mark.parametrize("M", [32, 16], ids=["M32-d4form", "M16-d8form"])
@pytest.mark.parametrize("nh,nhk", [(2, 2), (4, 2), (6, 2), (8, 2), (16, 2), (10, 2), (14, 2), (13, 1)], ids=["G1", "G2", "G3", "G4", "G8", "G5", "G7", "G13"])
def test_attn_replicated_head_forms(M, nh, nhk, env, oracle):
    """The streaming kernel's d_m = 4 / d_m = 8 forms (M = 32 / 16 at up to 4 query heads per kv head: query heads replicated over
    the column groups of the score tile, gathered V entries as the value product's B operand): every group size 1..4, paged and
    row-major, ring start > 0, three fused-append steps with device lengths; 8 heads per kv head take the packed form (M = 32) /
    run as two virtual kv heads of 4 query heads (M = 16, round 5; before: the tile kernel) and must agree too."""
    torch, ops = env
    bs, T, r0, ps, C = 2, 2500, 37, 64, 256
    c = synth.attn_case(7700 + M + nh, bs, nh, nhk, 128, M, C, T, r0, Lt=128)
    gold = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, M, C, ps), gold, "paged")
    _check(_run_rowmajor(torch, ops, c, M, C), gold, "rowmajor")
    t = _dev(torch, c)
    desc = ops.make_attn_desc(t["q"], t["k_res"], nh_k=nhk, M=M, C=C, n_tokens=T, r=r0, k_paged=True, v_paged=True,
                              page_size=ps, n_pages_cap=(T + ps - 1) // ps)
    from million_amd import _lib
    assert _lib.load().million_attn_kernel_kind(ctypes.byref(desc)) == 1      # (M = 16 at 5 .. 16 heads per kv head: ceil(G / 4) virtual kv heads, round 5)
    # fused append over three steps, device lengths, ring start 100
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    rs = np.random.RandomState(3)
    start = 100
    kr = torch.roll(t["k_res"], start, dims=2).contiguous()
    vr = torch.roll(t["v_res"], start, dims=2).contiguous()
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    kc, vc = torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda()
    lengths = torch.tensor([[T, r0, start, 0]] * bs, dtype=torch.int32, device="cuda")
    k_hist, v_hist = c["k_res"].copy(), c["v_res"].copy()
    r = r0
    for step in range(3):
        k_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        v_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, kr, vr, 0, M=M, C=C, resid_start=start, dev_lengths=lengths,
                                 k_new=torch.from_numpy(k_new).cuda(), v_new=torch.from_numpy(v_new).cuda(),
                                 k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps, n_tokens=T)
        torch.cuda.synchronize()
        k_hist[:, :, r], v_hist[:, :, r] = k_new[:, :, 0], v_new[:, :, 0]
        r += 1
        _check(out.cpu().numpy(), oracle.decode_attn(**dict(c, k_res=k_hist, v_res=v_hist, r=r)), f"fused append step {step}")
    assert lengths.cpu().numpy()[:, 1].tolist() == [r] * bs


def test_prepared_codebook_cache_is_not_keyed_on_address(env, oracle):
    """Regression: a new codebook allocated where a freed one lived must not reuse its prepared table."""
    torch, ops = env
    import bindings
    c = synth.attn_case(91, 1, 8, 2, 128, 64, 256, 300, 9)
    t = _dev(torch, c)
    fn = bindings.flash_decoding_allocated_buffer_f16u8_Ns16Lt128d128M64C256
    po = torch.empty(1, 8, 17, 128, dtype=torch.float16, device="cuda")
    pl = torch.empty(1, 8, 17, dtype=torch.float16, device="cuda")
    for seed in range(4):
        rs = np.random.RandomState(seed)
        kc = rs.standard_normal((64, 256, 2)).astype(np.float16)
        c2 = dict(c, k_cents=kc)
        kct = torch.from_numpy(kc).cuda()          # freed at the end of the iteration: the address is reused
        out = fn(t["q"], t["k_codes"], t["v_codes"], kct, t["v_cents"], t["k_res"], t["v_res"], c["r"], po, pl)
        _check(out.cpu().numpy(), oracle.decode_attn(**c2), f"codebook {seed}")
        del kct


def test_dynamic_cache_decode_sequence(env, oracle):
    """DynamicPQCache (row-major store, flush all Lt=128 rows when full, pq_utils.py:281-328) over a
    prefill + 300 decode steps: token accounting == the policy oracle, outputs == the attention oracle with
    codes produced by the encode oracle."""
    torch, ops = env
    from million_amd.pq_cache import DynamicPQCache
    bs, nh, nhk, d, M, C = 1, 8, 2, 128, 64, 256
    rs = np.random.RandomState(3)
    ck = rs.standard_normal((M, C, 2)).astype(np.float16)
    cv = rs.standard_normal((M, C, 2)).astype(np.float16)
    n_prompt, n_dec = 100, 300
    K = rs.standard_normal((bs, nhk, n_prompt + n_dec, d)).astype(np.float16)
    V = rs.standard_normal((bs, nhk, n_prompt + n_dec, d)).astype(np.float16)
    Q = rs.standard_normal((n_dec, bs, nh, 1, d)).astype(np.float16)
    cache = DynamicPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, max_tokens=1024)
    cache.set_cent(torch.from_numpy(ck).cuda(), torch.from_numpy(cv).cuda())
    Kd, Vd = torch.from_numpy(K).cuda(), torch.from_numpy(V).cuda()
    cache.prefill(torch.from_numpy(Q[0]).cuda().repeat(1, 1, n_prompt, 1), Kd[:, :, :n_prompt].contiguous(),
                  Vd[:, :, :n_prompt].contiguous(), 0)
    pol = oracle.DynamicPolicy(Lt=128, prefill=n_prompt)
    for i in range(n_dec):
        t = n_prompt + i
        out = cache.decoding(torch.from_numpy(Q[i]).cuda(), Kd[:, :, t:t + 1].contiguous(), Vd[:, :, t:t + 1].contiguous(), 0)
        T, r = pol.step()
        assert (cache._T[0], cache.residualed_tokens[0]) == (T, r)
        assert cache.key_cache[0].shape == (bs, nhk, T, M)
        if i % 37 == 0 or i == n_dec - 1:
            kc, vc = oracle.pq_encode(K[:, :, :T], ck), oracle.pq_encode(V[:, :, :T], cv)
            np.testing.assert_array_equal(cache.key_cache[0].cpu().numpy(), kc)
            kres = np.zeros((bs, nhk, 128, d), np.float16)
            vres = np.zeros((bs, nhk, 128, d), np.float16)
            kres[:, :, :r], vres[:, :, :r] = K[:, :, T:T + r], V[:, :, T:T + r]
            _check(out.cpu().numpy(), oracle.decode_attn(Q[i], kc, vc, ck, cv, kres, vres, r), f"dynamic step {i}")


def test_rows_reduce_selfcheck(env):
    """The kernel's VALU-only row reductions (v_permlane16_swap / v_permlane32_swap) against numpy: guards the
    hipcc pitfall of passing one SSA value as both swap operands (see attn_mfma.hip)."""
    torch, ops = env
    from million_amd import _lib as L
    lib = L.load()
    rs = np.random.RandomState(4)
    x = rs.standard_normal(64).astype(np.float32)
    x[5], x[37], x[60] = 1000.0, -7.0, np.float32(-np.inf)
    xd = torch.from_numpy(x).cuda()
    om, osum = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    assert lib.million_debug_rows_reduce(xd.data_ptr(), om.data_ptr(), osum.data_ptr(), None) == 0
    torch.cuda.synchronize()
    xr = x.reshape(4, 16)
    np.testing.assert_array_equal(om.cpu().numpy().reshape(4, 16), np.broadcast_to(xr.max(0), (4, 16)))
    want = (xr[0] + xr[1]) + (xr[2] + xr[3])
    np.testing.assert_array_equal(osum.cpu().numpy().reshape(4, 16), np.broadcast_to(want, (4, 16)))


def test_harness_graph_replay_matches_eager(env):
    """million_amd/harness.py: the whole decode step replayed from two hipGraphs (device-side lengths, page
    flushes inside the graph) generates the same tokens as the eager PagedPQCache path, across two flushes."""
    torch, ops = env
    from million_amd import harness as H
    shape = H.LlamaShape(hidden=256, n_layers=2, nh=32, nh_k=8, d=128, inter=512, vocab=1000)
    dev = torch.device("cuda", 0)
    model = H.LlamaShapeDecoder(shape, dev, seed=1)
    steps, ctx = 200, 1024
    toks = {}
    for mode in ("eager", "graph"):
        be = H.PQBackend(shape, 1, ctx, steps + 8, dev)
        tokens = torch.full((1,), 7, dtype=torch.long, device=dev)
        pos = torch.full((1,), ctx, dtype=torch.long, device=dev)
        seq = []
        if mode == "graph":
            gd = H.GraphedPQDecoder(model, be, tokens, pos)
        for i in range(steps):
            if mode == "graph":
                gd.step()
                seq.append(int(tokens.item()))
                tokens.fill_(toks["eager"][i])      # teacher forcing: an fp near-tie must not fork the sequence
            else:
                tokens.copy_(model.step(tokens, pos, be))
                pos.add_(1)
                seq.append(int(tokens.item()))
        toks[mode] = seq
        assert be.cache._T[0] == ctx + 128 and be.cache.residualed_tokens[0] == steps - 128
        assert int(pos.item()) == ctx + steps
    same = sum(a == b for a, b in zip(toks["eager"], toks["graph"]))
    assert same >= steps - 4, f"graph replay diverges from eager: {same}/{steps} tokens equal"
    assert len(set(toks["eager"])) > 4          # not a degenerate constant sequence


# ---- long contexts: the pipelined MFMA kernel (splits of 25..40 units of 32 tokens; 32 splits per kv head at
#      bs * nh_k = 8 on a 256-CU part) --------------------------------------------------------------------------
@pytest.mark.parametrize("T,r,nh,nhk", [
    (32768, 100, 32, 8),    # headline shape: 4 units per wave, no phantom units
    (32832, 128, 32, 8),    # one page more: waves 0-1 of every split carry a fifth unit
    (24608, 3, 32, 8),      # 26 units per split: waves with 3 units run one masked phantom unit
    (40960, 77, 8, 8),      # 40 units per split: every wave has five; G = 1
    (33000, 128, 64, 8),    # ragged end inside a unit; G = 8
    (40961, 5, 32, 8),      # 41 units per split: waves 0-1 carry a sixth unit (one whole round of four + two single units)
])
def test_attn_long_context(T, r, nh, nhk, env, oracle):
    torch, ops = env
    c = synth.attn_case(4000 + T % 977 + r, 1, nh, nhk, 128, 64, 256, T, r)
    gold = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 64), gold, f"paged T={T}")
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 128, k_paged=False, i64=True, shuffle=False), gold, f"mixed T={T}")


def test_fused_append_long_context_device_lengths(env, oracle):
    """Fused append + device-resident lengths on the pipelined kernel, crossing the 4 -> 5 units per wave boundary
    (the host bound n_tokens sizes the grid; the kernel reads T, r, start from the device)."""
    torch, ops = env
    bs, nh, nhk, ps, cap = 1, 32, 8, 64, 128
    T0, r0, start = 32768, 126, 70
    c = synth.attn_case(4242, bs, nh, nhk, 128, 64, 256, T0 + 64, cap)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    kc, vc = torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda()
    kr = torch.roll(t["k_res"], start, dims=2).contiguous()
    vr = torch.roll(t["v_res"], start, dims=2).contiguous()
    rs = np.random.RandomState(9)
    k_hist, v_hist = c["k_res"].copy(), c["v_res"].copy()
    for T, r in ((T0, r0), (T0 + 64, 60)):      # before / after a flush of one page (lengths rewritten on the device)
        lengths = torch.tensor([[T, r, start, 0]] * bs, dtype=torch.int32, device="cuda")
        k_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        v_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, kr, vr, 0, M=64, C=256, resid_start=start, dev_lengths=lengths,
                                 k_new=torch.from_numpy(k_new).cuda(), v_new=torch.from_numpy(v_new).cuda(),
                                 k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps, n_tokens=T0 + 64)
        torch.cuda.synchronize()
        k_hist[:, :, r], v_hist[:, :, r] = k_new[:, :, 0], v_new[:, :, 0]
        c2 = dict(c, k_codes=c["k_codes"][:, :, :T], v_codes=c["v_codes"][:, :, :T], k_res=k_hist, v_res=v_hist, r=r + 1)
        _check(out.cpu().numpy(), oracle.decode_attn(**c2), f"T={T} r={r}")
        assert lengths.cpu().numpy()[:, 1].tolist() == [r + 1] * bs


@pytest.mark.parametrize("shape,M,C,dm", [((2, 3, 257), 64, 256, 2), ((1, 8, 4096), 32, 256, 4), ((5,), 16, 128, 4),
                                          ((1, 1, 0), 64, 256, 2), ((7, 33), 16, 256, 8)])
def test_pq_decode_exact(shape, M, C, dm, env, oracle):
    """million_pq_decode = sa_decode_4d (pq_utils.py:501-540): an exact gather, compared with the oracle's."""
    torch, ops = env
    rs = np.random.RandomState(M + dm)
    codes = rs.randint(0, C, size=shape + (M,)).astype(np.uint8)
    cents = rs.standard_normal((M, C, dm)).astype(np.float16)
    out = ops.pq_decode(torch.from_numpy(codes).cuda(), torch.from_numpy(cents).cuda())
    torch.cuda.synchronize()
    gold = oracle.pq_decode_numpy(codes.reshape(1, 1, -1, M), cents).reshape(shape + (M * dm,))
    assert out.shape == gold.shape and out.dtype == torch.float16
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint16), gold.astype(np.float16).view(np.uint16))


# ---- M = 32 (d_m = 4, BASELINE configs[4]) on the MFMA kernel ---------------------------------------------------
@pytest.mark.parametrize("T,r,nh,nhk,bs", [(257, 17, 8, 2, 1), (4096, 128, 32, 8, 1), (5000, 1, 8, 8, 2), (33, 0, 64, 8, 1),
                                           (0, 77, 32, 8, 1), (65536 + 40, 100, 32, 8, 1)])
def test_attn_m32_mfma(T, r, nh, nhk, bs, env, oracle):
    torch, ops = env
    from million_amd import _lib
    c = synth.attn_case(7000 + T % 991 + r, bs, nh, nhk, 128, 32, 256, T, r)
    gold = oracle.decode_attn(**c)
    _check(_run_rowmajor(torch, ops, c, 32, 256), gold, "M=32 rowmajor (transpose + MFMA)")
    if T:
        _check(_run_paged(torch, ops, oracle, c, 32, 256, 64), gold, "M=32 paged")
        _check(_run_paged(torch, ops, oracle, c, 32, 256, 32, k_paged=False, i64=True), gold, "M=32 mixed ps=32")
    # and it really is the MFMA path
    t = _dev(torch, c)
    desc = ops.make_attn_desc(t["q"], t["k_res"], nh_k=nhk, M=32, C=256, n_tokens=max(T, 1), r=r, k_paged=True, v_paged=True,
                              page_size=64, n_pages_cap=(max(T, 1) + 63) // 64)
    assert _lib.load().million_attn_kernel_kind(ctypes.byref(desc)) == 1


def test_dynamic_cache_update_and_distort_recent(env, oracle):
    """DynamicPQCache.update / prefill(distort_recent=True): the reference's dequantise-then-attend path
    (pq_utils.py:166-260) on the HIP encode + decode kernels, against the oracle's encode/decode."""
    torch, ops = env
    from million_amd.pq_cache import DynamicPQCache
    bs, nh, nhk, M, d = 2, 8, 2, 64, 128
    rs = np.random.RandomState(31)
    cents_k = rs.standard_normal((M, 256, d // M)).astype(np.float16)
    cents_v = rs.standard_normal((M, 256, d // M)).astype(np.float16)
    cache = DynamicPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, max_tokens=512)
    cache.set_cent(torch.from_numpy(cents_k).cuda(), torch.from_numpy(cents_v).cuda())
    k1, v1 = rs.standard_normal((bs, nhk, 70, d)).astype(np.float16), rs.standard_normal((bs, nhk, 70, d)).astype(np.float16)
    k2, v2 = rs.standard_normal((bs, nhk, 3, d)).astype(np.float16), rs.standard_normal((bs, nhk, 3, d)).astype(np.float16)
    dq = lambda x, c: oracle.pq_decode_numpy(oracle.pq_encode(x, c), c)
    # first update: nothing stored yet -> the inputs come back unchanged
    K, V = cache.update(torch.from_numpy(k1).cuda(), torch.from_numpy(v1).cuda(), 0)
    np.testing.assert_array_equal(K.cpu().numpy(), k1)
    # second update: dequantised past ++ raw new rows
    K, V = cache.update(torch.from_numpy(k2).cuda(), torch.from_numpy(v2).cuda(), 0)
    np.testing.assert_array_equal(K.cpu().numpy(), np.concatenate([dq(k1, cents_k), k2], axis=2))
    np.testing.assert_array_equal(V.cpu().numpy(), np.concatenate([dq(v1, cents_v), v2], axis=2))
    assert cache.key_cache[0].shape == (bs, nhk, 73, M) and cache.seen_tokens[0] == 73
    # distort_recent: everything dequantised, the new rows too
    k3, v3 = rs.standard_normal((bs, nhk, 2, d)).astype(np.float16), rs.standard_normal((bs, nhk, 2, d)).astype(np.float16)
    K, V = cache.update(torch.from_numpy(k3).cuda(), torch.from_numpy(v3).cuda(), 0, distort_recent=True)
    np.testing.assert_array_equal(V.cpu().numpy(), np.concatenate([dq(v1, cents_v), dq(v2, cents_v), dq(v3, cents_v)], axis=2))
    # prefill(distort_recent=True) == causal SDPA over the dequantised prompt
    cache2 = DynamicPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, max_tokens=512)
    cache2.set_cent(torch.from_numpy(cents_k).cuda(), torch.from_numpy(cents_v).cuda())
    q = rs.standard_normal((bs, nh, 70, d)).astype(np.float16)
    out = cache2.prefill(torch.from_numpy(q).cuda(), torch.from_numpy(k1).cuda(), torch.from_numpy(v1).cuda(), 0, distort_recent=True)
    kd = torch.from_numpy(dq(k1, cents_k)).float().repeat_interleave(nh // nhk, dim=1)
    vd = torch.from_numpy(dq(v1, cents_v)).float().repeat_interleave(nh // nhk, dim=1)
    ref = torch.nn.functional.scaled_dot_product_attention(torch.from_numpy(q).float(), kd, vd, is_causal=True)
    _check(out.float().cpu().numpy(), ref.numpy(), "prefill distort_recent")


@pytest.mark.parametrize("use_dl", [False, True], ids=["host-lengths", "device-lengths"])
def test_paged_cache_pipeline(use_dl, env, oracle):
    """PagedPQCache end to end (paged_pq_utils.py:216-386): bulk prefill encode straight into pages (prompt length not
    a multiple of the page), 150 decode steps with fused append, two ring flushes that straddle pages; batch 2, GQA,
    two layers; every 25th step against the oracle fed with the oracle's own codes."""
    torch, ops = env
    from million_amd.pq_cache import PagedPQCache
    bs, nh, nhk, d, M, C, ps, L_ = 2, 8, 2, 128, 64, 256, 64, 2
    n_prompt, n_dec = 3000, 150
    rs = np.random.RandomState(12)
    ck, cv = rs.standard_normal((M, C, 2)).astype(np.float16), rs.standard_normal((M, C, 2)).astype(np.float16)
    K = rs.standard_normal((L_, bs, nhk, n_prompt + n_dec, d)).astype(np.float16)
    V = rs.standard_normal((L_, bs, nhk, n_prompt + n_dec, d)).astype(np.float16)
    Q = rs.standard_normal((n_dec, L_, bs, nh, 1, d)).astype(np.float16)
    cache = PagedPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=L_, d=d, page_size=ps,
                         extended_residual_size=128, max_tokens=n_prompt + n_dec + 256)
    cache.set_cent(torch.from_numpy(ck).cuda(), torch.from_numpy(cv).cuda())
    Kd, Vd = torch.from_numpy(K).cuda(), torch.from_numpy(V).cuda()
    for l in range(L_):
        qp = torch.from_numpy(rs.standard_normal((bs, nh, n_prompt, d)).astype(np.float16)).cuda()
        cache.prefill(qp, Kd[l, :, :, :n_prompt].contiguous(), Vd[l, :, :, :n_prompt].contiguous(), l)
    pol = oracle.PagedPolicy(page_size=ps, residual=128, prefill=n_prompt)
    for i in range(n_dec):
        t = n_prompt + i
        outs = [cache.decoding_with_pages(torch.from_numpy(Q[i, l]).cuda(), Kd[l, :, :, t:t + 1].contiguous(),
                                          Vd[l, :, :, t:t + 1].contiguous(), l, use_dev_lengths=use_dl) for l in range(L_)]
        T, r = pol.step()
        assert (cache._T[0], cache.residualed_tokens[0]) == (T, r)
        if i % 25 == 0 or i == n_dec - 1:
            for l in range(L_):
                kc, vc = oracle.pq_encode(K[l, :, :, :T], ck), oracle.pq_encode(V[l, :, :, :T], cv)
                kres, vres = np.zeros((bs, nhk, 128, d), np.float16), np.zeros((bs, nhk, 128, d), np.float16)
                kres[:, :, :r], vres[:, :, :r] = K[l, :, :, T:T + r], V[l, :, :, T:T + r]
                _check(outs[l].float().cpu().numpy(), oracle.decode_attn(Q[i, l], kc, vc, ck, cv, kres, vres, r),
                       f"step {i} layer {l}")
    if use_dl:
        assert cache.lengths[0].cpu().numpy()[:, :2].tolist() == [[cache._T[0], cache.residualed_tokens[0]]] * bs


@pytest.mark.parametrize("mode", ["host-lengths", "device-lengths", "graphs", "on-demand", "spread", "spread-graphs"])
def test_paged_cache_encode_ahead_equals_inline_flush(mode, env, oracle):
    """PagedPQCache.begin_step: the oldest window page of all layers is encoded by ONE launch (million_pq_flush_layers,
    advance = 0) a few steps after the previous flush, the flush step only commits the lengths.  Same codes, same pages,
    same lengths and bit-identical attention outputs as the in-line flush of the reference's schedule
    (paged_pq_utils.py:359-361), over two flush periods; every kind of step occurs; 'graphs': one hipGraph per kind of
    step with device-resident lengths, replayed."""
    torch, ops = env
    from million_amd.pq_cache import PagedPQCache
    bs, nh, nhk, d, M, C, ps, L_ = 2, 8, 2, 128, 64, 256, 64, 3
    n_prompt, n_dec = 700, 200      # the window fills at step 128, then every 64 steps: two flush periods
    rs = np.random.RandomState(21)
    ck, cv = rs.standard_normal((M, C, 2)).astype(np.float16), rs.standard_normal((M, C, 2)).astype(np.float16)
    K = rs.standard_normal((L_, bs, nhk, n_prompt + n_dec, d)).astype(np.float16)
    V = rs.standard_normal((L_, bs, nhk, n_prompt + n_dec, d)).astype(np.float16)
    Q = rs.standard_normal((n_dec, L_, bs, nh, 1, d)).astype(np.float16)
    Kd, Vd, Qd = torch.from_numpy(K).cuda(), torch.from_numpy(V).cuda(), torch.from_numpy(Q).cuda()
    use_dl = mode in ("device-lengths", "graphs", "spread-graphs")
    spread = 2 if mode.startswith("spread") else 1      # the layers' encode-ahead over two steps (large batches do that)

    def make():
        c = PagedPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=L_, d=d, page_size=ps,
                         extended_residual_size=128, max_tokens=n_prompt + n_dec + 256, preallocate=mode != "on-demand")
        c.set_cent(torch.from_numpy(ck).cuda(), torch.from_numpy(cv).cuda())
        for l in range(L_):
            c.prefill(Qd[0, l].expand(-1, -1, 1, -1).repeat(1, 1, n_prompt, 1), Kd[l, :, :, :n_prompt].contiguous(),
                      Vd[l, :, :, :n_prompt].contiguous(), l)
        return c

    ref, cache = make(), make()
    cache.encode_ahead_steps = spread
    kn = [torch.empty(bs, nhk, 1, d, device="cuda", dtype=torch.float16) for _ in range(L_)]
    vn = [torch.empty_like(kn[0]) for _ in range(L_)]
    qs = [torch.empty(bs, nh, 1, d, device="cuda", dtype=torch.float16) for _ in range(L_)]
    outs = [torch.empty(bs, nh, 1, d, device="cuda", dtype=torch.float16) for _ in range(L_)]

    def step():
        cache.begin_step(use_dev_lengths=use_dl)
        for l in range(L_):
            cache.decoding_with_pages(qs[l], kn[l], vn[l], l, out=outs[l], use_dev_lengths=use_dl)

    graphs = {}
    if mode.endswith("graphs"):
        st = cache.host_state()
        dl_backup = cache._lengths_all.clone()
        win = (cache._kres_all.clone(), cache._vres_all.clone())
        step()                                   # eager once: allocates the workspace (restored below)
        torch.cuda.synchronize()
        for name, state in cache.capture_states(st):
            cache.set_host_state(state)
            assert cache.next_step_kind() == name
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            graphs[name] = g
        cache.set_host_state(st)
        cache._lengths_all.copy_(dl_backup)
        cache._kres_all.copy_(win[0]); cache._vres_all.copy_(win[1])
    kinds = []
    for i in range(n_dec):
        t = n_prompt + i
        for l in range(L_):
            kn[l].copy_(Kd[l, :, :, t:t + 1]); vn[l].copy_(Vd[l, :, :, t:t + 1]); qs[l].copy_(Qd[i, l])
        want = [ref.decoding_with_pages(qs[l], kn[l], vn[l], l, use_dev_lengths=use_dl).clone() for l in range(L_)]
        kind = cache.next_step_kind()
        kinds.append(kind)
        if mode.endswith("graphs"):
            graphs[kind].replay()
            cache.note_replayed_step(kind)
        else:
            step()
        torch.cuda.synchronize()
        for l in range(L_):
            assert torch.equal(outs[l], want[l]), f"step {i} ({kind}) layer {l}"
        assert (cache._T[0], cache.residualed_tokens[0], cache._rstart[0]) == (ref._T[0], ref.residualed_tokens[0], ref._rstart[0])
    pre = [i for i, k in enumerate(kinds) if k.startswith("pre")]
    assert len(pre) == 2 * spread and kinds.count("commit") == 2 and "flush" not in kinds
    assert pre[spread - 1] < kinds.index("commit") < pre[spread]
    if spread == 2:
        assert [kinds[i] for i in pre] == ["pre0", "pre1"] * 2
    T = ref._T[0]
    assert T == n_prompt + 2 * ps
    for l in range(L_):
        rk, rv = ref._codes_rowmajor(l, T)
        ck_, cv_ = cache._codes_rowmajor(l, T)
        assert torch.equal(rk, ck_) and torch.equal(rv, cv_)
        np.testing.assert_array_equal(ck_.cpu().numpy(), oracle.pq_encode(K[l, :, :, :T], ck))
    if use_dl:
        assert torch.equal(cache._lengths_all, ref._lengths_all)


@pytest.mark.parametrize("G", [3, 5, 6, 7])
def test_attn_odd_group_sizes(G, env, oracle):
    """GQA group sizes that are not powers of two (scalar last-arriver combine, partially filled MFMA columns)."""
    torch, ops = env
    nhk = 2
    c = synth.attn_case(900 + G, 1, G * nhk, nhk, 128, 64, 256, 2100, 77)
    gold = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 64), gold, f"G={G} paged")
    _check(_run_rowmajor(torch, ops, c, 64, 256), gold, f"G={G} rowmajor")


def test_attn_nothing_to_attend(env):
    """T = 0 and r = 0: the reference divides 0 by 0 (NaN); here the documented result is zeros."""
    torch, ops = env
    c = synth.attn_case(5, 2, 8, 2, 128, 64, 256, 0, 0)
    out = _run_rowmajor(torch, ops, c, 64, 256)
    assert np.array_equal(out, np.zeros_like(out))


def test_paged_cache_on_demand_pages_and_stats(env, oracle):
    """PagedPQCache(preallocate=False): pages are taken from the PageManager as tokens arrive (prefill and each
    flush), the result is the same as with a preallocated table, and get_cache_stats reports the reference's keys
    (paged_pq_utils.py:898-939)."""
    torch, ops = env
    from million_amd.pq_cache import PagedPQCache
    bs, nh, nhk, d, M, ps = 1, 8, 2, 128, 64, 64
    rs = np.random.RandomState(21)
    ck, cv = rs.standard_normal((M, 256, 2)).astype(np.float16), rs.standard_normal((M, 256, 2)).astype(np.float16)
    n_prompt, n_dec = 130, 140
    K = torch.from_numpy(rs.standard_normal((bs, nhk, n_prompt + n_dec, d)).astype(np.float16)).cuda()
    V = torch.from_numpy(rs.standard_normal((bs, nhk, n_prompt + n_dec, d)).astype(np.float16)).cuda()
    Q = torch.from_numpy(rs.standard_normal((n_dec, bs, nh, 1, d)).astype(np.float16)).cuda()
    outs = {}
    for pre in (True, False):
        cache = PagedPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, page_size=ps,
                             extended_residual_size=128, max_tokens=1024, preallocate=pre)
        cache.set_cent(torch.from_numpy(ck).cuda(), torch.from_numpy(cv).cuda())
        if not pre:
            assert cache.page_manager.get_stats()["allocated_pages"] == 0
        cache.prefill(Q[0].expand(-1, -1, 1, -1).repeat(1, 1, n_prompt, 1), K[:, :, :n_prompt].contiguous(),
                      V[:, :, :n_prompt].contiguous(), 0)
        o = []
        for i in range(n_dec):
            t = n_prompt + i
            o.append(cache.decoding_with_pages(Q[i], K[:, :, t:t + 1].contiguous(), V[:, :, t:t + 1].contiguous(), 0))
        outs[pre] = torch.stack(o).float().cpu().numpy()
        st = cache.get_cache_stats()
        assert st["layer_stats"][0]["seen_tokens"] == n_prompt + n_dec
        assert st["layer_stats"][0]["key_cache_tokens"] == cache._T[0] == n_prompt + 64      # one flush (step 129)
        assert st["layer_stats"][0]["residual_tokens"] == cache.residualed_tokens[0]
        assert set(st["memory_breakdown"]) >= {"cache_memory_mb", "residual_memory_mb", "total_memory_mb"}
        if not pre:      # 194 tokens -> 4 pages per (b, hk), taken only when needed
            assert st["page_manager"]["allocated_pages"] == 4 * bs * nhk
    np.testing.assert_array_equal(outs[True], outs[False])


# ---- splits of >= 64 units (several ring refills per wave) --------------------------------------------------
@pytest.mark.parametrize("T,r,nh,nhk,bs", [
    (65536, 100, 32, 8, 1),      # 64 units per split: 8 per wave, two rounds, no phantom units
    (66000, 128, 32, 8, 1),      # ragged: waves with 8 or 9 units, the closing chain runs masked units
    (32768 + 64, 7, 8, 8, 2),    # batch 2 (16 splits per kv head, 65 units); G = 1
    (131072, 64, 16, 8, 1),      # 128 units per split: four rounds; G = 2
    (90000, 1, 64, 8, 1),        # G = 8
])
def test_attn_long_splits(T, r, nh, nhk, bs, env, oracle):
    torch, ops = env
    c = synth.attn_case(6000 + T % 983 + r, bs, nh, nhk, 128, 64, 256, T, r)
    gold = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 64), gold, f"paged T={T}")
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 128, k_paged=False, i64=True, shuffle=False), gold, f"mixed T={T}")


@pytest.mark.parametrize("Tmax,nh,nhk", [(1000, 8, 2), (33000, 32, 8)])
def test_ragged_batch_device_lengths(Tmax, nh, nhk, env, oracle):
    """Requests of different lengths in one batch: with device-resident lengths every batch item carries its own
    (T, r, start); the host descriptor only gives the bound.  (The reference assumes one length for the batch.)"""
    torch, ops = env
    bs, ps, cap = 3, 64, 128
    Ts, rs_, starts = [Tmax, 37, (Tmax * 2) // 3], [5, 128, 0], [0, 100, 7]
    c = synth.attn_case(8100 + Tmax % 97, bs, nh, nhk, 128, 64, 256, Tmax, cap)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    kr, vr = t["k_res"].clone(), t["v_res"].clone()
    for b in range(bs):      # ring start per batch item
        kr[b] = torch.roll(t["k_res"][b], starts[b], dims=1)
        vr[b] = torch.roll(t["v_res"][b], starts[b], dims=1)
    lengths = torch.tensor([[Ts[b], rs_[b], starts[b], 0] for b in range(bs)], dtype=torch.int32, device="cuda")
    out = ops.pq_decode_attn(t["q"], torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda(), kp, vp, kr, vr, 0,
                             M=64, C=256, n_tokens=Tmax, k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps,
                             dev_lengths=lengths)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for b in range(bs):
        cb = {k: (v[b:b + 1] if isinstance(v, np.ndarray) and v.shape[0] == bs else v) for k, v in c.items()}
        cb["k_codes"], cb["v_codes"], cb["r"] = cb["k_codes"][:, :, :Ts[b]], cb["v_codes"][:, :, :Ts[b]], rs_[b]
        _check(o[b:b + 1], oracle.decode_attn(**cb), f"batch item {b} (T={Ts[b]}, r={rs_[b]})")


# ---- every BASELINE.json config at its own shape (configs_untested) -----------------------------------
def test_baseline_config1_llama2_7b_4k(env, oracle):
    """configs[1]: Llama-2-7B-hf, 4K context, PQ M=64 nbits=8, batch 1: nh = nh_k = 32 (MHA, G = 1), T = 4096.
    Through the reference's three layouts: 10-arg row-major, fully paged, 13-arg row-major K + paged V."""
    torch, ops = env
    c = synth.attn_case(2201, 1, 32, 32, 128, 64, 256, 4096, 17)
    gold = oracle.decode_attn(**c)
    _check(_run_rowmajor(torch, ops, c, 64, 256), gold, "cfg1 rowmajor")
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 64), gold, "cfg1 paged")
    _check(_run_paged(torch, ops, oracle, c, 64, 256, 64, k_paged=False, i64=True), gold, "cfg1 mixed")


def test_baseline_config3_per_gpu_shape(env, oracle):
    """configs[3] per GPU: Llama-3.1-8B, 32K context, 16 requests over 8 GPUs = batch 2 per GPU: bs = 2, nh = 32,
    nh_k = 8, T = 32768, M = 64, PagedPQCache layout (paged K and V), fused append, device-resident lengths; two
    consecutive decode steps (the second reads the row the first one parked)."""
    torch, ops = env
    bs, nh, nhk, ps, cap, T, r0, start = 2, 32, 8, 64, 128, 32768, 100, 40
    c = synth.attn_case(2203, bs, nh, nhk, 128, 64, 256, T, cap)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    perm = np.random.RandomState(6).permutation(vpool.shape[0])
    vpool, kpool, ids = vpool[perm], kpool[perm], np.argsort(perm)[ids]
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    kc, vc = torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda()
    kr = torch.roll(t["k_res"], start, dims=2).contiguous()
    vr = torch.roll(t["v_res"], start, dims=2).contiguous()
    lengths = torch.tensor([[T, r0, start, 0]] * bs, dtype=torch.int32, device="cuda")
    rs = np.random.RandomState(10)
    k_hist, v_hist = c["k_res"].copy(), c["v_res"].copy()
    r = r0
    for step in range(2):
        k_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        v_new = rs.standard_normal((bs, nhk, 1, 128)).astype(np.float16)
        out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, kr, vr, 0, M=64, C=256, resid_start=start, dev_lengths=lengths,
                                 k_new=torch.from_numpy(k_new).cuda(), v_new=torch.from_numpy(v_new).cuda(),
                                 k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps, n_tokens=T)
        torch.cuda.synchronize()
        k_hist[:, :, r], v_hist[:, :, r] = k_new[:, :, 0], v_new[:, :, 0]
        r += 1
        _check(out.cpu().numpy(), oracle.decode_attn(**dict(c, k_res=k_hist, v_res=v_hist, r=r)), f"cfg3 step {step}")
    assert lengths.cpu().numpy()[:, 1].tolist() == [r] * bs


def test_baseline_config4_128k_m32(env, oracle):
    """configs[4]: Llama-3.1-8B, 128K context, PQ M=32 nbits=8, batch 1: nh = 32, nh_k = 8, T = 131072, paged."""
    torch, ops = env
    c = synth.attn_case(2204, 1, 32, 8, 128, 32, 256, 131072, 77)
    gold = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, 32, 256, 64), gold, "cfg4 paged")
    _check(_run_paged(torch, ops, oracle, c, 32, 256, 64, k_paged=False, i64=True, shuffle=False), gold, "cfg4 mixed")


def test_kernel_registry_get_kernel(env, oracle):
    """KernelRegistry.get_kernel(l) (pq_utils.py:43-96): resolves the binding by the Ns that l2Ns(l) picks, owns the
    (bs=1, nh, Ns+1, ...) partial buffers, returns the 8-arg closure DynamicPQCache.decoding calls (:315-325)."""
    torch, ops = env
    from million_amd.pq_cache import KernelRegistry, l2Ns
    nh, d, M, C = 8, 128, 64, 256
    reg = KernelRegistry(M=M, d=d, nbits=8, nh=nh, scalar_t=torch.float16, device="cuda")
    for l in (64, 65, 129, 257, 2049):
        r = 17
        c = synth.attn_case(300 + l, 1, nh, nh, d, M, C, l - r, r)
        t = _dev(torch, c)
        k = reg.get_kernel(l)
        Ns = l2Ns(l)
        assert k is reg.get_kernel(l) and set(reg.kernels) >= {Ns}
        assert reg.partial_out_buffers[Ns].shape == (1, nh, Ns + 1, d) and reg.partial_lse_buffers[Ns].shape == (1, nh, Ns + 1)
        out = k(t["q"], t["k_codes"], t["v_codes"], t["k_cents"], t["v_cents"], t["k_res"], t["v_res"], r)
        _check(out.cpu().numpy(), oracle.decode_attn(**c), f"registry l={l} Ns={Ns}")
    assert sorted(reg.kernels) == [1, 2, 4, 16, 32]
    with pytest.raises(NotImplementedError):
        KernelRegistry(M=M, d=d, nbits=12, nh=nh).get_kernel(100)


def test_dynamic_cache_decoding_through_registry(env, oracle):
    """DynamicPQCache.decoding(fused=False): the reference's call sequence (window copy, then the kernel that
    KernelRegistry.get_kernel(seen_tokens) resolved from `bindings`) gives the same output as the fused launch."""
    torch, ops = env
    from million_amd.pq_cache import DynamicPQCache
    bs, nh, nhk, d, M, C = 1, 8, 2, 128, 64, 256
    rs = np.random.RandomState(13)
    ck = torch.from_numpy(rs.standard_normal((M, C, 2)).astype(np.float16)).cuda()
    n_prompt, n_dec = 70, 140
    K = torch.from_numpy(rs.standard_normal((bs, nhk, n_prompt + n_dec, d)).astype(np.float16)).cuda()
    V = torch.from_numpy(rs.standard_normal((bs, nhk, n_prompt + n_dec, d)).astype(np.float16)).cuda()
    Q = torch.from_numpy(rs.standard_normal((n_dec, bs, nh, 1, d)).astype(np.float16)).cuda()
    outs = {}
    for fused in (True, False):
        cache = DynamicPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, max_tokens=512)
        cache.set_cent(ck, ck)
        cache.prefill(Q[0].repeat(1, 1, n_prompt, 1), K[:, :, :n_prompt].contiguous(), V[:, :, :n_prompt].contiguous(), 0)
        o = [cache.decoding(Q[i], K[:, :, n_prompt + i:n_prompt + i + 1].contiguous(),
                            V[:, :, n_prompt + i:n_prompt + i + 1].contiguous(), 0, fused=fused) for i in range(n_dec)]
        outs[fused] = torch.stack(o).float().cpu().numpy()
        if not fused:
            assert sorted(cache.registery.kernels) == [2, 4]      # l2Ns(71..128) = 2, l2Ns(129..210) = 4
    assert np.abs(outs[True] - outs[False]).max() < 2e-3


# ---- nbits 9..16: uint16 codes (reference nbits2dtype pq_utils.py:542-552, update :166-220) --------------
@pytest.mark.parametrize("case", synth.GOLDEN_ENCODE_U16, ids=[c[0] for c in synth.GOLDEN_ENCODE_U16])
def test_encode_decode_u16_bit_exact(case, env, oracle, golden_dir):
    torch, ops = env
    name, seed, bs, nhk, n, d, M, C = case
    c = synth.encode_case(seed, bs, nhk, n, d, M, C)
    Xd, cd = torch.from_numpy(c["X"]).cuda(), torch.from_numpy(c["cents"]).cuda()
    codes = ops.pq_encode(Xd, cd)
    assert codes.dtype == torch.uint16 and codes.shape == (bs, nhk, n, M)
    gold = oracle.pq_encode(c["X"], c["cents"])
    np.testing.assert_array_equal(codes.cpu().numpy(), gold)
    # the reference's own sa_encode_4d(target_dtype=uint16) output: near-tie flips of the cdist form only
    fix = np.load(golden_dir / f"encode_{name}.npz")
    assert (codes.cpu().numpy() != fix["codes"]).sum() <= 1
    dec = ops.pq_decode(torch.from_numpy(fix["codes"]).cuda(), cd)
    np.testing.assert_array_equal(dec.cpu().numpy().view(np.uint16), fix["decoded"].view(np.uint16))
    with pytest.raises(RuntimeError):
        ops.pq_decode(torch.zeros(1, 1, 1, M, dtype=torch.uint8, device="cuda"), cd)      # wrong code width for C > 256


def test_encode_u16_layouts_and_big(env, oracle):
    """uint16 codes in the three destination layouts, ragged sizes, d_m = 2 and 4, C = 512 / 4096."""
    torch, ops = env
    from million_amd import _lib as L
    for (M, C, n, seed) in ((64, 512, 300, 1), (32, 4096, 70, 2)):
        rs = np.random.RandomState(seed)
        d = 128
        cents = rs.standard_normal((M, C, d // M)).astype(np.float16)
        cents[:, C - 1] = cents[:, 5]                  # exact duplicates: lower index must win
        X = rs.standard_normal((2, 2, n, d)).astype(np.float16)
        X[0, 0, :3] = cents[:, 5].reshape(-1)
        gold = oracle.pq_encode(X, cents)
        assert gold.dtype == np.uint16 and (gold[0, 0, :3] == 5).all() and gold.max() > 255
        Xd, cd = torch.from_numpy(X).cuda(), torch.from_numpy(cents).cuda()
        np.testing.assert_array_equal(ops.pq_encode(Xd, cd).cpu().numpy(), gold)
        ps, t0 = 64, 64
        n_pages = (t0 + n + ps - 1) // ps
        ids = torch.arange(2 * 2 * n_pages, dtype=torch.int32).reshape(2, 2, n_pages).flip(2).contiguous().cuda()
        kpool = torch.zeros(2 * 2 * n_pages, ps, M, dtype=torch.uint16).cuda()
        vpool = torch.zeros(2 * 2 * n_pages, M, ps, dtype=torch.uint16).cuda()
        ops.pq_encode_into(Xd, cd, kpool, layout=L.MILLION_CODES_KPAGES, token_start=t0, page_ids=ids, page_size=ps)
        ops.pq_encode_into(Xd, cd, vpool, layout=L.MILLION_CODES_VPAGES, token_start=t0, page_ids=ids, page_size=ps)
        kp, vp, idn = kpool.cpu().numpy(), vpool.cpu().numpy(), ids.cpu().numpy()
        for b in range(2):
            for h in range(2):
                toks = np.arange(n) + t0
                pid = idn[b, h, toks // ps]
                np.testing.assert_array_equal(kp[pid, toks % ps], gold[b, h])
                np.testing.assert_array_equal(vp[pid, :, toks % ps], gold[b, h])
        # decode of wide codes == the oracle's gather
        dec = ops.pq_decode(torch.from_numpy(gold).cuda(), cd)
        np.testing.assert_array_equal(dec.cpu().numpy(), oracle.pq_decode_numpy(gold, cents).astype(np.float16))


def test_dynamic_cache_nbits10_update(env, oracle):
    """DynamicPQCache(nbits=10): uint16 code store; update / prefill(distort_recent) (pq_utils.py:166-260) against the
    oracle; the fused decode kernels refuse (uint8 only, as the reference's KernelRegistry, pq_utils.py:50-52)."""
    torch, ops = env
    from million_amd.pq_cache import DynamicPQCache, nbits2dtype
    bs, nh, nhk, M, d, nbits = 1, 8, 2, 64, 128, 10
    rs = np.random.RandomState(41)
    ck = rs.standard_normal((M, 2 ** nbits, d // M)).astype(np.float16)
    cv = rs.standard_normal((M, 2 ** nbits, d // M)).astype(np.float16)
    cache = DynamicPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, nbits=nbits, max_tokens=256)
    assert cache.dtype == nbits2dtype(nbits) == torch.uint16
    cache.set_cent(torch.from_numpy(ck).cuda(), torch.from_numpy(cv).cuda())
    k1, v1 = rs.standard_normal((bs, nhk, 50, d)).astype(np.float16), rs.standard_normal((bs, nhk, 50, d)).astype(np.float16)
    k2, v2 = rs.standard_normal((bs, nhk, 2, d)).astype(np.float16), rs.standard_normal((bs, nhk, 2, d)).astype(np.float16)
    dq = lambda x, c: oracle.pq_decode_numpy(oracle.pq_encode(x, c), c)
    cache.update(torch.from_numpy(k1).cuda(), torch.from_numpy(v1).cuda(), 0)
    K, V = cache.update(torch.from_numpy(k2).cuda(), torch.from_numpy(v2).cuda(), 0)
    np.testing.assert_array_equal(K.cpu().numpy(), np.concatenate([dq(k1, ck), k2], axis=2))
    np.testing.assert_array_equal(V.cpu().numpy(), np.concatenate([dq(v1, cv), v2], axis=2))
    assert cache.key_cache[0].dtype == torch.uint16 and cache.key_cache[0].shape == (bs, nhk, 52, M)
    np.testing.assert_array_equal(cache.key_cache[0].cpu().numpy(), oracle.pq_encode(np.concatenate([k1, k2], axis=2), ck))
    K, V = cache.update(torch.from_numpy(k2).cuda(), torch.from_numpy(v2).cuda(), 0, distort_recent=True)
    np.testing.assert_array_equal(V.cpu().numpy(), np.concatenate([dq(v1, cv), dq(v2, cv), dq(v2, cv)], axis=2))
    with pytest.raises(NotImplementedError):
        cache.decoding(torch.zeros(bs, nh, 1, d, dtype=torch.float16, device="cuda"), torch.from_numpy(k2[:, :, :1]).cuda(),
                       torch.from_numpy(v2[:, :, :1]).cuda(), 0)
    with pytest.raises(ValueError):
        DynamicPQCache(bs=1, nh=8, num_key_value_heads=2, M=M, layer_num=1, d=d, nbits=10, dtype=torch.uint8)


def test_flush_one_launch_equals_separate_encodes(env, oracle):
    """million_pq_flush (K encode + V encode + device-length advance in one launch) == two million_pq_encode calls +
    million_lengths_advance, for host and device lengths, ring start in the middle of the window, batch 2."""
    torch, ops = env
    from million_amd import _lib as L
    bs, nhk, d, M, C, ps, cap = 2, 3, 128, 64, 256, 64, 128
    rs = np.random.RandomState(51)
    ck = torch.from_numpy(rs.standard_normal((M, C, 2)).astype(np.float16)).cuda()
    cv = torch.from_numpy(rs.standard_normal((M, C, 2)).astype(np.float16)).cuda()
    kw = torch.from_numpy(rs.standard_normal((bs, nhk, cap, d)).astype(np.float16)).cuda()
    vw = torch.from_numpy(rs.standard_normal((bs, nhk, cap, d)).astype(np.float16)).cuda()
    n_pages = 6
    ids = torch.randperm(bs * nhk * n_pages).to(torch.int32).reshape(bs, nhk, n_pages).cuda()
    T0, start = 128, 70
    for use_dl in (False, True):
        pools = []
        for fused in (False, True):
            kpool = torch.zeros(bs * nhk * n_pages, ps, M, dtype=torch.uint8, device="cuda")
            vpool = torch.zeros(bs * nhk * n_pages, M, ps, dtype=torch.uint8, device="cuda")
            dl = torch.tensor([[T0, cap, start, 0]] * bs, dtype=torch.int32, device="cuda") if use_dl else None
            if fused:
                ops.pq_flush(kw, vw, ck, cv, kpool, vpool, ids, n=ps, page_size=ps, token_start=T0, x_row_start=start, dev_lengths=dl)
            else:
                kwargs = dict(token_start=T0, n=ps, page_ids=ids, page_size=ps, x_row_start=start, x_row_mod=cap, dev_lengths=dl)
                ops.pq_encode_into(kw, ck, kpool, layout=L.MILLION_CODES_KPAGES, **kwargs)
                ops.pq_encode_into(vw, cv, vpool, layout=L.MILLION_CODES_VPAGES, **kwargs)
                if use_dl:
                    ops.lengths_advance(dl, ps, cap)
            torch.cuda.synchronize()
            pools.append((kpool.cpu().numpy(), vpool.cpu().numpy(), None if dl is None else dl.cpu().numpy()))
        np.testing.assert_array_equal(pools[0][0], pools[1][0])
        np.testing.assert_array_equal(pools[0][1], pools[1][1])
        if use_dl:
            np.testing.assert_array_equal(pools[0][2], pools[1][2])
            assert pools[1][2].tolist() == [[T0 + ps, cap - ps, (start + ps) % cap, 0]] * bs
    # and against the oracle: page 2 of (b, hk) holds the codes of ring rows start .. start + 63
    rows = (np.arange(ps) + start) % cap
    gold = oracle.pq_encode(kw.cpu().numpy()[:, :, rows], ck.cpu().numpy())
    idn = ids.cpu().numpy()
    for b in range(bs):
        for h in range(nhk):
            np.testing.assert_array_equal(pools[1][0][idn[b, h, T0 // ps]], gold[b, h])


@pytest.mark.parametrize("bs,nhk,d,M,C,ps,cap,n", [(1, 8, 128, 64, 256, 64, 128, 64), (1, 32, 128, 64, 256, 64, 128, 64),
                                                   (2, 3, 128, 32, 256, 64, 256, 128), (1, 5, 64, 16, 128, 32, 128, 32),
                                                   (3, 1, 128, 16, 256, 64, 128, 64), (1, 20, 64, 64, 256, 128, 512, 128)])
def test_flush_kernel_head_groups_and_shapes(bs, nhk, d, M, C, ps, cap, n, env, oracle):
    """The one-launch flush deals kv heads to 16-wave workgroups (hp heads x 16 / hp parts of the centroid range): every
    grouping the launcher can pick (8 heads -> 2 groups of 4, 32 heads -> 2 passes of 16, 3 / 5 / 20 heads -> idle waves
    and a short last group, 1 head -> 16 parts), d_m in {1, 2, 4, 8}, C = 128, two token blocks; codes vs the oracle,
    lengths advanced on the device."""
    torch, ops = env
    rs = np.random.RandomState(600 + nhk + M)
    ck = rs.standard_normal((M, C, d // M)).astype(np.float16)
    cv = rs.standard_normal((M, C, d // M)).astype(np.float16)
    kw = rs.standard_normal((bs, nhk, cap, d)).astype(np.float16)
    vw = rs.standard_normal((bs, nhk, cap, d)).astype(np.float16)
    n_pages, T0, start = 7, 2 * ps, cap - 9
    ids = torch.randperm(bs * nhk * n_pages).to(torch.int32).reshape(bs, nhk, n_pages).cuda()
    kpool = torch.zeros(bs * nhk * n_pages, ps, M, dtype=torch.uint8, device="cuda")
    vpool = torch.zeros(bs * nhk * n_pages, M, ps, dtype=torch.uint8, device="cuda")
    dl = torch.tensor([[T0, cap, start, 0]] * bs, dtype=torch.int32, device="cuda")
    ops.pq_flush(torch.from_numpy(kw).cuda(), torch.from_numpy(vw).cuda(), torch.from_numpy(ck).cuda(),
                 torch.from_numpy(cv).cuda(), kpool, vpool, ids, n=n, page_size=ps, dev_lengths=dl)
    torch.cuda.synchronize()
    assert dl.cpu().tolist() == [[T0 + n, cap - n, (start + n) % cap, 0]] * bs
    rows = (np.arange(n) + start) % cap
    gk, gv = oracle.pq_encode(kw[:, :, rows], ck), oracle.pq_encode(vw[:, :, rows], cv)
    idn = ids.cpu().numpy()
    np.testing.assert_array_equal(oracle.pool_to_k_rowmajor(kpool.cpu().numpy(), idn, T0 + n)[:, :, T0:], gk)
    np.testing.assert_array_equal(oracle.pool_to_v_rowmajor(vpool.cpu().numpy(), idn, T0 + n)[:, :, T0:], gv)
    assert not oracle.pool_to_k_rowmajor(kpool.cpu().numpy(), idn, T0)[:, :, :T0].any()      # nothing written elsewhere


def test_flush_device_lengths_are_clamped_not_trusted(env, oracle):
    """Corrupt device lengths handed to the flush / encode kernels (destination token past the page table or negative,
    ring start outside the window) must drop stores or fall back to start 0 - never write outside the pools - and the
    advanced lengths stay inside their ranges."""
    torch, ops = env
    from million_amd import _lib as L
    bs, nhk, d, M, C, ps, cap, n_pages = 4, 2, 128, 64, 256, 64, 128, 3
    rs = np.random.RandomState(77)
    ck = torch.from_numpy(rs.standard_normal((M, C, 2)).astype(np.float16)).cuda()
    kw = rs.standard_normal((bs, nhk, cap, d)).astype(np.float16)
    kwd = torch.from_numpy(kw).cuda()
    ids = torch.arange(bs * nhk * n_pages, dtype=torch.int32).reshape(bs, nhk, n_pages).cuda()
    bad = [[n_pages * ps + 5000, cap, 0, 0], [-64, cap, 0, 0], [ps, cap, cap + 40, 0], [ps, 10, -3, 0]]
    guard = 4      # guard pages around the pools: must stay zero
    for which in ("flush", "encode_small", "encode_bulk"):
        kpool = torch.zeros((bs * nhk * n_pages + 2 * guard), ps, M, dtype=torch.uint8, device="cuda")
        vpool = torch.zeros((bs * nhk * n_pages + 2 * guard), M, ps, dtype=torch.uint8, device="cuda")
        kin, vin = kpool[guard:-guard], vpool[guard:-guard]
        dl = torch.tensor(bad, dtype=torch.int32, device="cuda")
        if which == "flush":
            ops.pq_flush(kwd, kwd, ck, ck, kin, vin, ids, n=ps, page_size=ps, dev_lengths=dl)
        else:
            nn = ps if which == "encode_small" else cap
            src = kwd if which == "encode_small" else kwd.repeat(1, 8, 1, 1)[:, :nhk * 8]
            idw = ids if which == "encode_small" else ids.repeat(1, 8, 1)
            if which == "encode_bulk":      # 16 kv heads x 128 rows: enough waves for the bulk kernel
                kin = torch.zeros(bs * nhk * 8 * n_pages, ps, M, dtype=torch.uint8, device="cuda")
                vin = torch.zeros(bs * nhk * 8 * n_pages, M, ps, dtype=torch.uint8, device="cuda")
                idw = torch.arange(bs * nhk * 8 * n_pages, dtype=torch.int32).reshape(bs, nhk * 8, n_pages).cuda()
            kw_ = dict(n=nn, page_ids=idw, page_size=ps, x_row_start=0, x_row_mod=cap, dev_lengths=dl)
            ops.pq_encode_into(src, ck, kin, layout=L.MILLION_CODES_KPAGES, **kw_)
            ops.pq_encode_into(src, ck, vin, layout=L.MILLION_CODES_VPAGES, **kw_)
        torch.cuda.synchronize()
        kp = kpool.cpu().numpy()
        assert not kp[:guard].any() and not kp[-guard:].any() and not vpool.cpu().numpy()[:guard].any()
        kn = kin.cpu().numpy()
        idn = (ids if which != "encode_bulk" else idw).cpu().numpy()
        nn = ps if which != "encode_bulk" else cap
        srcn = kw if which != "encode_bulk" else np.tile(kw, (1, 8, 1, 1))
        # items 0, 1: destination outside the table / negative -> nothing stored
        for b in (0, 1):
            assert not kn[idn[b].ravel()].any(), (which, b)
        # items 2, 3: ring start outside the window -> rows from start 0, stored at token ps
        gold = oracle.pq_encode(srcn[:, :, :nn], ck.cpu().numpy())
        for b in (2, 3):
            for h in range(idn.shape[1]):
                if which == "encode_bulk" and nn > ps:      # tokens ps .. ps + 127: pages 1 and 2
                    got = np.concatenate([kn[idn[b, h, 1]], kn[idn[b, h, 2]]])
                else:
                    got = kn[idn[b, h, 1]]
                np.testing.assert_array_equal(got, gold[b, h], err_msg=f"{which} item {b} head {h}")
        if which == "flush":
            out = dl.cpu().numpy()
            assert (out[:, 3] == 0).all()
            assert (out[:, 0] >= 0).all() and (out[:, 0] <= n_pages * ps).all()
            assert (out[:, 1] >= 0).all() and (out[:, 2] >= 0).all() and (out[:, 2] < cap).all()


def test_rowmajor_v_shadow_reuse_and_invalidation(env, oracle):
    """The 10-argument layout (row-major V) on the fast shapes: the transposed pages of a V code tensor are made once and
    reused while the same tensor object is passed (the reference passes value_cache[layer] unchanged between flushes);
    an in-place write or a new tensor gets fresh pages."""
    torch, ops = env
    c = synth.attn_case(611, 1, 32, 8, 128, 64, 256, 3000, 40)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
    call = lambda vc: ops.pq_decode_attn(t["q"], t["k_codes"], vc, kp, vp, t["k_res"], t["v_res"], c["r"], M=64, C=256)
    ops._vshadow.clear()
    o1 = call(t["v_codes"])
    pages = ops._vshadow[id(t["v_codes"])][2]
    o2 = call(t["v_codes"])
    assert ops._vshadow[id(t["v_codes"])][2] is pages and len(ops._vshadow) == 1      # reused, not re-made
    torch.cuda.synchronize()
    gold = oracle.decode_attn(**c)
    _check(o1.cpu().numpy(), gold, "first call")
    _check(o2.cpu().numpy(), gold, "second call (shadow hit)")
    # in-place change of the codes (version counter moves): pages are rebuilt
    t["v_codes"][:, :, 100:200] = 7
    c2 = dict(c, v_codes=t["v_codes"].cpu().numpy())
    o3 = call(t["v_codes"])
    assert ops._vshadow[id(t["v_codes"])][2] is not pages
    _check(o3.cpu().numpy(), oracle.decode_attn(**c2), "after in-place write")
    # a different tensor object with equal content: its own entry
    v2 = t["v_codes"].clone()
    _check(call(v2).cpu().numpy(), oracle.decode_attn(**c2), "clone")
    assert len(ops._vshadow) == 2
    del v2
    call(t["v_codes"])
    assert len(ops._vshadow) == 1 + sum(1 for v in ops._vshadow.values() if v[0]() is None)
    # fresh VIEWS of one store on every call (what DynamicPQCache.decoding(fused=False) and the reference's PagedPQCache
    # pass: store[:, :, :T]) hit the pages made for the previous view: the shadow is keyed on the base tensor
    ops._vshadow.clear()
    store = torch.zeros(1, 8, 4096, 64, dtype=torch.uint8, device="cuda")
    store[:, :, :3000] = t["v_codes"]
    o4 = call(store[:, :, :3000])
    pages4 = ops._vshadow[id(store)][2]
    o5 = call(store[:, :, :3000])
    assert ops._vshadow[id(store)][2] is pages4 and len(ops._vshadow) == 1
    _check(o4.cpu().numpy(), oracle.decode_attn(**c2), "view, first call")
    _check(o5.cpu().numpy(), oracle.decode_attn(**c2), "view, second call (hit through the base)")
    # a kernel of this library writing into a view of the store (no version bump) drops the store's shadow
    from million_amd import _lib as L
    X = torch.randn(1, 8, 64, 128, device="cuda").half()
    ops.pq_encode_into(X, t["v_cents"], store[:, :, :3064], token_start=3000, n=64)
    assert id(store) not in ops._vshadow


def test_harness_pq_step_attention_against_oracle(env, oracle):
    """million_amd/harness.py, PQ backend: inside a real decode step of the Llama-shaped model (q/k/v projections and
    RoPE in torch, then PagedPQCache.decoding_with_pages as attn_forward_custom_kernel calls it,
    modeling_llama.py:455-554) the attention output of every layer equals the oracle's, evaluated on the cache's own
    state (codes read back through the page table, window rows, the new K/V row); the prompt is really prefilled (bulk
    encode into pages), the decode steps cross a window flush; eager launches and hipGraph replay."""
    torch, ops = env
    from million_amd import harness as H
    shape = H.LlamaShape(hidden=256, n_layers=2, nh=16, nh_k=4, d=128, inter=512, vocab=500)
    dev = torch.device("cuda", 0)
    model = H.LlamaShapeDecoder(shape, dev, seed=3)
    n_prompt, bs, nl = 700, 1, shape.n_layers
    for mode in ("eager", "graph"):
        be = H.PQBackend(shape, bs, n_prompt, 256, dev, synthetic_fill=False)
        cache = be.cache
        prompt = torch.randint(0, shape.vocab, (bs, n_prompt), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        tokens = model.prefill(prompt, be).clone()
        assert cache._T[0] == n_prompt and cache.residualed_tokens[0] == 0
        pos = torch.full((bs,), n_prompt, dtype=torch.long, device=dev)
        captured, orig = [], be.attend

        def spy(layer, q, k, v):
            out = orig(layer, q, k, v)
            captured.append((layer, q.clone(), k.clone(), v.clone(), out))      # under capture the clones are graph nodes
            return out
        be.attend = spy
        caps_of = None
        if mode == "graph":      # ctor: one eager step, then one capture per kind of step; a replay refreshes its clones
            gd = H.GraphedPQDecoder(model, be, tokens, pos)
            names = [n for n, _ in cache.capture_states() if n in gd.graphs]
            assert names == ["plain", "pre", "commit", "flush"] and len(captured) == (1 + len(names)) * nl
            caps_of = {n: captured[(i + 1) * nl:(i + 2) * nl] for i, n in enumerate(names)}
        ck, cv = cache.key_cent.cpu().numpy(), cache.value_cent.cpu().numpy()
        for step in range(140):                                 # the window fills after 128 steps: one flush inside
            flush, kind = cache.next_step_flushes(), cache.next_step_kind()
            T, r, rs = cache._T[0], cache.residualed_tokens[0], cache._rstart[0]      # what the launches of this step read
            if flush:
                T, r, rs = T + 64, r - 64, (rs + 64) % 128
            # the oldest page is encoded ahead at step 72 (PagedPQCache.begin_step), step 128 commits it
            assert kind == {72: "pre", 128: "commit", 136: "pre"}.get(step, "plain")
            check = step in (0, 1, 72, 73, 127, 128, 129, 139)
            if check:
                win = [(cache.key_residual_cache[l].cpu().numpy().copy(), cache.value_residual_cache[l].cpu().numpy().copy())
                       for l in range(nl)]
            captured.clear()
            if mode == "graph":
                gd.step()
            else:
                tokens.copy_(model.step(tokens, pos, be))
                pos.add_(1)
            torch.cuda.synchronize()
            if not check:
                continue
            assert flush == (step == 128)
            caps = caps_of[kind] if mode == "graph" else list(captured)
            assert len(caps) == nl
            kpool, vpool = cache.key_page_pool.cpu().numpy(), cache.value_page_pool.cpu().numpy()
            for (layer, q, k, v, out) in caps:
                ids = cache.page_ids[layer].cpu().numpy()
                n_pages = (T + 63) // 64
                kc = np.concatenate([kpool[ids[:, :, p]] for p in range(n_pages)], axis=2)[:, :, :T]          # (bs, nh_k, T, M)
                vc = np.concatenate([vpool[ids[:, :, p]].transpose(0, 1, 3, 2) for p in range(n_pages)], axis=2)[:, :, :T]
                kw, vw = win[layer]
                if flush:      # the 64 oldest window rows became codes: they must be the oracle's encode of those rows
                    old = (np.arange(64) + (rs - 64) % 128) % 128
                    np.testing.assert_array_equal(kc[:, :, T - 64:], oracle.pq_encode(kw[:, :, old], ck))
                    np.testing.assert_array_equal(vc[:, :, T - 64:], oracle.pq_encode(vw[:, :, old], cv))
                rows = (np.arange(r) + rs) % 128
                kres = np.zeros((bs, shape.nh_k, 128, 128), np.float16)
                vres = np.zeros((bs, shape.nh_k, 128, 128), np.float16)
                kres[:, :, :r], vres[:, :, :r] = kw[:, :, rows], vw[:, :, rows]
                kres[:, :, r], vres[:, :, r] = k.cpu().numpy()[:, :, 0], v.cpu().numpy()[:, :, 0]
                gold = oracle.decode_attn(q.cpu().numpy(), kc, vc, ck, cv, kres, vres, r + 1)
                _check(out.float().cpu().numpy(), gold, f"{mode} step {step} layer {layer}")
        assert cache._T[0] == n_prompt + 64 and cache.residualed_tokens[0] == 140 - 64


@pytest.mark.parametrize("T,r,start,M", [(0, 256, 0, 64), (40, 200, 100, 64), (5000, 256, 255, 64), (33000, 129, 7, 32)])
def test_attn_window_of_256_rows_on_mfma(T, r, start, M, env, oracle):
    """extended_residual_size = 256 (the reference's flash_decoding_paged_v_*_Lt256 names; PagedPQCache's advertised
    extended_residual in {64, 128, 256}): windows longer than 128 rows stay on the MFMA kernels - the launch uses at
    least two splits, each takes every nsplit-th row into its 128 tile slots."""
    torch, ops = env
    from million_amd import _lib
    nh, nhk, ps, cap = 32, 8, 64, 256
    c = synth.attn_case(9300 + T % 71 + r, 1, nh, nhk, 128, M, 256, T, r, Lt=cap)
    gold = oracle.decode_attn(**c)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda() if T else torch.zeros(1, nhk, 1, dtype=torch.int32, device="cuda")
    kr = torch.roll(t["k_res"], start, dims=2).contiguous()
    vr = torch.roll(t["v_res"], start, dims=2).contiguous()
    out = ops.pq_decode_attn(t["q"], torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda(), kp, vp, kr, vr, r,
                             M=M, C=256, n_tokens=T, resid_start=start, k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps)
    torch.cuda.synchronize()
    _check(out.cpu().numpy(), gold, f"cap 256 T={T} r={r}")
    desc = ops.make_attn_desc(t["q"], kr, nh_k=nhk, M=M, C=256, n_tokens=max(T, 1), r=r, k_paged=True, v_paged=True,
                              page_size=ps, n_pages_cap=max(ids_t.shape[2], 1))
    assert _lib.load().million_attn_kernel_kind(ctypes.byref(desc)) == 1
    if T and M == 64:      # the 13-argument name with Lt256 (paged_pq_utils.py:547), row-major K + paged V, int64 ids
        import bindings
        fn = getattr(bindings, "flash_decoding_paged_v_f16u8_Ns32Lt256d128M64C256")
        po = torch.empty(1, nh, 33, 128, dtype=torch.float16, device="cuda")
        pl = torch.empty(1, nh, 33, dtype=torch.float16, device="cuda")
        out2 = fn(t["q"], t["k_codes"], t["k_cents"], t["k_res"][:, :, :r], torch.from_numpy(ids).cuda(),
                  torch.from_numpy(vpool).cuda(), t["v_cents"], t["v_res"][:, :, :r], r, ids.shape[2], ps, po, pl)
        _check(out2.cpu().numpy(), gold, "bindings Lt256")


@pytest.mark.parametrize("bs,cap,T,r", [(32, 256, 1500, 200), (20, 128, 2100, 77), (40, 256, 700, 256)])
def test_attn_more_workgroups_than_resident_slots(bs, cap, T, r, env, oracle):
    """Grids with more workgroups than the chip holds at once (one 138-KiB workgroup per CU) AND several splits per (b, kv
    head): bs * nh_k = 256 with a 256-row window (two splits forced by the window), 160 pairs x 2 splits, 320 pairs x 2.  With
    a multiple of 8 pairs every split-0 workgroup is dispatched before any split-1 workgroup: round 3's tail let a split-0
    workgroup poll for the flag of a split-1 workgroup that could not be dispatched.  The merge now belongs
    to the last-arriving workgroup alone; no poll may run out of its bound (million_debug_tail_faults)."""
    torch, ops = env
    from million_amd import _lib
    nh, nhk, ps, M = 32, 8, 64, 64
    c = synth.attn_case(9400 + bs + T % 13, bs, nh, nhk, 128, M, 256, T, r, Lt=cap)
    gold = oracle.decode_attn(**c)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    _lib.load().million_debug_tail_faults()      # clear
    t0 = time.perf_counter()
    for _ in range(3):                            # the workspace's records must be at rest after each launch
        out = ops.pq_decode_attn(t["q"], torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda(), kp, vp, t["k_res"],
                                 t["v_res"], r, M=M, C=256, n_tokens=T, k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0, "a launch stalled in its tail"
    assert _lib.load().million_debug_tail_faults() == 0
    _check(out.cpu().numpy(), gold, f"bs={bs} cap={cap}")
    desc = ops.make_attn_desc(t["q"], t["k_res"], nh_k=nhk, M=M, C=256, n_tokens=T, r=r, k_paged=True, v_paged=True,
                              page_size=ps, n_pages_cap=ids_t.shape[2])
    assert _lib.load().million_attn_kernel_kind(ctypes.byref(desc)) == 1


@pytest.mark.parametrize("bs,nh,nhk,T,r", [(1, 32, 8, 32768, 100), (2, 32, 8, 9000, 5), (1, 128, 8, 20000, 64), (1, 8, 8, 4096, 17)],
                         ids=["headline", "two-requests", "sixteen-heads", "one-head"])
def test_attn_merge_helpers_give_up(bs, nh, nhk, T, r, env, oracle):
    """The split merge of the MFMA kernels: the last-arriving workgroup of a (b, kv head) is responsible for every head; the
    workgroups that arrived just before it help (one head each) when they see every split's flag within a short bound, and
    give up otherwise through a bit in the pair's ticket word, which the last arriver's own ticket returns.
    million_set_force_generic(4) presets every give-up bit, (8) takes the helpers' patience away (they give up through their
    own atomic unless every workgroup has already taken its ticket): the last arriver must then merge the heads concerned
    itself - same output, no fault; and with the helpers on, the same."""
    torch, ops = env
    from million_amd import _lib
    c = synth.attn_case(9700 + nh + T % 31, bs, nh, nhk, 128, 64, 256, T, r)
    gold = oracle.decode_attn(**c)
    lib = _lib.load()
    lib.million_debug_tail_faults()
    for mode in (4, 8):      # 4: every give-up bit preset; 8: the helpers give up through their own atomic (no polls first)
        try:
            ops.set_force_generic(mode)
            for _ in range(3):      # the ticket word of launch n must not confuse launch n + 1
                out = _run_paged(torch, ops, oracle, c, 64, 256, 64)
                _check(out, gold, f"helpers give up (mode {mode})")
        finally:
            ops.set_force_generic(0)
    for _ in range(2):
        out = _run_paged(torch, ops, oracle, c, 64, 256, 64)
    _check(out, gold, "helpers take their heads")
    assert lib.million_debug_tail_faults() == 0


def test_attn_merge_every_workgroup_is_a_merger(env, oracle):
    """8 requests x 8 kv heads at 32K tokens: 4 splits per (b, kv head) and 4 query heads, so EVERY workgroup of the launch
    is a merger and the first arriver of a pair is helper 0 - it waits for workgroups still streaming tens of microseconds of
    codes, runs out of patience on real launches and gives up through its atomic; the primary must then find the bit in its
    own ticket.  (Round 4 shipped a build for an hour whose give-up was added by all 64 lanes of the wave - bit k + 6, which
    the primary masks off: heads never written.  No test had helpers that give up on their own.)  Repeated: the outcome depends
    on arrival order."""
    torch, ops = env
    from million_amd import _lib
    c = synth.attn_case(9911, 8, 32, 8, 128, 64, 256, 32768, 64)
    gold = oracle.decode_attn(**c)
    lib = _lib.load()
    lib.million_debug_tail_faults()
    for mode in (0, 8, 0):
        try:
            ops.set_force_generic(mode)
            for it in range(4):
                out = _run_paged(torch, ops, oracle, c, 64, 256, 64, poison_out=True)
                _check(out, gold, f"every workgroup merges (mode {mode}, launch {it})")
        finally:
            ops.set_force_generic(0)
    assert lib.million_debug_tail_faults() == 0


@pytest.mark.parametrize("M,T,r,bs", [(64, 5000, 17, 1), (32, 40000, 128, 1), (64, 33000, 64, 2), (64, 0, 40, 1), (32, 100, 1, 1)])
def test_attn_c128_on_mfma(M, T, r, bs, env, oracle):
    """C = 128 centroids per subspace (nbits 7; the reference compiles C in {128, 256}, setup.py:15) on the streaming MFMA
    kernel: codebook images of half the size, same code layout; T = 0 falls back to the generic kernel."""
    torch, ops = env
    from million_amd import _lib
    nh, nhk = 32, 8
    c = synth.attn_case(9500 + T % 61 + M, bs, nh, nhk, 128, M, 128, T, r)
    assert c["k_codes"].max(initial=0) < 128
    gold = oracle.decode_attn(**c)
    _check(_run_rowmajor(torch, ops, c, M, 128), gold, "C=128 rowmajor")
    if T:
        _check(_run_paged(torch, ops, oracle, c, M, 128, 64), gold, "C=128 paged")
        _check(_run_paged(torch, ops, oracle, c, M, 128, 128, k_paged=False, i64=True), gold, "C=128 mixed ps=128")
        t = _dev(torch, c)
        desc = ops.make_attn_desc(t["q"], t["k_res"], nh_k=nhk, M=M, C=128, n_tokens=T, r=r, k_paged=True, v_paged=True,
                                  page_size=64, n_pages_cap=(T + 63) // 64)
        assert _lib.load().million_attn_kernel_kind(ctypes.byref(desc)) == 1


@pytest.mark.parametrize("policy", [0, 2, 1], ids=["stream", "grouped", "generic"])
def test_device_lengths_are_clamped_not_trusted(policy, env, oracle):
    """Device-resident lengths outside their ranges (T above the host bound, r above the window, ring start past the
    capacity, negative values) must become a shorter context / window, never an out-of-bounds read: the kernels clamp
    T to [0, host bound], r to [0, cap] (cap - 1 with a fused append) and an out-of-range start to 0."""
    torch, ops = env
    bs, nh, nhk, ps, cap, T = 4, 8, 2, 64, 128, 1500
    c = synth.attn_case(9700, bs, nh, nhk, 128, 64, 256, T, cap)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    bad = [[T + 100000, cap + 77, 0, 0], [-5, -3, 0, 0], [700, 40, cap + 9, 0], [T, 20, -1, 0]]
    eff = [(T, cap, 0), (0, 0, 0), (700, 40, 0), (T, 20, 0)]        # what the clamps make of them
    lengths = torch.tensor(bad, dtype=torch.int32, device="cuda")
    ops.set_force_generic(policy)
    try:
        out = ops.pq_decode_attn(t["q"], torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda(), kp, vp, t["k_res"],
                                 t["v_res"], 0, M=64, C=256, n_tokens=T, k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps,
                                 dev_lengths=lengths)
        torch.cuda.synchronize()
    finally:
        ops.set_force_generic(0)
    o = out.cpu().numpy()
    for b, (Tb, rb, sb) in enumerate(eff):
        cb = {k: (v[b:b + 1] if isinstance(v, np.ndarray) and v.shape[0] == bs else v) for k, v in c.items()}
        cb["k_codes"], cb["v_codes"], cb["r"] = cb["k_codes"][:, :, :Tb], cb["v_codes"][:, :, :Tb], rb
        if Tb == 0 and rb == 0:
            assert np.array_equal(o[b], np.zeros_like(o[b]))
        else:
            _check(o[b:b + 1], oracle.decode_attn(**cb), f"batch item {b}: lengths {bad[b]} -> {eff[b]}")


# ---------------------------------------------------------------- tile kernel: the rest of the binding surface ----------
_TILE_SHAPES = [(64, 16), (64, 32), (64, 64), (128, 16)]      # (d, M): d_m = 4, 2, 1, 8 (setup.py:12-15 of the reference)


def _kind(torch, ops, q, k_res, **kw):
    from million_amd import _lib
    desc = ops.make_attn_desc(q, k_res, **kw)
    return _lib.load().million_attn_kernel_kind(ctypes.byref(desc))


@pytest.mark.parametrize("d,M", _TILE_SHAPES, ids=[f"d{d}M{M}" for d, M in _TILE_SHAPES])
@pytest.mark.parametrize("C", [256, 128])
@pytest.mark.parametrize("T,r,nh,nhk,bs", [(0, 17, 4, 2, 1), (1, 1, 8, 8, 1), (17, 128, 8, 2, 1), (300, 33, 16, 2, 2),
                                           (5000, 64, 32, 8, 1), (4097, 0, 6, 2, 1)])
def test_attn_tile_kernel_shapes(d, M, C, T, r, nh, nhk, bs, env, oracle):
    """d = 64 / M = 16 shapes of the reference's build matrix run the tile MFMA kernel (not the scalar fallback):
    row-major (10-arg layout, transposed once), fully paged at every page size, and the 13-arg mixed layout."""
    torch, ops = env
    if T == 0 and r == 0:
        pytest.skip("nothing to attend")
    c = synth.attn_case(7000 + d + M + C + T + r, bs, nh, nhk, d, M, C, T, r, Lt=128)
    gold = oracle.decode_attn(**c)
    t = _dev(torch, c)
    if T:
        # d = 128 / M = 16 with up to 4 query heads per kv head runs the streaming kernel's d_m = 8 form (round 4); d = 64 with M = 32 /
        # 16 / 64 (d_m = 2 / 4 / 1), 256 centroids and up to 4 heads per kv head the lean kernel (round 5) on pages of 64 / 128 tokens
        stream16 = d == 128 and M == 16 and nh // nhk <= 16
        lean64 = d == 64 and nh // nhk <= 16      # M = 64 (d_m = 1) runs as d_m = 2 with zero odd dims; C = 128 too;
        # 5 .. 16 heads per kv head as ceil(G / 4) virtual kv heads of 3 / 4 (the last part: what is left)
        assert _kind(torch, ops, t["q"], t["k_res"], nh_k=nhk, M=M, C=C, n_tokens=T, r=r, k_codes=t["k_codes"],
                     v_codes=t["v_codes"]) == (2 if stream16 or lean64 else 4)
        assert _kind(torch, ops, t["q"], t["k_res"], nh_k=nhk, M=M, C=C, n_tokens=T, r=r, k_paged=True, v_paged=True,
                     page_size=64, n_pages_cap=(T + 63) // 64) == (1 if stream16 or lean64 else 3)
        assert _kind(torch, ops, t["q"], t["k_res"], nh_k=nhk, M=M, C=C, n_tokens=T, r=r, k_paged=True, v_paged=True,
                     page_size=32, n_pages_cap=(T + 31) // 32) == (1 if stream16 else 3)      # 32-token pages: never the lean kernel
    _check(_run_rowmajor(torch, ops, c, M, C), gold, "tile rowmajor")
    if T:
        for ps in (32, 64, 128):
            _check(_run_paged(torch, ops, oracle, c, M, C, ps), gold, f"tile paged ps={ps}")
        _check(_run_paged(torch, ops, oracle, c, M, C, 64, k_paged=False, i64=True), gold, "tile mixed i64")


def test_attn_kernel_kind_mirrors_the_hand_back(env, oracle):
    """million_attn_kernel_kind answers what a call WOULD run: d = 128 / M = 64 / C = 128 is the streaming kernel's shape, but
    with nothing quantised yet (T = 0) launch_attn_mfma hands the call to the tile kernel (kind 3); C = 256 keeps the grouped
    MFMA kernel for that (kind 5)."""
    torch, ops = env
    c = synth.attn_case(7700, 1, 8, 2, 128, 64, 128, 0, 40, Lt=128)
    t = _dev(torch, c)
    kw = dict(nh_k=2, M=64, r=40, k_paged=True, v_paged=True, page_size=64)
    assert _kind(torch, ops, t["q"], t["k_res"], C=128, n_tokens=0, n_pages_cap=1, **kw) == 3
    assert _kind(torch, ops, t["q"], t["k_res"], C=128, n_tokens=4096, n_pages_cap=64, **kw) == 1
    assert _kind(torch, ops, t["q"], t["k_res"], C=256, n_tokens=0, n_pages_cap=1, **kw) == 5      # grouped MFMA kernel (no code units)
    _check(_run_rowmajor(torch, ops, c, 64, 128), oracle.decode_attn(**c), "C=128 T=0 (tile kernel)")


@pytest.mark.parametrize("bs,nh,nhk,T,C,r", [(16, 32, 8, 40000, 256, 100), (8, 32, 32, 32768, 256, 17), (32, 32, 8, 20000, 128, 128)],
                         ids=["16x8-pairs-40K", "8x32-pairs-32K", "32x8-pairs-20K-C128"])
def test_attn_many_pairs_long_context_stay_on_streaming_kernel(bs, nh, nhk, T, C, r, env, oracle):
    """More than 64 rounds per wave at the default split count (bs * nh_k >= 128 with T > 32768, >= 256 with T > 16384): rounds
    2-3 dropped these calls to the grouped kernel, and C = 128 to the scalar one.  They now get more
    splits - more workgroups than CUs - and stay on the streaming kernel, like the reference's one kernel for any (bs, nh, T)
    (Interface.template.cu:45,62-77)."""
    torch, ops = env
    from million_amd import _lib
    ps, M = 64, 64
    c = synth.attn_case(9900 + bs + T % 17, bs, nh, nhk, 128, M, C, T, r)
    gold = oracle.decode_attn(**c)
    t = _dev(torch, c)
    desc = ops.make_attn_desc(t["q"], t["k_res"], nh_k=nhk, M=M, C=C, n_tokens=T, r=r, k_paged=True, v_paged=True,
                              page_size=ps, n_pages_cap=(T + ps - 1) // ps)
    assert _lib.load().million_attn_kernel_kind(ctypes.byref(desc)) == 1      # streaming, not grouped (5) / scalar (0)
    _lib.load().million_debug_tail_faults()
    _check(_run_paged(torch, ops, oracle, c, M, C, ps), gold, f"bs={bs} nh_k={nhk} T={T} C={C}")
    assert _lib.load().million_debug_tail_faults() == 0


@pytest.mark.parametrize("d,M", [(64, 32), (128, 16), (64, 64)], ids=["d64M32", "d128M16", "d64M64"])
@pytest.mark.parametrize("use_dl", [False, True], ids=["host-lengths", "device-lengths"])
def test_attn_tile_kernel_ring_append_ragged(d, M, use_dl, env, oracle):
    """Tile kernel: residual ring (start > 0, wrap), fused append over three steps, per-request device lengths."""
    torch, ops = env
    bs, nh, nhk, T, r0, ps, C = 2, 8, 2, 700, 37, 64, 256
    c = synth.attn_case(7100 + d + M, bs, nh, nhk, d, M, C, T, r0, Lt=128)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    rs = np.random.RandomState(2)
    start = 100
    kr = torch.roll(t["k_res"], start, dims=2).contiguous()
    vr = torch.roll(t["v_res"], start, dims=2).contiguous()
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    kc, vc = torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda()
    kw = dict(k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps, n_tokens=T)
    # request 1 is shorter than request 0 when the lengths live on the device
    T1 = 333 if use_dl else T
    lengths = torch.tensor([[T, r0, start, 0], [T1, r0, start, 0]], dtype=torch.int32, device="cuda") if use_dl else None
    k_hist, v_hist = c["k_res"].copy(), c["v_res"].copy()
    r = r0
    for step in range(3):
        k_new = rs.standard_normal((bs, nhk, 1, d)).astype(np.float16)
        v_new = rs.standard_normal((bs, nhk, 1, d)).astype(np.float16)
        out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, kr, vr, 0 if use_dl else r, M=M, C=C, resid_start=start,
                                 dev_lengths=lengths, k_new=torch.from_numpy(k_new).cuda(),
                                 v_new=torch.from_numpy(v_new).cuda(), **kw)
        torch.cuda.synchronize()
        k_hist[:, :, r], v_hist[:, :, r] = k_new[:, :, 0], v_new[:, :, 0]
        r += 1
        for b, Tb in enumerate((T, T1)):
            cb = dict(c, q=c["q"][b:b + 1], k_codes=c["k_codes"][b:b + 1, :, :Tb], v_codes=c["v_codes"][b:b + 1, :, :Tb],
                      k_res=k_hist[b:b + 1], v_res=v_hist[b:b + 1], r=r)
            _check(out[b:b + 1].cpu().numpy(), oracle.decode_attn(**cb), f"tile fused append step {step} request {b}")
    got = torch.roll(kr, -start, dims=2).cpu().numpy()
    np.testing.assert_array_equal(got[:, :, r0:r], k_hist[:, :, r0:r])
    if use_dl:
        assert lengths.cpu().numpy()[:, 1].tolist() == [r] * bs


def test_attn_tile_kernel_peaked_and_long(env, oracle):
    """Tile kernel: a late dominant key forces the running rescale; a long context exercises many tiles per wave and
    the maximum split count; G = 8 fills the workgroup partial."""
    torch, ops = env
    d, M, C = 64, 32, 256
    c = synth.attn_case(7200, 1, 8, 2, d, M, C, 3000, 9)
    kc = c["k_cents"].astype(np.float32)
    q0 = c["q"][0, 0, 0].astype(np.float32).reshape(M, d // M)
    kc[:, 0, :] = 3.0 * q0
    c["k_cents"] = kc.astype(np.float16)
    c["k_codes"][0, 0, 2500, :] = 0
    c["k_codes"][0, 0, 10, :M // 2] = 0
    _check(_run_paged(torch, ops, oracle, c, M, C, 64), oracle.decode_attn(**c), "tile peaked")
    # (round 5 moved these shapes to the streaming / lean kernels - 8 heads per kv head as two virtual kv heads of 4; policy 16 keeps
    # them on the tile kernel: both must agree with the oracle)
    try:
        for pol in (16, 0):
            ops.set_force_generic(pol)
            c = synth.attn_case(7201, 1, 16, 2, 128, 16, 256, 40000, 128, Lt=128)
            _check(_run_paged(torch, ops, oracle, c, 16, 256, 128), oracle.decode_attn(**c), f"long d128M16 G8 policy {pol}")
            c = synth.attn_case(7202, 2, 8, 8, 64, 16, 128, 33000, 77, Lt=128)
            _check(_run_rowmajor(torch, ops, c, 16, 128), oracle.decode_attn(**c), f"long d64M16 bs2 policy {pol}")
            # d = 64 with more than 128 tiles per CU: the two-workgroups-of-four-waves variant (smaller launches use 16 waves)
            for M, seed in ((32, 7203), (64, 7204)):
                c = synth.attn_case(seed, 3, 16, 8, 64, M, 256, 24000, 50, Lt=128)
                _check(_run_paged(torch, ops, oracle, c, M, 256, 64), oracle.decode_attn(**c), f"d64M{M} bs3 4-wave variant policy {pol}")
    finally:
        ops.set_force_generic(0)


def test_bindings_names_on_tile_shapes(env, oracle):
    """The generated binding names of the d = 64 / M = 16 part of the reference's build matrix (setup.py:12-15,48-53)."""
    torch, ops = env
    import bindings
    for d, M, C in [(64, 16, 128), (64, 32, 256), (64, 64, 256), (128, 16, 256)]:
        bs, nh, T, r, Ns = 1, 8, 1000, 17, 8
        c = synth.attn_case(7300 + d + M, bs, nh, nh, d, M, C, T, r)
        t = _dev(torch, c)
        fn = getattr(bindings, f"flash_decoding_allocated_buffer_f16u8_Ns{Ns}Lt{d}d{d}M{M}C{C}")
        po = torch.empty(bs, nh, Ns + 1, d, dtype=torch.float16, device="cuda")
        pl = torch.empty(bs, nh, Ns + 1, dtype=torch.float16, device="cuda")
        out = fn(t["q"], t["k_codes"], t["v_codes"], t["k_cents"], t["v_cents"], t["k_res"], t["v_res"], r, po, pl)   # Lt = d rows
        _check(out.cpu().numpy(), oracle.decode_attn(**c), f"bindings d{d}M{M}C{C}")


def test_attn_randomized_sweep(env, oracle):
    """Seeded sweep over the whole descriptor space of the fused call (both MFMA kernels and the scalar one behind it):
    d, M, C, group size, batch, context, window fill / capacity / ring start, page size, layout, id width."""
    torch, ops = env
    import os
    rs = np.random.RandomState(int(os.environ.get("MILLION_SWEEP_SEED", "20261004")))
    kinds = set()
    for it in range(int(os.environ.get("MILLION_SWEEP_CASES", "48"))):      # more cases / another seed for a soak run
        d = int(rs.choice([64, 128]))
        M = int(rs.choice([16, 32, 64]))
        C = int(rs.choice([128, 256]))
        nhk = int(rs.choice([1, 2, 4]))
        G = int(rs.choice([1, 2, 3, 4, 8, 12, 16]))
        bs = int(rs.choice([1, 2, 3]))
        T = int(rs.choice([0, 1, 15, 16, 17, 63, 64, 65, 511, 1000, 2049, 3000]))
        cap = int(rs.choice([64, 128, 256]))
        r = int(rs.randint(0 if T else 1, cap + 1))
        start = int(rs.randint(0, cap))
        ps = int(rs.choice([32, 64, 128]))
        layout = rs.choice(["rowmajor", "paged", "mixed"]) if T else "rowmajor"
        c = synth.attn_case(9000 + it, bs, nhk * G, nhk, d, M, C, T, r, Lt=cap)
        gold = oracle.decode_attn(**c)
        t = _dev(torch, c)
        kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
        kr = torch.roll(t["k_res"], start, dims=2).contiguous()
        vr = torch.roll(t["v_res"], start, dims=2).contiguous()
        kw = {}
        kc, vc = t["k_codes"], t["v_codes"]
        if layout != "rowmajor":
            vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
            kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
            perm = rs.permutation(vpool.shape[0])
            vpool, kpool, ids = vpool[perm], kpool[perm], np.argsort(perm)[ids]
            i64 = bool(rs.randint(2))
            ids_t = torch.from_numpy(ids.astype(np.int64 if i64 else np.int32)).cuda()
            vc = torch.from_numpy(vpool).cuda()
            kw = dict(v_page_ids=ids_t, page_size=ps, n_tokens=T)
            if layout == "paged":
                kc = torch.from_numpy(kpool).cuda()
                kw["k_page_ids"] = ids_t
        out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, kr, vr, r, M=M, C=C, resid_start=start, **kw)
        torch.cuda.synchronize()
        _check(out.cpu().numpy(), gold, f"sweep {it}: d={d} M={M} C={C} nhk={nhk} G={G} bs={bs} T={T} r={r}/{cap}@{start} ps={ps} {layout}")
        kinds.add((d, M))
    assert len(kinds) == 6


def test_two_streams_concurrent_calls(env, oracle):
    """The library launches on the caller's stream, keeps no per-call global state and wants one workspace per stream of
    concurrent calls: two streams hammer different shapes (streaming and tile kernels) at the same time."""
    torch, ops = env
    cases = [synth.attn_case(9100, 1, 32, 8, 128, 64, 256, 9000, 77, Lt=128),
             synth.attn_case(9101, 2, 8, 2, 64, 32, 256, 5000, 30, Lt=128)]
    golds = [oracle.decode_attn(**c) for c in cases]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    state = []
    for c, (M, C) in zip(cases, ((64, 256), (32, 256))):
        t = _dev(torch, c)
        vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], 64)
        kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], 64)
        state.append(dict(t=t, kp=ops.prepare_cents(t["k_cents"], cache=False), vp=ops.prepare_cents(t["v_cents"], cache=False),
                          kc=torch.from_numpy(kpool).cuda(), vc=torch.from_numpy(vpool).cuda(),
                          ids=torch.from_numpy(ids.astype(np.int32)).cuda(), M=M, C=C, T=c["k_codes"].shape[2], r=c["r"],
                          outs=[]))
    torch.cuda.synchronize()
    for it in range(20):
        for s, st in zip(streams, state):
            with torch.cuda.stream(s):
                t = st["t"]
                st["outs"].append(ops.pq_decode_attn(t["q"], st["kc"], st["vc"], st["kp"], st["vp"], t["k_res"], t["v_res"],
                                                     st["r"], M=st["M"], C=st["C"], n_tokens=st["T"], k_page_ids=st["ids"],
                                                     v_page_ids=st["ids"], page_size=64))
    torch.cuda.synchronize()
    for st, gold in zip(state, golds):
        for i, o in enumerate(st["outs"]):
            _check(o.cpu().numpy(), gold, f"stream call {i}")


@pytest.mark.parametrize("G", [12, 16, 32])
@pytest.mark.parametrize("d,M", [(128, 64), (128, 32), (64, 32), (128, 16)], ids=["stream64", "stream32", "tile-d64", "tile-m16"])
def test_attn_query_groups_above_eight(G, d, M, env, oracle):
    """nh / nh_k > 8 (e.g. 128 query heads over 8 kv heads): the MFMA kernels serve up to 16 heads per kv head in one launch
    (32: two), the tile and scalar kernels up to 8 per launch; the reference takes any group size (one kernel block per
    query head, Kernel.cuh:44-52)."""
    torch, ops = env
    nhk, T, r, C = 2, 1500, 40, 256
    c = synth.attn_case(9200 + G + d + M, 2, G * nhk, nhk, d, M, C, T, r, Lt=128)
    gold = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, M, C, 64), gold, f"G={G} paged")
    _check(_run_rowmajor(torch, ops, c, M, C), gold, f"G={G} rowmajor")
    ops.set_force_generic(True)
    try:
        _check(_run_rowmajor(torch, ops, c, M, C), gold, f"G={G} scalar kernel")
    finally:
        ops.set_force_generic(False)


@pytest.mark.parametrize("G", [16, 24], ids=["one-launch", "two-launches"])
@pytest.mark.parametrize("use_dl", [False, True], ids=["host-lengths", "device-lengths"])
def test_fused_append_with_sixteen_heads_per_kv_head(use_dl, G, env, oracle):
    """Fused append with G = 16 (one launch on the MFMA kernels since round 3) and G = 24 (launches of 16 + 8 heads): the
    row is appended once, every head attends to it, and device-resident r advances by exactly one per call."""
    torch, ops = env
    bs, nhk, d, M, C, T, r0, ps = 2, 2, 128, 64, 256, 700, 37, 64
    c = synth.attn_case(9300, bs, G * nhk, nhk, d, M, C, T, r0, Lt=128)
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"]), ops.prepare_cents(t["v_cents"])
    rs = np.random.RandomState(4)
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    kc, vc = torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda()
    kr, vr = t["k_res"].clone(), t["v_res"].clone()
    lengths = torch.tensor([[T, r0, 0, 0]] * bs, dtype=torch.int32, device="cuda") if use_dl else None
    k_hist, v_hist = c["k_res"].copy(), c["v_res"].copy()
    r = r0
    for step in range(3):
        k_new = rs.standard_normal((bs, nhk, 1, d)).astype(np.float16)
        v_new = rs.standard_normal((bs, nhk, 1, d)).astype(np.float16)
        out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, kr, vr, 0 if use_dl else r, M=M, C=C, dev_lengths=lengths,
                                 k_new=torch.from_numpy(k_new).cuda(), v_new=torch.from_numpy(v_new).cuda(),
                                 k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps, n_tokens=T)
        torch.cuda.synchronize()
        k_hist[:, :, r], v_hist[:, :, r] = k_new[:, :, 0], v_new[:, :, 0]
        r += 1
        _check(out.cpu().numpy(), oracle.decode_attn(**dict(c, k_res=k_hist, v_res=v_hist, r=r)), f"G={G} fused append step {step}")
    np.testing.assert_array_equal(kr.cpu().numpy()[:, :, r0:r], k_hist[:, :, r0:r])
    if use_dl:
        assert lengths.cpu().numpy()[:, 1].tolist() == [r] * bs


def test_debug_build_bounds_checks_page_ids(env):
    """`make debug-ids` (-DMILLION_DEBUG_CHECK_IDS): the three decode-attention kernels map page ids outside the pools to
    page 0 and count them; the product library trusts page ids like the reference (paged_pq_utils.py:440-441) and answers
    -1.  The diagnostic library is loaded by a child process (one library per process: million_amd/_lib.py)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    from million_amd import _lib
    assert _lib.load().million_debug_bad_page_ids() == -1            # product build: no check compiled in
    root = Path(__file__).resolve().parents[1]
    dbg = root / "million_amd" / "libmillion_hip_dbgids.so"
    if not dbg.exists():
        pytest.skip("million_amd/libmillion_hip_dbgids.so not built (make debug-ids)")
    env_ = dict(os.environ, MILLION_HIP_LIB=str(dbg))
    r = subprocess.run([sys.executable, str(root / "tests" / "dbgids_child.py")], env=env_, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["lib"] == dbg.name
    assert [c["name"] for c in res["cases"]] == ["stream", "stream-i64", "tile", "scalar"]
    for c in res["cases"]:
        assert c["clean_bad_ids"] == 0 and c["clean_err"] < 1e-3, c
        assert c["bad_ids"] >= 3 and c["finite"] and c["other_head_err"] < 1e-3 and c["hurt_head_moved"], c


# ---------------------------------------------------------------- size-independent properties at the BASELINE sizes --------
_BASELINE_SHAPES = [("configs2", 1, 32, 8, 64, 32768 + 64), ("configs3-per-gpu", 2, 32, 8, 64, 32768 + 64),
                    ("configs4", 1, 32, 8, 32, 131072), ("configs1", 1, 32, 32, 64, 4096)]


@pytest.mark.parametrize("name,bs,nh,nhk,M,T", _BASELINE_SHAPES, ids=[c[0] for c in _BASELINE_SHAPES])
def test_attn_properties_at_baseline_sizes(name, bs, nh, nhk, M, T, env):
    """No oracle at these sizes beyond the sampled comparisons elsewhere: properties of the operator itself, on the fused
    paged launch at every BASELINE.json shape.
    (1) the physical order of the pages does not matter: a shuffled pool + matching ids gives the SAME BITS (the kernel reads
        the same codes in the same order);
    (2) requests are independent: request 0 alone gives what it gives inside the batch (batch shapes; to rounding: it is cut
        into other splits), and two query heads fed the same q row give the same output row, bit for bit;
    (3) the output is linear in the value codebook: 2 x v_cents doubles it;
    (4) it is a convex combination: every output coordinate lies inside the range of the centroid / window values of its dim."""
    torch, ops = env
    d, C, ps, r = 128, 256, 64, 100
    g = torch.Generator(device="cuda").manual_seed(1234 + T + M + bs)
    n_pages = T // ps
    G = nh // nhk
    q = torch.randn(bs, nh, 1, d, device="cuda", generator=g).half()
    q[:, 1] = q[:, 0]                                               # (2) two heads of kv head 0 with the same query
    kc = torch.randn(M, C, d // M, device="cuda", generator=g).half()
    vc = torch.randn(M, C, d // M, device="cuda", generator=g).half()
    kres = torch.randn(bs, nhk, 128, d, device="cuda", generator=g).half()
    vres = torch.randn(bs, nhk, 128, d, device="cuda", generator=g).half()
    n_pool = bs * nhk * n_pages
    kpool = torch.randint(0, C, (n_pool, ps, M), device="cuda", generator=g, dtype=torch.uint8)
    vpool = torch.randint(0, C, (n_pool, M, ps), device="cuda", generator=g, dtype=torch.uint8)
    ids = torch.arange(n_pool, device="cuda", dtype=torch.int32).view(bs, nhk, n_pages)
    kp, vp = ops.prepare_cents(kc, cache=False), ops.prepare_cents(vc, cache=False)

    def run(kpool_, vpool_, ids_, vp_=vp, sel=slice(None)):
        out = ops.pq_decode_attn(q[sel].contiguous(), kpool_, vpool_, kp, vp_, kres[sel].contiguous(), vres[sel].contiguous(), r,
                                 M=M, C=C, n_tokens=T, k_page_ids=ids_[sel].contiguous(), v_page_ids=ids_[sel].contiguous(), page_size=ps)
        torch.cuda.synchronize()
        return out

    base = run(kpool, vpool, ids)
    assert torch.isfinite(base.float()).all()
    # (1) page order
    perm = torch.randperm(n_pool, device="cuda", generator=g)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n_pool, device="cuda")
    assert torch.equal(run(kpool[perm], vpool[perm], inv[ids.long()].int()), base)
    # (2) independence of requests and of heads
    if bs > 1:
        # (alone the request is cut into twice as many splits: another summation order, not another result)
        assert torch.allclose(run(kpool, vpool, ids, sel=slice(0, 1)).float(), base[0:1].float(), rtol=2e-3, atol=2e-4)
    if G > 1:
        assert torch.equal(base[:, 0], base[:, 1])
    # (3) linearity in the value codebook (the window's V rows are scaled with it)
    vres2 = vres * 2
    out2 = ops.pq_decode_attn(q, kpool, vpool, kp, ops.prepare_cents(vc * 2, cache=False), kres, vres2, r, M=M, C=C, n_tokens=T,
                              k_page_ids=ids, v_page_ids=ids, page_size=ps)
    torch.cuda.synchronize()
    # (not bit for bit: fp16 denormals among the centroids are flushed by the MFMA, their doubles are not)
    assert torch.allclose(out2.float(), base.float() * 2, rtol=2e-3, atol=2e-4)
    # (4) convexity: per output dim, within [min, max] of what a token can contribute there (centroid table / window rows)
    dm = d // M
    lo = torch.minimum(vc.float().amin(1).reshape(d), vres[:, :, :r].float().amin((0, 1, 2)))
    hi = torch.maximum(vc.float().amax(1).reshape(d), vres[:, :, :r].float().amax((0, 1, 2)))
    o = base.float().reshape(-1, d)
    assert (o >= lo - 1e-2).all() and (o <= hi + 1e-2).all()
    assert dm in (2, 4)


def test_encode_properties_at_baseline_size(env):
    """PQ encode at the 32K-token prompt size of configs[2] (8 kv heads): idempotence - a decoded row encodes to a code that
    decodes to the same row (the nearest centroid of a centroid is itself) - and layout independence: the codes
    that land in K pages / transposed V pages are the row-major codes, byte for byte."""
    torch, ops = env
    bs, nhk, n, d, M, C, ps = 1, 8, 32768, 128, 64, 256, 64
    g = torch.Generator(device="cuda").manual_seed(77)
    X = torch.randn(bs, nhk, n, d, device="cuda", generator=g).half()
    cents = torch.randn(M, C, d // M, device="cuda", generator=g).half()
    codes = ops.pq_encode(X, cents)
    dec = ops.pq_decode(codes, cents)
    again = ops.pq_encode(dec, cents)
    assert torch.equal(ops.pq_decode(again, cents), dec)          # (codes may differ only where two centroids are the same fp16 pair)
    assert (again != codes).sum().item() <= codes.numel() // 10000
    n_pages = n // ps
    ids = torch.randperm(bs * nhk * n_pages, device="cuda", generator=g).int().view(bs, nhk, n_pages)
    kpool = torch.zeros(bs * nhk * n_pages, ps, M, dtype=torch.uint8, device="cuda")
    vpool = torch.zeros(bs * nhk * n_pages, M, ps, dtype=torch.uint8, device="cuda")
    from million_amd import _lib as L
    ops.pq_encode_into(X, cents, kpool, layout=L.MILLION_CODES_KPAGES, page_ids=ids, page_size=ps, n=n, token_start=0)
    ops.pq_encode_into(X, cents, vpool, layout=L.MILLION_CODES_VPAGES, page_ids=ids, page_size=ps, n=n, token_start=0)
    torch.cuda.synchronize()
    want = codes.view(bs, nhk, n_pages, ps, M)
    assert torch.equal(kpool[ids.long()], want)
    assert torch.equal(vpool[ids.long()].transpose(-1, -2), want)


# ---------------------------------------------------------------- prompt (prefill) attention on fp16 K/V -------------------
def _sdpa_ref_rows(q, k, v, rows, q_pos0=0, causal=True):
    """fp64 reference of selected query rows: q (bs, nh, n_q, d), k / v (bs, nh_k, n_kv, d) numpy -> (bs, nh, len(rows), d)."""
    bs, nh, n_q, d = q.shape
    nhk, n_kv = k.shape[1], k.shape[2]
    G = nh // nhk
    out = np.zeros((bs, nh, len(rows), d))
    kf, vf = k.astype(np.float64), v.astype(np.float64)
    for b in range(bs):
        for h in range(nh):
            s = q[b, h, rows].astype(np.float64) @ kf[b, h // G].T / np.sqrt(d)      # (rows, n_kv)
            if causal:
                j = np.arange(n_kv)[None, :]
                s = np.where(j <= (q_pos0 + np.asarray(rows))[:, None], s, -np.inf)
            s = s - s.max(axis=1, keepdims=True)
            pr = np.exp(s)
            out[b, h] = (pr / pr.sum(axis=1, keepdims=True)) @ vf[b, h // G]
    return out


def _sdpa_ref_rows_at(qs, k, v, rows):
    """The same for query rows that were sliced out already: qs (bs, nh, len(rows), d) are the rows `rows` of a long causal
    prompt (the full q of a 128K prompt is 1 GB: only the sampled rows leave the GPU)."""
    bs, nh, _, d = qs.shape
    nhk, n_kv = k.shape[1], k.shape[2]
    G = nh // nhk
    out = np.zeros((bs, nh, len(rows), d))
    for b in range(bs):
        for hk in range(nhk):
            kf, vf = k[b, hk].astype(np.float64), v[b, hk].astype(np.float64)
            for h in range(hk * G, (hk + 1) * G):
                s = qs[b, h].astype(np.float64) @ kf.T / np.sqrt(d)
                s = np.where(np.arange(n_kv)[None, :] <= np.asarray(rows)[:, None], s, -np.inf)
                s = s - s.max(axis=1, keepdims=True)
                pr = np.exp(s)
                out[b, h] = (pr / pr.sum(axis=1, keepdims=True)) @ vf
    return out


@pytest.mark.parametrize("bs,nh,nhk,n_q,n_kv,q_pos0,causal", [
    (1, 8, 2, 1, 1, 0, True), (1, 8, 2, 33, 33, 0, True), (2, 4, 4, 100, 100, 0, True), (1, 6, 2, 257, 257, 0, True),
    (1, 8, 1, 300, 300, 0, True), (1, 16, 8, 130, 130, 0, True), (1, 8, 2, 64, 200, 136, True), (1, 8, 2, 70, 333, 0, False),
    (1, 32, 8, 1000, 1000, 0, True)])
def test_prefill_attn_small_shapes(bs, nh, nhk, n_q, n_kv, q_pos0, causal, env):
    """The MFMA prompt-attention kernel (csrc/prefill.hip) against an fp64 reference: ragged lengths (one row, one tile,
    partial last tile and last query block), every heads-per-workgroup grouping (G = 1, 2, 3 -> 1, 4, 8), batch 2, a
    prompt chunk behind cached rows (q_pos0 > 0, n_kv > n_q) and the non-causal form."""
    torch, ops = env
    rs = np.random.RandomState(900 + n_q + nh)
    q = rs.standard_normal((bs, nh, n_q, 128)).astype(np.float16)
    k = rs.standard_normal((bs, nhk, n_kv, 128)).astype(np.float16)
    v = rs.standard_normal((bs, nhk, n_kv, 128)).astype(np.float16)
    gold = _sdpa_ref_rows(q, k, v, list(range(n_q)), q_pos0, causal)
    try:
        for pol in (0, 64):      # the pipelined form (default at d = 128) and the plain one (million_set_force_generic(64))
            ops.set_force_generic(pol)
            out = ops.prefill_attn(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(),
                                   causal=causal, q_pos0=q_pos0)
            torch.cuda.synchronize()
            _check(out.cpu().numpy(), gold, f"prefill {bs} {nh} {nhk} {n_q} {n_kv} policy {pol}")
    finally:
        ops.set_force_generic(0)


def test_prefill_attn_strided_inputs_and_peaked_rows(env):
    """(bs, n, h, d) projections viewed as (bs, h, n, d) (what a model's q/k/v look like before .contiguous()), and a key
    that dominates late rows (forces the running-maximum rescale of the accumulators)."""
    torch, ops = env
    rs = np.random.RandomState(77)
    bs, nh, nhk, n, d = 1, 8, 2, 400, 128
    qkv = rs.standard_normal((bs, n, nh + 2 * nhk, d)).astype(np.float16)
    qkv[0, 300, nh] = 6.0 * qkv[0, 350, 0]          # key 300 of kv head 0 lines up with query 350 of head 0
    t = torch.from_numpy(qkv).cuda()
    q, k, v = t[:, :, :nh].transpose(1, 2), t[:, :, nh:nh + nhk].transpose(1, 2), t[:, :, nh + nhk:].transpose(1, 2)
    out = ops.prefill_attn(q, k, v)
    torch.cuda.synchronize()
    gold = _sdpa_ref_rows(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), list(range(n)))
    _check(out.cpu().numpy(), gold, "prefill strided")
    with pytest.raises(RuntimeError):
        ops.prefill_attn(q.float(), k, v)
    with pytest.raises(RuntimeError):
        ops.prefill_attn(torch.zeros(1, 8, 4, 32, dtype=torch.float16, device="cuda"),
                         torch.zeros(1, 2, 4, 32, dtype=torch.float16, device="cuda"),
                         torch.zeros(1, 2, 4, 32, dtype=torch.float16, device="cuda"))      # d = 32: not a head size of the reference


@pytest.mark.parametrize("bs,nh,nhk,n_q,n_kv,q_pos0,causal", [(1, 8, 2, 1, 1, 0, True), (2, 8, 8, 257, 257, 0, True),
                                                                (1, 6, 2, 333, 333, 0, True), (1, 16, 2, 64, 500, 436, True),
                                                                (1, 8, 4, 130, 700, 0, False), (1, 32, 8, 2048, 2048, 0, True)])
def test_prefill_attn_d64(bs, nh, nhk, n_q, n_kv, q_pos0, causal, env):
    """Head size 64 (the reference's other build, setup.py:12): 128-byte LDS rows with their own swizzle."""
    torch, ops = env
    rs = np.random.RandomState(1900 + n_q + nh)
    q = rs.standard_normal((bs, nh, n_q, 64)).astype(np.float16)
    k = rs.standard_normal((bs, nhk, n_kv, 64)).astype(np.float16)
    v = rs.standard_normal((bs, nhk, n_kv, 64)).astype(np.float16)
    out = ops.prefill_attn(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(),
                           causal=causal, q_pos0=q_pos0)
    torch.cuda.synchronize()
    _check(out.cpu().numpy(), _sdpa_ref_rows(q, k, v, list(range(n_q)), q_pos0, causal), f"prefill d64 {nh} {nhk} {n_q} {n_kv}")


def test_prefill_attn_llama_4k_vs_torch_fp32(env):
    """(1, 32 q heads, 8 kv heads, 4096 tokens, d 128): every output row against torch's fp32 CPU
    scaled_dot_product_attention(q, repeat_kv(k), repeat_kv(v), is_causal=True) - the reference's prompt attention
    (pq_utils.py:249-260) in fp32."""
    torch, ops = env
    g = torch.Generator().manual_seed(4)
    q = torch.randn(1, 32, 4096, 128, generator=g).half()
    k = torch.randn(1, 8, 4096, 128, generator=g).half()
    v = torch.randn(1, 8, 4096, 128, generator=g).half()
    out = ops.prefill_attn(q.cuda(), k.cuda(), v.cuda())
    torch.cuda.synchronize()
    ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float().repeat_interleave(4, dim=1),
                                                           v.float().repeat_interleave(4, dim=1), is_causal=True)
    _check(out.cpu().numpy(), ref.numpy(), "prefill 4k vs torch fp32")


def test_prefill_attn_32k_sampled_rows(env):
    """BASELINE configs[2]'s prompt length: (1, 32, 8, 32768, 128); sampled query rows (first, tile edges, last) of every
    head against the fp64 reference."""
    torch, ops = env
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(1, 32, 32768, 128, generator=g, device="cuda").half()
    k = torch.randn(1, 8, 32768, 128, generator=g, device="cuda").half()
    v = torch.randn(1, 8, 32768, 128, generator=g, device="cuda").half()
    out = ops.prefill_attn(q, k, v)
    torch.cuda.synchronize()
    rows = [0, 1, 63, 64, 65, 4095, 4096, 20000, 32703, 32704, 32767]
    gold = _sdpa_ref_rows(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), rows)
    _check(out[:, :, rows].cpu().numpy(), gold, "prefill 32k sampled rows")


def test_prefill_attn_128k_sampled_rows(env):
    """BASELINE configs[4]'s prompt length: (1, 32, 8, 131072, 128) - 4 x the rows of the 32K test, so the row / tile / key-tile
    index arithmetic of prefill.hip (t * kKV, pf_off, kv_end_wg) runs at the size the bench times it at.  First row, last
    row and rows at tile edges of every head against the fp64 reference (pq_utils.py:249-260: causal SDPA of the prompt)."""
    torch, ops = env
    n = 131072
    g = torch.Generator(device="cuda").manual_seed(6)
    q = torch.randn(1, 32, n, 128, generator=g, device="cuda").half()
    k = torch.randn(1, 8, n, 128, generator=g, device="cuda").half()
    v = torch.randn(1, 8, n, 128, generator=g, device="cuda").half()
    out = ops.prefill_attn(q, k, v)
    torch.cuda.synchronize()
    rows = [0, 63, 64, 255, 256, 65535, 65536, 100001, n - 257, n - 256, n - 1]
    qs = q[:, :, rows].cpu().numpy()
    gold = _sdpa_ref_rows_at(qs, k.cpu().numpy(), v.cpu().numpy(), rows)
    _check(out[:, :, rows].cpu().numpy(), gold, "prefill 128k sampled rows")
    del q, k, v, out
    torch.cuda.empty_cache()


def test_paged_cache_prefill_uses_hip_attention(env, oracle):
    """PagedPQCache.prefill / DynamicPQCache.prefill return the prompt attention of this library's kernel (no torch SDPA,
    no repeat_kv), also with distort_recent (the dequantised prompt is attended to)."""
    torch, ops = env
    from million_amd.pq_cache import DynamicPQCache, PagedPQCache
    rs = np.random.RandomState(31)
    bs, nh, nhk, n, d, M = 1, 8, 2, 200, 128, 64
    q = rs.standard_normal((bs, nh, n, d)).astype(np.float16)
    k = rs.standard_normal((bs, nhk, n, d)).astype(np.float16)
    v = rs.standard_normal((bs, nhk, n, d)).astype(np.float16)
    cents = rs.standard_normal((M, 256, 2)).astype(np.float16)
    gold = _sdpa_ref_rows(q, k, v, list(range(n)))
    for cls in (PagedPQCache, DynamicPQCache):
        cache = cls(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, max_tokens=1024, device="cuda")
        cache.set_cent(torch.from_numpy(cents).cuda(), torch.from_numpy(cents).cuda())
        out = cache.prefill(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(), 0)
        torch.cuda.synchronize()
        _check(out.cpu().numpy(), gold, cls.__name__)
    dyn = DynamicPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, max_tokens=1024, device="cuda")
    dyn.set_cent(torch.from_numpy(cents).cuda(), torch.from_numpy(cents).cuda())
    out = dyn.prefill(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(), 0, distort_recent=True)
    kq = oracle.pq_decode(oracle.pq_encode(k, cents), cents).astype(np.float16)
    vq = oracle.pq_decode(oracle.pq_encode(v, cents), cents).astype(np.float16)
    _check(out.cpu().numpy(), _sdpa_ref_rows(q, kq, vq, list(range(n))), "distort_recent")


def test_paged_cache_request_lifecycle_recycled_slot(env, oracle):
    """Request lifecycle on the paged store (reference allocator dynamic_paged_pq_utils.py:137-241, cleanup
    paged_pq_utils.py:1082-1118): batch slot 1 finishes, is released (pages back to the PageManager, lengths zeroed on the
    device) and starts a NEW request with a shorter prompt while slot 0 keeps decoding; from then on the two requests have
    different lengths, share every launch through the device-resident lengths, and flush their windows at different steps
    (the flush launch skips the request whose window is not full).  Every checked step: both outputs vs the fp64 oracle on
    each request's own history."""
    torch, ops = env
    from million_amd.pq_cache import PagedPQCache
    bs, nh, nhk, d, M, C, ps, cap = 2, 8, 2, 128, 64, 256, 64, 128
    rs = np.random.RandomState(123)
    ck = rs.standard_normal((M, C, 2)).astype(np.float16)
    cv = rs.standard_normal((M, C, 2)).astype(np.float16)
    cache = PagedPQCache(bs=bs, nh=nh, num_key_value_heads=nhk, M=M, layer_num=1, d=d, page_size=ps,
                         extended_residual_size=cap, max_tokens=1024, preallocate=False, device="cuda")
    cache.set_cent(torch.from_numpy(ck).cuda(), torch.from_numpy(cv).cuda())
    n0, n1, steps_a, steps_b = 200, 70, 100, 150
    total = n0 + steps_a + steps_b + 8
    K = rs.standard_normal((bs, nhk, total, d)).astype(np.float16)      # history of request (slot 0, slot 1 first request)
    V = rs.standard_normal((bs, nhk, total, d)).astype(np.float16)
    K2 = rs.standard_normal((1, nhk, total, d)).astype(np.float16)      # history of slot 1's second request
    V2 = rs.standard_normal((1, nhk, total, d)).astype(np.float16)
    Q = rs.standard_normal((steps_a + steps_b, bs, nh, 1, d)).astype(np.float16)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    cache.prefill(dev(Q[0]).expand(-1, -1, 1, -1).repeat(1, 1, n0, 1), dev(K[:, :, :n0]), dev(V[:, :, :n0]), 0)
    used_before = cache.page_manager.get_stats()["allocated_pages"]
    pol = [oracle.PagedPolicy(page_size=ps, residual=cap, prefill=n0) for _ in range(bs)]
    hist = [(K[0:1], V[0:1]), (K[1:2], V[1:2])]
    pos = [n0, n0]                                                        # next history row of each request

    def check(i, out):
        for b in range(bs):
            Kb, Vb = hist[b]
            T, r = pol[b].T, pol[b].r
            kc, vc = oracle.pq_encode(Kb[:, :, :T], ck), oracle.pq_encode(Vb[:, :, :T], cv)
            kres = np.zeros((1, nhk, cap, d), np.float16)
            vres = np.zeros((1, nhk, cap, d), np.float16)
            kres[:, :, :r], vres[:, :, :r] = Kb[:, :, T:T + r], Vb[:, :, T:T + r]
            gold = oracle.decode_attn(Q[i, b:b + 1], kc, vc, ck, cv, kres, vres, r)
            _check(out[b:b + 1], gold, f"step {i} request in slot {b} (T={T}, r={r})")

    for i in range(steps_a + steps_b):
        if i == steps_a:      # slot 1's request is done; a new one with a 70-token prompt takes the slot
            cache.release(1)
            st = cache.page_manager.get_stats()
            assert st["allocated_pages"] < used_before + 8 and cache.lengths[0].cpu().numpy()[1].tolist() == [0, 0, 0, 0]
            assert cache.lengths[0].cpu().numpy()[0, 0] == pol[0].T
            cache.prefill_request(1, dev(Q[0, 1:2]).expand(-1, -1, 1, -1).repeat(1, 1, n1, 1), dev(K2[:, :, :n1]), dev(V2[:, :, :n1]), 0)
            pol[1] = oracle.PagedPolicy(page_size=ps, residual=cap, prefill=n1)
            hist[1] = (K2, V2)
            pos[1] = n1
        kn = np.concatenate([hist[b][0][:, :, pos[b]:pos[b] + 1] for b in range(bs)])
        vn = np.concatenate([hist[b][1][:, :, pos[b]:pos[b] + 1] for b in range(bs)])
        out = cache.decoding_with_pages(dev(Q[i]), dev(kn), dev(vn), 0, use_dev_lengths=True)
        for b in range(bs):
            pol[b].step()
            pos[b] += 1
        if i % 23 == 0 or i in (steps_a - 1, steps_a, steps_a + 1, steps_a + 57, steps_a + 58, steps_a + 59, steps_a + steps_b - 1):
            torch.cuda.synchronize()
            dl = cache.lengths[0].cpu().numpy()
            assert [dl[b, :2].tolist() for b in range(bs)] == [[pol[b].T, pol[b].r] for b in range(bs)], (i, dl)
            check(i, out.float().cpu().numpy().astype(np.float64))
    # update(): the dequantise-then-attend path on pages, and cleanup()
    cache.cleanup()
    assert cache.page_manager.get_stats()["allocated_pages"] == 0 and not cache.lengths[0].cpu().numpy().any()
    kf, vf = cache.update(dev(K[:, :, :90]), dev(V[:, :, :90]), 0)
    np.testing.assert_array_equal(kf.cpu().numpy(), K[:, :, :90])
    kf2, vf2 = cache.update(dev(K[:, :, 90:100]), dev(V[:, :, 90:100]), 0)
    torch.cuda.synchronize()
    want = np.concatenate([oracle.pq_decode(oracle.pq_encode(K[:, :, :90], ck), ck).astype(np.float16), K[:, :, 90:100]], axis=2)
    np.testing.assert_array_equal(kf2.cpu().numpy(), want)
    want_v = np.concatenate([oracle.pq_decode(oracle.pq_encode(V[:, :, :90], cv), cv).astype(np.float16), V[:, :, 90:100]], axis=2)
    np.testing.assert_array_equal(vf2.cpu().numpy(), want_v)


def test_attn_two_streams_share_the_chip(env, oracle):
    """Two streams launch decode attention at the same time (separate workspaces): the workgroups of the two grids share the
    CUs, so a workgroup of one launch may wait long for its later arrivals to be dispatched - the case in which round 3's
    unbounded mergers could stall and merge stale partials.  The merge protocol (last arriver owns the merge, helpers with a
    bounded wait) must give the same outputs as the calls one after the other, with no fault."""
    torch, ops = env
    from million_amd import _lib
    nh, nhk, M, C, ps = 32, 8, 64, 256, 64
    cases = [synth.attn_case(9800 + i, 2, nh, nhk, 128, M, C, 9000 + 64 * i, 50 + i) for i in range(2)]
    golds = [oracle.decode_attn(**c) for c in cases]
    lib = _lib.load()
    lib.million_debug_tail_faults()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    prepared = []
    for c in cases:
        t = _dev(torch, c)
        vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
        kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
        prepared.append(dict(t=t, kpool=torch.from_numpy(kpool).cuda(), vpool=torch.from_numpy(vpool).cuda(),
                             ids=torch.from_numpy(ids.astype(np.int32)).cuda(),
                             kp=ops.prepare_cents(t["k_cents"], cache=False), vp=ops.prepare_cents(t["v_cents"], cache=False)))
    torch.cuda.synchronize()
    outs = [None, None]
    for rep in range(20):                       # many overlapping pairs of launches
        for i, (st, pr, c) in enumerate(zip(streams, prepared, cases)):
            with torch.cuda.stream(st):
                outs[i] = ops.pq_decode_attn(pr["t"]["q"], pr["kpool"], pr["vpool"], pr["kp"], pr["vp"], pr["t"]["k_res"],
                                             pr["t"]["v_res"], c["r"], M=M, C=C, n_tokens=c["k_codes"].shape[2],
                                             k_page_ids=pr["ids"], v_page_ids=pr["ids"], page_size=ps)
    torch.cuda.synchronize()
    assert lib.million_debug_tail_faults() == 0
    for o, g in zip(outs, golds):
        _check(o.cpu().numpy(), g, "two streams")


# ---------------------------------------------------------------- C ABI: error paths with real pointers, threads, the C caller ----
def _raw_attn_call(torch, ops, L, c, M, C, **over):
    """One million_pq_decode_attn call through ctypes with REAL device pointers; `over` tampers with single arguments."""
    lib = L.load()
    t = _dev(torch, c)
    kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
    T = c["k_codes"].shape[2]
    desc = ops.make_attn_desc(t["q"], t["k_res"], nh_k=t["k_res"].shape[1], M=M, C=C, n_tokens=T, r=c["r"],
                              k_codes=t["k_codes"], v_codes=t["v_codes"])
    for k in ("k_stride_h", "v_stride_b", "resid_stride_h"):
        if k in over:
            setattr(desc, k, over[k])
    need = lib.million_attn_workspace_bytes(ctypes.byref(desc))
    ws = torch.zeros(need + 64, dtype=torch.uint8, device="cuda")
    out = torch.full_like(t["q"], float("nan"))
    p = lambda name, x: x.data_ptr() + over.get(name + "_off", 0)
    rc = lib.million_pq_decode_attn(ctypes.byref(desc), p("q", t["q"]), p("k_codes", t["k_codes"]), p("v_codes", t["v_codes"]), 0, 0,
                                    p("k_prep", kp), p("v_prep", vp), p("k_res", t["k_res"]), p("v_res", t["v_res"]), p("out", out),
                                    p("ws", ws), need + over.get("ws_bytes_delta", 0), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rc, lib.million_last_error().decode(), out, need


def test_cabi_error_paths_with_real_pointers(env, oracle):
    """MILLION_ERR_WORKSPACE and MILLION_ERR_ALIGN (million_api.hip attn_impl / million_transpose_v_codes) with real device
    pointers: a workspace one byte short, pointers moved by 2 bytes, strides that break the 16-byte rule.  The reference checks
    nothing and exits the process on a launch error (Interface.template.cu:3-11); here every refusal is a status + a message,
    nothing is launched (the NaN-poisoned output stays untouched) and the very next good call works."""
    torch, ops = env
    from million_amd import _lib as L
    M, C = 64, 256
    c = synth.attn_case(9901, 1, 8, 2, 128, M, C, 700, 21)
    gold = oracle.decode_attn(**c)
    rc, msg, out, need = _raw_attn_call(torch, ops, L, c, M, C)
    assert rc == 0, msg
    _check(out.cpu().numpy(), gold, "raw ctypes call")
    rc, msg, out, _ = _raw_attn_call(torch, ops, L, c, M, C, ws_bytes_delta=-1)
    assert rc == -4 and "workspace" in msg and str(need) in msg, (rc, msg)            # MILLION_ERR_WORKSPACE, says how much it wants
    assert torch.isnan(out).all()
    for name in ("q", "k_codes", "v_codes", "k_res", "v_res", "out", "ws", "k_prep", "v_prep"):
        rc, msg, out, _ = _raw_attn_call(torch, ops, L, c, M, C, **{name + "_off": 2})
        assert rc == -2 and "16-byte aligned" in msg, (name, rc, msg)                  # MILLION_ERR_ALIGN
        assert torch.isnan(out).all() or name == "out"
    for k, v, what in (("k_stride_h", 700 * 64 + 8, "code strides"), ("v_stride_b", 2 * 700 * 64 + 1, "code strides"),
                       ("resid_stride_h", 128 * 128 + 4, "residual strides")):
        rc, msg, out, _ = _raw_attn_call(torch, ops, L, c, M, C, **{k: v})
        assert rc == -2 and what in msg, (k, rc, msg)
        assert torch.isnan(out).all()
    # the fused-append entry: k_new / v_new alignment; transpose entry: pointer and stride alignment
    lib = L.load()
    t = _dev(torch, c)
    kp = ops.prepare_cents(t["k_cents"], cache=False)
    desc = ops.make_attn_desc(t["q"], t["k_res"], nh_k=2, M=M, C=C, n_tokens=700, r=21, k_codes=t["k_codes"], v_codes=t["v_codes"])
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    new = torch.zeros(1, 2, 1, 136, dtype=torch.float16, device="cuda")
    rc = lib.million_pq_decode_attn_append(ctypes.byref(desc), t["q"].data_ptr(), new.data_ptr() + 2, new.data_ptr(), t["k_codes"].data_ptr(),
                                           t["v_codes"].data_ptr(), 0, 0, kp.data_ptr(), kp.data_ptr(), t["k_res"].data_ptr(),
                                           t["v_res"].data_ptr(), t["q"].data_ptr(), ws.data_ptr(), need, 0)
    assert rc == -2 and "k_new / v_new" in lib.million_last_error().decode()
    pages = torch.zeros(2 * 11 * 64 * 64 + 16, dtype=torch.uint8, device="cuda")
    assert lib.million_transpose_v_codes(t["v_codes"].data_ptr(), pages.data_ptr() + 2, 1, 2, 700, 64, 2 * 700 * 64, 700 * 64, 0) == -2
    assert lib.million_transpose_v_codes(t["v_codes"].data_ptr(), pages.data_ptr(), 1, 2, 700, 64, 2 * 700 * 64 + 4, 700 * 64, 0) == -2
    assert "16-byte alignment" in lib.million_last_error().decode()
    rc, msg, out, _ = _raw_attn_call(torch, ops, L, c, M, C)                           # and the library is none the worse for it
    assert rc == 0, msg
    _check(out.cpu().numpy(), gold, "good call after the refused ones")
    assert ops.tail_faults() == 0


def test_bindings_10arg_from_two_host_threads(env, oracle):
    """Re-entrancy ("no globals except a per-device table guarded by a mutex"): the reference's 10-argument call from
    TWO HOST THREADS at once, each on its own stream (ctypes drops the GIL around the C call, so the two threads are inside
    libmillion_hip.so together), on two shapes that take two different kernels; a third thread keeps provoking refused calls and
    reads its own thread-local million_last_error().  Every output against the oracle."""
    import threading
    torch, ops = env
    import bindings
    from million_amd import _lib as L
    lib = L.load()
    specs = [(9911, 1, 32, 8, 128, 64, 256, 6000, 77), (9912, 2, 8, 2, 64, 32, 256, 3000, 30)]
    cases = [synth.attn_case(s, bs, nh, nhk, d, M, C, T, r, Lt=d) for s, bs, nh, nhk, d, M, C, T, r in specs]
    golds = [oracle.decode_attn(**c) for c in cases]
    n_calls, results, errors = 40, [[], []], []
    start = threading.Barrier(3)

    def worker(i):
        try:
            _, bs, nh, nhk, d, M, C, T, r = specs[i]
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                t = _dev(torch, cases[i])
                fn = getattr(bindings, f"flash_decoding_allocated_buffer_f16u8_Ns16Lt{d}d{d}M{M}C{C}")
                po = torch.empty(bs, nh, 17, d, dtype=torch.float16, device="cuda")
                pl = torch.empty(bs, nh, 17, dtype=torch.float16, device="cuda")
                stream.synchronize()
                start.wait()
                for _ in range(n_calls):
                    results[i].append(fn(t["q"], t["k_codes"], t["v_codes"], t["k_cents"], t["v_cents"], t["k_res"], t["v_res"], r, po, pl))
                stream.synchronize()
        except Exception as e:      # noqa: BLE001 - reported by the main thread
            errors.append((i, repr(e)))

    def refuser():
        try:
            start.wait()
            for k in range(200):
                assert lib.million_workspace_init(0, 16, 0) == -3
                assert lib.million_last_error() == b"workspace_init: null"
                assert lib.million_lengths_advance(0, 1, 64, 128, 0) == -3
                assert lib.million_last_error() == b"lengths_advance: bad argument"
        except Exception as e:      # noqa: BLE001
            errors.append(("refuser", repr(e)))

    threads = [threading.Thread(target=worker, args=(0,)), threading.Thread(target=worker, args=(1,)), threading.Thread(target=refuser)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(2):
        assert len(results[i]) == n_calls
        for k, o in enumerate(results[i]):
            _check(o.cpu().numpy(), golds[i], f"thread {i} call {k}")
    assert ops.tail_faults() == 0


def test_cabi_bench_c_program(env):
    """tools/cabi_bench.c - the plain-C caller of the ABI (no Python, no torch; the reference's counterpart is the C++ harness
    Kernel_Test/main.cu:59-226): codebook -> prompt encoded into pages -> decode launches through million_pq_decode_attn_append
    (graph replay) -> its own host check of codes (bit-exact) and outputs (1e-3).  Built here if the box has no binary yet."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    exe = root / "build" / "cabi_bench"
    src = root / "tools" / "cabi_bench.c"
    if not exe.exists() or exe.stat().st_mtime < src.stat().st_mtime:
        b = subprocess.run(["make", "-C", str(root), "cabi-bench"], capture_output=True, text=True, timeout=600)
        assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    for extra in (["--ctx", "4096", "--layers", "4", "--launches", "16"], ["--ctx", "8192", "--M", "32", "--bs", "2", "--layers", "2", "--launches", "8"]):
        r = subprocess.run([str(exe), *extra, "--reps", "2"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "cabi_bench: PASS" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
        assert "0 of" in r.stdout and "kernel kind 1" in r.stdout, r.stdout


# ---------------------------------------------------------------- lean kernel (round 5) ----------------------------------------
@pytest.mark.parametrize("policy", [0, 16], ids=["lean", "streaming-form"])
@pytest.mark.parametrize("G", [1, 2, 3, 4])
def test_attn_lean_kernel_unit_counts_and_forms(G, policy, env, oracle):
    """The lean kernel (attn_mfma.hip attn_lean_kernel: 64-token units, lane = token, 4x4x4 score products, z-row value products)
    takes M = 64 / C = 256 / up to 4 query heads per kv head / pages of 64 or 128 tokens; million_set_force_generic(16) keeps those
    shapes on the streaming kernel's parity-V form, which otherwise only runs at G > 4 or on 32-token pages.  Context lengths chosen
    so that a wave has 0, 1, 2, 3, 4 and 5 units (the scores-alone prologue, the block chain with its odd / even ends, the
    values-alone epilogue), with and without a ragged last unit, a last unit of one token, and windows of 1 .. 128 rows."""
    torch, ops = env
    nhk, M, C = 2, 64, 256
    try:
        ops.set_force_generic(policy)
        for T, r, ps in ((1, 1, 64), (63, 5, 64), (64, 128, 64), (65, 17, 128), (511, 3, 64), (513, 64, 64), (1024, 100, 128), (1537, 31, 64),
                         (2048 + 17, 9, 64), (2560, 77, 128), (4096 + 65, 127, 64), (5120 + 1, 2, 64)):
            c = synth.attn_case(9300 + T + G, 1, G * nhk, nhk, 128, M, C, T, r, Lt=128)
            gold = oracle.decode_attn(**c)
            _check(_run_paged(torch, ops, oracle, c, M, C, ps, poison_out=True), gold, f"G={G} T={T} r={r} ps={ps} paged")
            _check(_run_paged(torch, ops, oracle, c, M, C, ps, k_paged=False, i64=True), gold, f"G={G} T={T} row-major K, int64 ids")
            _check(_run_rowmajor(torch, ops, c, M, C), gold, f"G={G} T={T} 10-arg layout")
    finally:
        ops.set_force_generic(0)
    assert ops.tail_faults() == 0


@pytest.mark.parametrize("M", [64, 32, 16], ids=["dm1-padded", "dm2", "dm4"])
def test_attn_lean_kernel_d64_forms(M, env, oracle):
    """d = 64 on the lean kernel: M = 32 / 16 (d_m = 2 / 4) and M = 64 (d_m = 1, run as d_m = 2 with every odd dim zero: codebooks
    widened into LDS, accumulators packed back in front of the tail) over 0 .. 5 units per wave, windows of 1 .. 128 rows, every
    head grouping up to 4, the fused append (its residual rows are read at d = 64)."""
    torch, ops = env
    C = 256
    for G, nhk in ((4, 2), (3, 1), (1, 4), (2, 2), (8, 2), (6, 1), (16, 1), (12, 2), (5, 2), (7, 1), (9, 1), (13, 1)):      # 5 .. 16: virtual kv heads of 3 / 4 query heads
        for T, r, ps in ((1, 1, 64), (64, 128, 64), (513, 64, 128), (1537, 31, 64), (4096 + 65, 127, 64), (5120 + 1, 2, 128)):
            C = 128 if (T + G) % 3 == 0 else 256      # 128 centroids: the K rows are spread to the 256-entry stride on their way into LDS
            c = synth.attn_case(9500 + T + G + M, 1, G * nhk, nhk, 64, M, C, T, r, Lt=128)
            gold = oracle.decode_attn(**c)
            t = _dev(torch, c)
            assert _kind(torch, ops, t["q"], t["k_res"], nh_k=nhk, M=M, C=C, n_tokens=T, r=r, k_paged=True, v_paged=True,
                         page_size=ps, n_pages_cap=(T + ps - 1) // ps) == 1
            _check(_run_paged(torch, ops, oracle, c, M, C, ps, poison_out=True), gold, f"M={M} G={G} T={T} r={r} ps={ps} paged")
            _check(_run_paged(torch, ops, oracle, c, M, C, ps, k_paged=False, i64=True), gold, f"M={M} G={G} T={T} row-major K, int64 ids")
            _check(_run_rowmajor(torch, ops, c, M, C), gold, f"M={M} G={G} T={T} 10-arg layout")
    C = 256
    # fused append at d = 64: the new row joins the window (and the attention) inside the launch; G = 8 runs two parts per kv head
    # (only part 0 stores the row, both attend to it), two requests
    for nhk, G, T, r in ((2, 4, 3000, 40), (2, 8, 3000, 40), (2, 7, 3000, 40)):
        c = synth.attn_case(9600 + M + G, 2, G * nhk, nhk, 64, M, C, T, r + 1, Lt=128)
        gold = oracle.decode_attn(**c)
        t = _dev(torch, c)
        vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], 64)
        kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], 64)
        kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
        k_res, v_res = t["k_res"].clone(), t["v_res"].clone()
        k_new, v_new = k_res[:, :, r:r + 1].clone(), v_res[:, :, r:r + 1].clone()
        k_res[:, :, r] = 0
        v_res[:, :, r] = 0
        ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
        out = ops.pq_decode_attn(t["q"], torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda(), kp, vp, k_res, v_res, r, M=M, C=C,
                                 n_tokens=T, k_page_ids=ids_t, v_page_ids=ids_t, page_size=64, k_new=k_new, v_new=v_new)
        torch.cuda.synchronize()
        _check(out.cpu().numpy(), gold, f"M={M} G={G} fused append")
        assert torch.equal(k_res[:, :, r], t["k_res"][:, :, r]) and torch.equal(v_res[:, :, r], t["v_res"][:, :, r])
    assert ops.tail_faults() == 0


def test_attn_lean_kernel_many_units_per_wave_and_batch(env, oracle):
    """Long splits (many blocks per wave through the pair loop), a batch whose requests have different lengths (device lengths: a
    wave of the short request has fewer units than the grid was sized for), the fused append, peaked scores that move the softmax
    reference in the middle of a wave's units (the lean softmax rescales its 8 accumulator registers by quad broadcasts)."""
    torch, ops = env
    M, C, ps = 64, 256, 64
    c = synth.attn_case(9400, 2, 8, 2, 128, M, C, 40000, 50, Lt=128)
    # request 1 is shorter; a handful of K rows far larger than the rest (late in the context: the reference has to move)
    rs = np.random.RandomState(3)
    T1 = 23000
    c["q"][:, :, :, :] *= 3.0
    gold_full = oracle.decode_attn(**c)
    _check(_run_paged(torch, ops, oracle, c, M, C, ps), gold_full, "40K tokens, 2 requests")
    t = _dev(torch, c)
    vpool, ids = oracle.v_rowmajor_to_pool(c["v_codes"], ps)
    kpool, _ = oracle.k_rowmajor_to_pool(c["k_codes"], ps)
    kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
    lengths = torch.tensor([[40000, 50, 0, 0], [T1, 49, 0, 0]], dtype=torch.int32, device="cuda")
    k_new = rs.standard_normal((2, 2, 1, 128)).astype(np.float16)
    v_new = rs.standard_normal((2, 2, 1, 128)).astype(np.float16)
    ids_t = torch.from_numpy(ids.astype(np.int32)).cuda()
    out = ops.pq_decode_attn(t["q"], torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda(), kp, vp, t["k_res"], t["v_res"], 0, M=M, C=C,
                             n_tokens=40000, k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps, dev_lengths=lengths,
                             k_new=torch.from_numpy(k_new).cuda(), v_new=torch.from_numpy(v_new).cuda())
    torch.cuda.synchronize()
    for b, (Tb, rb) in enumerate(((40000, 50), (T1, 49))):
        cb = {k: (v[b:b + 1].copy() if isinstance(v, np.ndarray) and v.shape[0] == 2 else v) for k, v in c.items()}
        cb["k_codes"], cb["v_codes"] = cb["k_codes"][:, :, :Tb], cb["v_codes"][:, :, :Tb]
        cb["k_res"][:, :, rb], cb["v_res"][:, :, rb] = k_new[b:b + 1, :, 0], v_new[b:b + 1, :, 0]
        cb["r"] = rb + 1
        _check(out[b:b + 1].cpu().numpy(), oracle.decode_attn(**cb), f"request {b}: T={Tb}, appended row {rb}")
    assert lengths.cpu().numpy()[:, 1].tolist() == [51, 50]
    assert ops.tail_faults() == 0
