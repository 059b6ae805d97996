#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python functions on CPU.

Run in the build container only (needs /root/reference; the GPU box has no reference):
    python tools/gen_golden.py

What is imported from the reference (scripts/utils/pq_utils.py): sa_encode_4d (:410-449),
sa_decode_4d (:501-540), l2Ns (:8-22), nbits2dtype (:542-552) — pure torch, CPU-runnable.
`pykeops` (third-party, pinned 2.2.3, absent here) is only needed by sa_encode_4d_keops, which is
NOT called; a two-line placeholder module lets `from pykeops.torch import LazyTensor` at the top
of pq_utils.py succeed.  Nothing from the reference is copied into the repo: the
fixtures hold inputs' seeds and the reference's OUTPUTS only.

Attention fixture = the reference's own check formula (pq_utils.py:360-368):
    sdpa(q, cat(sa_decode_4d(Kc), K_res[:r]), cat(sa_decode_4d(Vc), V_res[:r]))   (non-causal)
evaluated in fp32 with torch on CPU, GQA by repeat_interleave (== transformers.repeat_kv).
"""
import hashlib
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True

from tests import synth  # noqa: E402


def import_reference():
    pk = types.ModuleType("pykeops")
    pkt = types.ModuleType("pykeops.torch")

    class LazyTensor:  # placeholder: the KeOps path is never called here
        def __init__(self, *a, **k):
            raise RuntimeError("pykeops is not installed")

    pkt.LazyTensor = LazyTensor
    pk.torch = pkt
    sys.modules.setdefault("pykeops", pk)
    sys.modules.setdefault("pykeops.torch", pkt)
    sys.path.insert(0, "/root/reference")
    from scripts.utils import pq_utils as R
    return R


def main():
    R = import_reference()
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    manifest = {"generator": "tools/gen_golden.py", "torch": torch.__version__,
                "reference_functions": ["sa_encode_4d", "sa_decode_4d", "l2Ns", "nbits2dtype"]}

    # ---- l2Ns table ----
    ls = [0, 1, 63, 64, 65, 128, 129, 256, 257, 2048, 2049, 4096, 32768, 131072]
    manifest["l2Ns"] = {str(l): int(R.l2Ns(l)) for l in ls}
    manifest["nbits2dtype"] = {str(n): str(R.nbits2dtype(n)) for n in (4, 8, 9, 12, 16)}

    # ---- attention cases ----
    for (name, seed, bs, nh, nhk, d, M, C, T, r) in synth.GOLDEN_ATTN:
        c = synth.attn_case(seed, bs, nh, nhk, d, M, C, T, r)
        t = {k: torch.from_numpy(v) for k, v in c.items() if k != "r"}
        kc32, vc32 = t["k_cents"].float(), t["v_cents"].float()
        K_hat = R.sa_decode_4d(t["k_codes"], kc32)
        V_hat = R.sa_decode_4d(t["v_codes"], vc32)
        K = torch.cat([K_hat, t["k_res"].float()[:, :, :r]], dim=2)
        V = torch.cat([V_hat, t["v_res"].float()[:, :, :r]], dim=2)
        G = nh // nhk
        K = K.repeat_interleave(G, dim=1)
        V = V.repeat_interleave(G, dim=1)
        out = torch.nn.functional.scaled_dot_product_attention(t["q"].float(), K, V)  # non-causal
        np.savez_compressed(out_dir / f"attn_{name}.npz",
                            out=out.numpy().astype(np.float32),
                            k_hat_sum=K_hat.double().sum(dim=(2,)).numpy(),  # decode checksum per (b,h,dim)
                            params=np.array([seed, bs, nh, nhk, d, M, C, T, r], dtype=np.int64))
        print("attn", name, tuple(out.shape))

    # ---- encode cases: reference sa_encode_4d (cdist form) on fp32-upcast inputs ----
    for (name, seed, bs, nhk, n, d, M, C) in synth.GOLDEN_ENCODE:
        c = synth.encode_case(seed, bs, nhk, n, d, M, C)
        X, cents = torch.from_numpy(c["X"]).float(), torch.from_numpy(c["cents"]).float()
        codes = R.sa_encode_4d(X, cents)
        dec = R.sa_decode_4d(codes, cents)
        np.savez_compressed(out_dir / f"encode_{name}.npz", codes=codes.numpy(),
                            decoded=dec.numpy().astype(np.float16),
                            params=np.array([seed, bs, nhk, n, d, M, C], dtype=np.int64))
        print("encode", name, tuple(codes.shape))

    # ---- nbits 9..12: uint16 codes (nbits2dtype(nbits), pq_utils.py:542-552, as main_pq.py passes it) ----
    for (name, seed, bs, nhk, n, d, M, C) in synth.GOLDEN_ENCODE_U16:
        c = synth.encode_case(seed, bs, nhk, n, d, M, C)
        X, cents = torch.from_numpy(c["X"]).float(), torch.from_numpy(c["cents"]).float()
        nbits = int(np.log2(C))
        assert R.nbits2dtype(nbits) == torch.uint16
        codes = R.sa_encode_4d(X, cents, target_dtype=R.nbits2dtype(nbits))
        dec = R.sa_decode_4d(codes, cents)
        np.savez_compressed(out_dir / f"encode_{name}.npz", codes=codes.numpy().astype(np.uint16),
                            decoded=dec.numpy().astype(np.float16),
                            params=np.array([seed, bs, nhk, n, d, M, C], dtype=np.int64))
        print("encode", name, tuple(codes.shape), codes.dtype)

    # ---- big encode case: hash + the reference's code values where the two forms disagree ----
    (name, seed, bs, nhk, n, d, M, C) = synth.GOLDEN_ENCODE_BIG
    c = synth.encode_case(seed, bs, nhk, n, d, M, C)
    X, cents = torch.from_numpy(c["X"]).float(), torch.from_numpy(c["cents"]).float()
    codes = R.sa_encode_4d(X, cents).numpy()
    # positions where the cdist form differs from the direct-form oracle are data, not source:
    sys.path.insert(0, str(ROOT))
    from oracle import oracle as O
    direct, gap = O.pq_encode_with_gap(c["X"], c["cents"])
    diff = np.argwhere(direct != codes)
    manifest["encode_big"] = {
        "name": name, "sha256_reference_cdist_codes": hashlib.sha256(codes.tobytes()).hexdigest(),
        "sha256_direct_oracle_codes": hashlib.sha256(direct.tobytes()).hexdigest(),
        "n_codes": int(codes.size), "n_diff_cdist_vs_direct": int(diff.shape[0]),
        "diff_positions": diff.tolist()[:64],
        "diff_gaps": [float(gap[tuple(p)]) for p in diff[:64]],
        # the reference's (cdist-form) code at each disagreeing position: patching them into the direct-form codes
        # must reproduce sha256_reference_cdist_codes (tests/test_oracle.py)
        "diff_reference_codes": [int(codes[tuple(p)]) for p in diff[:64]],
        "diff_direct_codes": [int(direct[tuple(p)]) for p in diff[:64]],
    }
    print("encode big:", manifest["encode_big"]["n_diff_cdist_vs_direct"], "of", codes.size, "differ")

    # ---- BASELINE configs[4]-sized prefill encode (128K tokens, M = 32, one kv head): hashes + flip list, as above ----
    (name, seed, bs, nhk, n, d, M, C) = synth.GOLDEN_ENCODE_128K
    c = synth.encode_case(seed, bs, nhk, n, d, M, C)
    X, cents = torch.from_numpy(c["X"]).float(), torch.from_numpy(c["cents"]).float()
    codes = R.sa_encode_4d(X, cents).numpy()      # one call on the whole prompt, as the reference's prefill makes it
    direct, gap = O.pq_encode_with_gap(c["X"], c["cents"])
    diff = np.argwhere(direct != codes)
    manifest["encode_128k_m32"] = {
        "name": name, "sha256_reference_cdist_codes": hashlib.sha256(codes.tobytes()).hexdigest(),
        "sha256_direct_oracle_codes": hashlib.sha256(direct.tobytes()).hexdigest(),
        "n_codes": int(codes.size), "n_diff_cdist_vs_direct": int(diff.shape[0]),
        "diff_positions": diff.tolist()[:256],
        "diff_gaps": [float(gap[tuple(p)]) for p in diff[:256]],
        "diff_reference_codes": [int(codes[tuple(p)]) for p in diff[:256]],
        "diff_direct_codes": [int(direct[tuple(p)]) for p in diff[:256]],
    }
    print("encode 128k m32:", manifest["encode_128k_m32"]["n_diff_cdist_vs_direct"], "of", codes.size, "differ")

    # ---- on-disk formats written BY the reference's own writers (data files) ----
    # .fvecs: scripts/utils/fvecio.py:35-43 write_fvecs (default append mode: creates, then appends)
    from scripts.utils import fvecio as RF
    rs = np.random.RandomState(77)
    va, vb = rs.standard_normal((3, 5)).astype(np.float32), rs.standard_normal((2, 5)).astype(np.float32)
    fv = out_dir / "ref_written.fvecs"
    if fv.exists():
        fv.unlink()
    RF.write_fvecs(fv, va)
    RF.write_fvecs(fv, vb)
    back = RF.read_fvecs(fv)
    assert back.shape == (5, 5)
    # .pq.pt: main_pq.py:222-226 `from torch import save; save(key_cent, cent_root / f'key_cent_{M}_{nbits}.pq.pt')` with
    # key_cent the fp32 (M, 2**nbits, d/M) tensor train_pq returns (pq_utils.py:586-609; faiss itself is absent here,
    # so the VALUES are synthetic - the file format is the reference's call)
    from torch import save
    kc = torch.from_numpy(rs.standard_normal((4, 4, 2)).astype(np.float32))
    vc = torch.from_numpy(rs.standard_normal((4, 4, 2)).astype(np.float32))
    save(kc, out_dir / "key_cent_4_2.pq.pt")
    save(vc, out_dir / "val_cent_4_2.pq.pt")
    manifest["formats"] = {"fvecs_file": fv.name, "fvecs_expected": np.concatenate([va, vb]).tolist(),
                           "fvecs_sha256": hashlib.sha256(fv.read_bytes()).hexdigest(),
                           "pq_pt": {"M": 4, "nbits": 2, "d": 8, "key": kc.tolist(), "val": vc.tolist()}}
    (out_dir / "manifest.json").write_text(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
