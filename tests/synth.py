"""Deterministic synthetic inputs shared by tools/gen_golden.py, the tests and bench.py.

Distributions follow the reference's own micro-benchmark (scripts/modeldb/bindings/test_kernel.py:59-65):
query / residuals / centroids ~ N(0,1) fp16, codes ~ U{0..C-1} u8; encode inputs X ~ N(0,1) fp16.
numpy's legacy RandomState is used because its streams are frozen across numpy versions, so a
fixture generated in one container reproduces bit for bit on the GPU box.
"""
from __future__ import annotations

import numpy as np


def attn_case(seed, bs, nh, nh_k, d, M, C, T, r, Lt=None):
    Lt = Lt or d
    rs = np.random.RandomState(seed)
    dm = d // M
    q = rs.standard_normal((bs, nh, 1, d)).astype(np.float16)
    k_codes = rs.randint(0, C, size=(bs, nh_k, T, M)).astype(np.uint8)
    v_codes = rs.randint(0, C, size=(bs, nh_k, T, M)).astype(np.uint8)
    k_cents = rs.standard_normal((M, C, dm)).astype(np.float16)
    v_cents = rs.standard_normal((M, C, dm)).astype(np.float16)
    k_res = rs.standard_normal((bs, nh_k, Lt, d)).astype(np.float16)
    v_res = rs.standard_normal((bs, nh_k, Lt, d)).astype(np.float16)
    return dict(q=q, k_codes=k_codes, v_codes=v_codes, k_cents=k_cents, v_cents=v_cents,
                k_res=k_res, v_res=v_res, r=r)


def encode_case(seed, bs, nh_k, n, d, M, C):
    rs = np.random.RandomState(seed)
    X = rs.standard_normal((bs, nh_k, n, d)).astype(np.float16)
    cents = rs.standard_normal((M, C, d // M)).astype(np.float16)
    return dict(X=X, cents=cents)


# Small cases whose full expected outputs are committed under tests/golden/ (SURVEY.md 8c).
GOLDEN_ATTN = [
    # name,            seed, bs, nh, nh_k, d,  M,  C,   T,    r
    ("t0_r17",          101, 1,  8,  2,  128, 64, 256, 0,    17),
    ("t1_r1",           102, 1,  8,  2,  128, 64, 256, 1,    1),
    ("t63_r128",        103, 1,  8,  2,  128, 64, 256, 63,   128),
    ("t64_r64",         104, 1,  8,  8,  128, 64, 256, 64,   64),
    ("t129_r17",        105, 1,  8,  2,  128, 64, 256, 129,  17),
    ("t257_r17_m32",    106, 1,  8,  2,  128, 32, 256, 257,  17),
    ("t1000_r17",       107, 1,  32, 8,  128, 64, 256, 1000, 17),
    ("t1000_r17_mha",   108, 2,  4,  4,  128, 64, 256, 1000, 17),
    ("t300_d64_m16_c128", 109, 1, 4, 2,  64,  16, 128, 300,  33),
]

GOLDEN_ENCODE = [
    # name,        seed, bs, nh_k, n,   d,   M,  C
    ("n1",          201, 1,  2,   1,   128, 64, 256),
    ("n64",         202, 1,  2,   64,  128, 64, 256),
    ("n129_m32",    203, 2,  2,   129, 128, 32, 256),
    ("n128",        204, 1,  8,   128, 128, 64, 256),
    ("n100_d64_m16_c128", 205, 1, 2, 100, 64, 16, 128),
]

# nbits 9..12 (uint16 codes, reference nbits2dtype pq_utils.py:542-552): sa_encode_4d(target_dtype=uint16) outputs.
GOLDEN_ENCODE_U16 = [
    # name,          seed, bs, nh_k, n,  d,   M,  C
    ("u16_n40_c512",   211, 1,  2,   40, 128, 64, 512),
    ("u16_n33_c1024_m32", 212, 2, 1, 33, 128, 32, 1024),
    ("u16_n20_c4096",  213, 1,  1,   20, 128, 64, 4096),
]

# Large encode case: only a SHA-256 of the reference codes and the list of disagreeing positions.
GOLDEN_ENCODE_BIG = ("cfg1_n4096", 42, 1, 8, 4096, 128, 64, 256)

# BASELINE configs[4]-sized prefill encode (128K tokens, M = 32), one kv head: SHA-256 + flip list in the manifest.
GOLDEN_ENCODE_128K = ("cfg4_n131072_m32", 43, 1, 1, 131072, 128, 32, 256)
